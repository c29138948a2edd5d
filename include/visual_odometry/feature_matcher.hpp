// GPU-backed drop-in for the reference's feature_matcher interface
// (ref:include/visual_odometry/feature_matcher.hpp:12-44).
#pragma once

#include <span>
#include <vector>

#include <visual_odometry/compat/cv_min.hpp>

namespace visual_odometry {

constexpr auto default_ratio_threshold = 0.75f;   // Lowe's ratio

struct match_result {
    std::vector<cv::Point2f> points1;   // keypoints1[queryIdx].pt of every surviving match
    std::vector<cv::Point2f> points2;   // keypoints2[trainIdx].pt
    std::vector<cv::DMatch> matches;    // in query order, imgIdx = 0
};

struct feature_matcher_config {
    float ratio_threshold{default_ratio_threshold};
};

// Brute-force Hamming 2-NN of descriptors1 in descriptors2 + ratio test
// (d0 < ratio * d1, float32).  Either side empty -> empty result.
[[nodiscard]] auto match_features(cv::Mat const& descriptors1, cv::Mat const& descriptors2,
                                  std::span<cv::KeyPoint const> keypoints1,
                                  std::span<cv::KeyPoint const> keypoints2,
                                  feature_matcher_config const& config = {}) -> match_result;

// The two images side by side in colour with a line per match
// (ref:include/visual_odometry/feature_matcher.hpp:55, ref:src/feature_matcher.cpp:44-55:
// cv::drawMatches, green, unmatched keypoints not drawn).  Visualisation only, host code.
[[nodiscard]] auto draw_feature_matches(cv::Mat const& image1, std::vector<cv::KeyPoint> const& keypoints1,
                                        cv::Mat const& image2, std::vector<cv::KeyPoint> const& keypoints2,
                                        std::vector<cv::DMatch> const& matches) -> cv::Mat;

}  // namespace visual_odometry
