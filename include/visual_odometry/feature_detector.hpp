// GPU-backed drop-in for the reference's feature_detector interface
// (ref:include/visual_odometry/feature_detector.hpp:11-47).  Same names,
// defaults and by-value results; the work is done by libvo_hip.so on an MI355X.
#pragma once

#include <vector>

#include <visual_odometry/compat/cv_min.hpp>

namespace visual_odometry {

constexpr auto default_max_features = 2000;   // ORB nfeatures when the caller does not say

struct detection_result {
    std::vector<cv::KeyPoint> keypoints;
    cv::Mat descriptors;   // keypoints.size() x 32, CV_8UC1
};

struct feature_detector_config {
    int max_features{default_max_features};
};

// ORB keypoints + 256-bit descriptors of an 8-bit grayscale image.  Empty image -> empty result.
[[nodiscard]] auto detect_features(cv::Mat const& image, feature_detector_config const& config = {})
    -> detection_result;

// The same keypoints without the blur/descriptor stage.
[[nodiscard]] auto detect_keypoints_only(cv::Mat const& image, feature_detector_config const& config = {})
    -> std::vector<cv::KeyPoint>;

// Colour copy of the image with every keypoint drawn as a circle of its size and a radius at its
// angle (ref:include/visual_odometry/feature_detector.hpp:55, ref:src/feature_detector.cpp:29-35:
// cv::drawKeypoints with DRAW_RICH_KEYPOINTS, green).  Visualisation only, host code; with the
// stand-in types the circles are rasterised here, with OpenCV present the OpenCV call is made.
[[nodiscard]] auto draw_keypoints(cv::Mat const& image, std::vector<cv::KeyPoint> const& keypoints) -> cv::Mat;

// Message of the most recent GPU-side failure on this thread ("" if none).  The three
// hot-path functions keep the reference's no-throw contract: on a device error they return
// the empty / invalid result and leave the reason here.
[[nodiscard]] auto last_error() -> char const*;

}  // namespace visual_odometry
