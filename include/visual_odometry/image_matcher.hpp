// GPU-backed drop-in for the ORB half of the reference's image_matcher interface
// (ref:include/visual_odometry/image_matcher.hpp:30-62,108-116).  The LightGlue/ONNX
// alternative of the reference's variant is a different path and is not provided.
#pragma once

#include <string_view>
#include <variant>

#include <visual_odometry/feature_detector.hpp>
#include <visual_odometry/feature_matcher.hpp>
#include <visual_odometry/matcher_concept.hpp>

namespace visual_odometry {

// detect_features x2 + match_features, with both images' features kept in HBM in between.
[[nodiscard]] auto match_images_orb(cv::Mat const& img1, cv::Mat const& img2,
                                    feature_detector_config const& detector_config = {},
                                    feature_matcher_config const& matcher_config = {}) -> match_result;

struct orb_matcher {
    explicit orb_matcher(int max_features = default_max_features, float ratio_threshold = default_ratio_threshold)
        : detector_config_{.max_features = max_features}, matcher_config_{.ratio_threshold = ratio_threshold} {}

    [[nodiscard]] auto match_images(cv::Mat const& img1, cv::Mat const& img2) const -> match_result {
        return match_images_orb(img1, img2, detector_config_, matcher_config_);
    }
    [[nodiscard]] auto name() const noexcept -> std::string_view { return "ORB"; }

private:
    feature_detector_config detector_config_;
    feature_matcher_config matcher_config_;
};

static_assert(matcher_like<orb_matcher>);

using image_matcher = std::variant<orb_matcher>;

// "orb" or "" -> orb_matcher; anything else throws std::runtime_error("Unknown matcher: ...").
[[nodiscard]] auto create_image_matcher(std::string_view name) -> image_matcher;

}  // namespace visual_odometry
