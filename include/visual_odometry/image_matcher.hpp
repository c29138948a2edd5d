// GPU-backed drop-in for the ORB half of the reference's image_matcher interface
// (ref:include/visual_odometry/image_matcher.hpp:30-62,108-116).  The LightGlue/ONNX
// alternative of the reference's variant is a different path and is not implemented here:
// a build that has OpenCV and ONNX Runtime keeps the reference's own lightglue half of
// src/image_matcher.cpp and defines VO_WITH_LIGHTGLUE, which restores the declaration and the
// two-alternative variant below (the nanobind module binds lightglue_matcher,
// ref:src/python_bindings.cpp:213, and needs it).
#pragma once

#include <filesystem>
#include <memory>
#include <string_view>
#include <variant>

#include <visual_odometry/feature_detector.hpp>
#include <visual_odometry/feature_matcher.hpp>
#include <visual_odometry/matcher_concept.hpp>

namespace visual_odometry {

#ifdef VO_WITH_LIGHTGLUE
class onnx_session;
#endif

// detect_features x2 + match_features as one GPU round trip (vo_match_images): the two images go
// up once, the descriptors never leave HBM, both keypoint sets and the matches come back once.
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

#ifdef VO_WITH_LIGHTGLUE
// Declaration only (ref:include/visual_odometry/image_matcher.hpp:71-95); the definition and
// create_image_matcher stay in the reference's src/image_matcher.cpp, built with
// -DVO_WITH_LIGHTGLUE against these headers (libvisual_odometry_hip.so then leaves
// create_image_matcher out, see csrc/host/vo_cpp_api.cpp).
struct lightglue_matcher {
    explicit lightglue_matcher(std::filesystem::path model_path = "models/disk_lightglue_end2end.onnx");
    lightglue_matcher(lightglue_matcher const&) = delete;
    auto operator=(lightglue_matcher const&) -> lightglue_matcher& = delete;
    lightglue_matcher(lightglue_matcher&&) noexcept;
    auto operator=(lightglue_matcher&&) noexcept -> lightglue_matcher&;
    ~lightglue_matcher();
    [[nodiscard]] auto match_images(cv::Mat const& img1, cv::Mat const& img2) const -> match_result;
    [[nodiscard]] auto name() const noexcept -> std::string_view { return "LightGlue"; }

private:
    std::filesystem::path model_path_;
    std::unique_ptr<onnx_session> session_;
};
static_assert(matcher_like<lightglue_matcher>);
using image_matcher = std::variant<orb_matcher, lightglue_matcher>;
#else
using image_matcher = std::variant<orb_matcher>;
#endif

// "orb" or "" -> orb_matcher ("lightglue" with VO_WITH_LIGHTGLUE); anything else throws
// std::runtime_error("Unknown matcher: ...").
[[nodiscard]] auto create_image_matcher(std::string_view name) -> image_matcher;

}  // namespace visual_odometry
