// GPU-backed drop-in for the reference's motion_estimator interface
// (ref:include/visual_odometry/motion_estimator.hpp:17-74).
#pragma once

#include <span>
#include <string>
#include <string_view>

#include <visual_odometry/compat/cv_min.hpp>
#include <visual_odometry/compat/eigen_min.hpp>
#include <visual_odometry/compat/expected_min.hpp>

namespace visual_odometry {

constexpr auto min_motion_inliers = 10;    // fewer pose inliers -> invalid
constexpr auto min_essential_points = 5;   // fewer correspondences -> invalid

struct camera_intrinsics {
    double fx;
    double fy;
    double cx;
    double cy;

    // 3x3 CV_64F [fx 0 cx; 0 fy cy; 0 0 1]
    [[nodiscard]] auto camera_matrix() const -> cv::Mat;

    // Reads fx, fy, cx, cy from an OpenCV-FileStorage style YAML file.
    [[nodiscard]] static auto load_from_yaml(std::string_view filepath)
        -> tl::expected<camera_intrinsics, std::string>;
};

struct motion_estimator_config {
    double ransac_threshold{1.0};      // pixels
    double ransac_confidence{0.999};
};

struct motion_estimate {
    Eigen::Matrix3d rotation;
    Eigen::Vector3d translation;   // unit norm
    int inliers;
    bool valid;
};

// Essential-matrix RANSAC (5-point, Sampson error) + cheirality pose recovery.
[[nodiscard]] auto estimate_motion(std::span<cv::Point2f const> points1,
                                   std::span<cv::Point2f const> points2,
                                   camera_intrinsics const& intrinsics,
                                   motion_estimator_config const& config = {}) -> motion_estimate;

}  // namespace visual_odometry
