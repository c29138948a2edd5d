// Pose accumulation behind the reference's trajectory interface
// (ref:include/visual_odometry/trajectory.hpp:13-101): absolute poses composed from the
// relative motions the GPU path estimates, and the JSON artefact the reference's scripts
// read (ref:src/trajectory.cpp:68-115).  Host-side code: one 3x3 product per frame pair.
#pragma once

#include <cstddef>
#include <filesystem>
#include <string>
#include <vector>

#include <visual_odometry/compat/eigen_min.hpp>
#include <visual_odometry/compat/expected_min.hpp>
#include <visual_odometry/motion_estimator.hpp>

namespace visual_odometry {

// Absolute camera pose in the world frame.
struct pose {
    Eigen::Matrix3d rotation{Eigen::Matrix3d::Identity()};
    Eigen::Vector3d translation{};
    double timestamp{0.0};

    [[nodiscard]] static auto identity(double ts = 0.0) noexcept -> pose;
    // T_world_new = T_world_this * T_relative: R = R_this R_rel, t = R_this t_rel + t_this.
    // The timestamp of the result is 0 (the trajectory sets it).
    [[nodiscard]] auto compose(motion_estimate const& relative) const noexcept -> pose;
};

class Trajectory {
public:
    explicit Trajectory(double initial_timestamp = 0.0);

    // Appends current_pose().compose(motion) stamped `timestamp`; an invalid motion is
    // dropped (returns false, nothing appended).
    auto add_motion(motion_estimate const& motion, double timestamp = 0.0) -> bool;

    [[nodiscard]] auto poses() const noexcept -> std::vector<pose> const&;
    [[nodiscard]] auto current_pose() const noexcept -> pose const&;
    [[nodiscard]] auto size() const noexcept -> size_t;
    [[nodiscard]] auto empty() const noexcept -> bool;   // only the origin
    auto reset() noexcept -> void;

    // {"poses": [{"timestamp": fixed 10 decimals, "rotation": 3x3, "translation": [3]}]},
    // numbers with 10 significant digits.
    [[nodiscard]] auto to_json() const -> std::string;
    [[nodiscard]] auto save_to_json(std::filesystem::path const& filepath) const -> tl::expected<void, std::string>;

private:
    std::vector<pose> poses_;
};

}  // namespace visual_odometry
