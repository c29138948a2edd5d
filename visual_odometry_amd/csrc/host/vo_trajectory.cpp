// Trajectory accumulation and its JSON form (interface: include/visual_odometry/trajectory.hpp;
// behaviour of ref:src/trajectory.cpp:16-130).
#include <cstdio>
#include <fstream>

#include <visual_odometry/trajectory.hpp>

namespace visual_odometry {

auto pose::identity(double ts) noexcept -> pose {
    pose p;
    p.timestamp = ts;
    return p;
}

auto pose::compose(motion_estimate const& relative) const noexcept -> pose {
    pose out;
    out.rotation = rotation * relative.rotation;
    out.translation = rotation * relative.translation + translation;
    return out;
}

Trajectory::Trajectory(double initial_timestamp) { poses_.push_back(pose::identity(initial_timestamp)); }

auto Trajectory::add_motion(motion_estimate const& motion, double timestamp) -> bool {
    if (!motion.valid) return false;
    pose next = poses_.back().compose(motion);
    next.timestamp = timestamp;
    poses_.push_back(next);
    return true;
}

auto Trajectory::poses() const noexcept -> std::vector<pose> const& { return poses_; }
auto Trajectory::current_pose() const noexcept -> pose const& { return poses_.back(); }
auto Trajectory::size() const noexcept -> size_t { return poses_.size(); }
auto Trajectory::empty() const noexcept -> bool { return poses_.size() <= 1; }

auto Trajectory::reset() noexcept -> void {
    poses_.clear();
    poses_.push_back(pose::identity());
}

namespace {
// ten significant digits, shortest form (what an ostream prints at precision 10)
void put_number(std::string& out, double v) {
    char buf[40];
    std::snprintf(buf, sizeof buf, "%.10g", v);
    out += buf;
}
}  // namespace

auto Trajectory::to_json() const -> std::string {
    std::string js = "{\n  \"poses\": [\n";
    for (size_t i = 0; i < poses_.size(); ++i) {
        pose const& p = poses_[i];
        char ts[64];
        std::snprintf(ts, sizeof ts, "%.10f", p.timestamp);   // fixed: epoch seconds keep their fraction
        js += "    {\n      \"timestamp\": ";
        js += ts;
        js += ",\n      \"rotation\": [\n";
        for (int r = 0; r < 3; ++r) {
            js += "        [";
            for (int c = 0; c < 3; ++c) {
                put_number(js, p.rotation(r, c));
                if (c < 2) js += ", ";
            }
            js += r < 2 ? "],\n" : "]\n";
        }
        js += "      ],\n      \"translation\": [";
        for (int k = 0; k < 3; ++k) {
            put_number(js, p.translation(k));
            if (k < 2) js += ", ";
        }
        js += "]\n    }";
        js += i + 1 < poses_.size() ? ",\n" : "\n";
    }
    js += "  ]\n}\n";
    return js;
}

auto Trajectory::save_to_json(std::filesystem::path const& filepath) const -> tl::expected<void, std::string> {
    std::ofstream file(filepath);
    if (!file.is_open()) return tl::unexpected("Failed to open file for writing: " + filepath.string());
    file << to_json();
    if (!file.good()) return tl::unexpected("Error writing to file: " + filepath.string());
    return {};
}

}  // namespace visual_odometry
