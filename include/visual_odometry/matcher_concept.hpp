// ref:include/visual_odometry/matcher_concept.hpp:21-25
#pragma once

#include <concepts>
#include <string_view>

#include <visual_odometry/feature_matcher.hpp>

namespace visual_odometry {

template <typename T>
concept matcher_like = requires(T m, cv::Mat const& img) {
    { m.match_images(img, img) } -> std::same_as<match_result>;
    { m.name() } -> std::convertible_to<std::string_view>;
};

}  // namespace visual_odometry
