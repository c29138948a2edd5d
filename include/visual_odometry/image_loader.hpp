// Sequential image source behind the reference's image_loader interface
// (ref:include/visual_odometry/image_loader.hpp:17-131, ref:src/image_loader.cpp:58-216):
// sorted directory scan of .png/.jpg/.jpeg files, TUM rgb.txt timestamps from the parent
// directory, grayscale decode.  No OpenCV here: PNG (8/16-bit, gray/RGB/palette/alpha,
// non-interlaced and Adam7) is decoded with zlib, binary PGM/PPM is accepted under any of the
// scanned extensions, JPEG files fail with the reference's "Failed to load image" error.
#pragma once

#include <cstddef>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include <visual_odometry/compat/cv_min.hpp>
#include <visual_odometry/compat/expected_min.hpp>

namespace visual_odometry {

struct timestamped_image {
    cv::Mat image;
    double timestamp{0.0};
};

struct timestamped_image_pair {
    timestamped_image first;
    timestamped_image second;
};

class image_loader {
public:
    [[nodiscard]] static auto create(std::string_view image_directory) -> tl::expected<image_loader, std::string>;

    [[nodiscard]] auto load_image(size_t index) const -> tl::expected<cv::Mat, std::string>;
    [[nodiscard]] auto load_image_with_timestamp(size_t index) const -> tl::expected<timestamped_image, std::string>;
    [[nodiscard]] auto load_image_pair(size_t index) const -> tl::expected<std::pair<cv::Mat, cv::Mat>, std::string>;
    [[nodiscard]] auto load_image_pair_with_timestamps(size_t index) const
        -> tl::expected<timestamped_image_pair, std::string>;
    [[nodiscard]] auto next_pair() -> tl::expected<std::pair<cv::Mat, cv::Mat>, std::string>;
    [[nodiscard]] auto next_pair_with_timestamps() -> tl::expected<timestamped_image_pair, std::string>;
    [[nodiscard]] auto has_next() const noexcept -> bool;
    auto reset() noexcept -> void;
    [[nodiscard]] auto size() const noexcept -> size_t;
    [[nodiscard]] auto get_timestamp(size_t index) const noexcept -> double;   // 0.0 if unknown
    [[nodiscard]] auto has_timestamps() const noexcept -> bool;
    [[nodiscard]] auto path(size_t index) const -> std::string const&;

private:
    image_loader(std::string image_directory, std::vector<std::string> image_paths, std::vector<double> timestamps);

    std::string image_directory_;
    std::vector<std::string> image_paths_;
    std::vector<double> timestamps_;
    size_t current_index_{0};
};

// Decodes one image file to 8-bit grayscale (empty Mat on failure).  Colour is reduced with
// the BT.601 fixed-point weights (R*4899 + G*9617 + B*1868 + 8192) >> 14.
[[nodiscard]] auto read_gray_image(std::string const& path) -> cv::Mat;

}  // namespace visual_odometry
