// Directory image source and grayscale decoders (interface: include/visual_odometry/image_loader.hpp;
// behaviour of ref:src/image_loader.cpp:31-216 with cv::imread(..., IMREAD_GRAYSCALE) replaced by
// a zlib-based PNG reader, a PGM/PPM reader and the sequential-JPEG luma decoder of vo_jpeg.cpp).
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <unordered_map>

#include <zlib.h>

#include <visual_odometry/image_loader.hpp>

namespace visual_odometry {

namespace detail {
auto decode_jpeg_gray(std::vector<uint8_t> const& file) -> cv::Mat;   // vo_jpeg.cpp
}

namespace {

namespace fs = std::filesystem;

auto read_file(std::string const& path, std::vector<uint8_t>& out) -> bool {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    if (n <= 0) return false;
    f.seekg(0);
    out.resize((size_t)n);
    f.read(reinterpret_cast<char*>(out.data()), n);
    return f.good();
}

inline auto be32(uint8_t const* p) -> uint32_t { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline auto gray_of(int r, int g, int b) -> uint8_t { return (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14); }

// ---- PNG
struct PngInfo {
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    [[nodiscard]] int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
    [[nodiscard]] int bits_per_pixel() const { return channels() * depth; }
};

// Undoes the per-row filter of one (sub-)image in place; rows are stride+1 bytes (filter byte first).
// One loop per filter type (the type is a property of the row, not of the byte); the first
// bpp_bytes bytes of a row have no left neighbour.
auto unfilter(uint8_t* data, size_t rows, size_t stride, int bpp_bytes) -> bool {
    std::vector<uint8_t> zero(stride, 0);
    uint8_t const* prev = zero.data();
    size_t const bpp = (size_t)bpp_bytes, head = std::min(bpp, stride);
    for (size_t y = 0; y < rows; ++y) {
        uint8_t* row = data + y * (stride + 1);
        int const ft = row[0];
        uint8_t* cur = row + 1;
        switch (ft) {
            case 0: break;
            case 1:
                for (size_t i = head; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
                break;
            case 2:
                for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
                break;
            case 3:
                for (size_t i = 0; i < head; ++i) cur[i] = (uint8_t)(cur[i] + (prev[i] >> 1));
                for (size_t i = head; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - bpp] + prev[i]) >> 1));
                break;
            case 4:
                for (size_t i = 0; i < head; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);   // Paeth(0, b, 0) = b
                for (size_t i = head; i < stride; ++i) {
                    int const a = cur[i - bpp], b2 = prev[i], c = prev[i - bpp];
                    int const p = a + b2 - c, pa = std::abs(p - a), pb = std::abs(p - b2), pc = std::abs(p - c);
                    cur[i] = (uint8_t)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b2 : c)));
                }
                break;
            default: return false;
        }
        prev = cur;
    }
    return true;
}

// Gray value of pixel x of an unfiltered row.
inline auto png_gray(PngInfo const& ih, uint8_t const* row, uint32_t x, std::vector<uint8_t> const& palette) -> uint8_t {
    auto sample = [&](int idx) -> int {   // idx-th sample of the row, reduced to 8 bits (index for palettes)
        if (ih.depth == 8) return row[idx];
        if (ih.depth == 16) return row[2 * idx];   // high byte
        int const per_byte = 8 / ih.depth, shift = (per_byte - 1 - idx % per_byte) * ih.depth;
        int const v = (row[idx / per_byte] >> shift) & ((1 << ih.depth) - 1);
        return ih.color == 3 ? v : v * 255 / ((1 << ih.depth) - 1);
    };
    switch (ih.color) {
        case 0: return (uint8_t)sample((int)x);
        case 4: return (uint8_t)sample(2 * (int)x);
        case 2: return gray_of(sample(3 * (int)x), sample(3 * (int)x + 1), sample(3 * (int)x + 2));
        case 6: return gray_of(sample(4 * (int)x), sample(4 * (int)x + 1), sample(4 * (int)x + 2));
        default: {
            size_t const k = (size_t)sample((int)x) * 3;
            return k + 2 < palette.size() ? gray_of(palette[k], palette[k + 1], palette[k + 2]) : 0;
        }
    }
}

auto decode_png(std::vector<uint8_t> const& file) -> cv::Mat {
    static uint8_t const sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return {};
    PngInfo ih;
    std::vector<uint8_t> idat, palette;
    size_t pos = 8;
    bool have_ihdr = false;
    while (pos + 12 <= file.size()) {
        uint32_t const len = be32(&file[pos]);
        uint8_t const* type = &file[pos + 4];
        uint8_t const* body = &file[pos + 8];
        if ((size_t)len > file.size() - pos - 12) return {};
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            ih.w = be32(body);
            ih.h = be32(body + 4);
            ih.depth = body[8];
            ih.color = body[9];
            ih.interlace = body[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    bool const depth_ok = ih.depth == 1 || ih.depth == 2 || ih.depth == 4 || ih.depth == 8 || ih.depth == 16;
    bool const color_ok = ih.color == 0 || ih.color == 2 || ih.color == 3 || ih.color == 4 || ih.color == 6;
    if (!have_ihdr || !depth_ok || !color_ok || ih.w == 0 || ih.h == 0 || ih.w > 32768 || ih.h > 32768 || idat.empty())
        return {};
    if ((ih.color == 2 || ih.color == 4 || ih.color == 6) && ih.depth < 8) return {};
    if (ih.color == 3 && (ih.depth == 16 || palette.empty())) return {};

    // (sub-)images: the whole picture, or the seven Adam7 passes
    struct Pass { uint32_t x0, y0, dx, dy; };
    static Pass const adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static Pass const whole[1] = {{0, 0, 1, 1}};
    Pass const* passes = ih.interlace ? adam7 : whole;
    int const n_pass = ih.interlace ? 7 : 1;
    int const bpp = ih.bits_per_pixel(), bpp_bytes = std::max(1, bpp / 8);
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; ++k) {
        uint32_t const pw = (ih.w - passes[k].x0 + passes[k].dx - 1) / passes[k].dx;
        uint32_t const ph = (ih.h - passes[k].y0 + passes[k].dy - 1) / passes[k].dy;
        if (ih.w <= passes[k].x0 || ih.h <= passes[k].y0 || pw == 0 || ph == 0) continue;
        raw_size += (size_t)ph * (((size_t)pw * bpp + 7) / 8 + 1);
    }
    // deflate expands by at most 1032:1, so a header that promises more pixels than the compressed
    // stream can hold is a broken (or hostile) file: refuse it before allocating anything
    if (raw_size > idat.size() * 1032 + 64) return {};
    std::vector<uint8_t> raw;
    try {
        raw.resize(raw_size);
    } catch (std::bad_alloc const&) {
        return {};
    }
    uLongf got = (uLongf)raw_size;
    int const zrc = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || got != raw_size) return {};

    cv::Mat img((int)ih.h, (int)ih.w, CV_8UC1);
    size_t off = 0;
    for (int k = 0; k < n_pass; ++k) {
        Pass const& ps = passes[k];
        if (ih.w <= ps.x0 || ih.h <= ps.y0) continue;
        uint32_t const pw = (ih.w - ps.x0 + ps.dx - 1) / ps.dx, ph = (ih.h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw == 0 || ph == 0) continue;
        size_t const stride = ((size_t)pw * bpp + 7) / 8;
        if (!unfilter(raw.data() + off, ph, stride, bpp_bytes)) return {};
        for (uint32_t y = 0; y < ph; ++y) {
            uint8_t const* row = raw.data() + off + (size_t)y * (stride + 1) + 1;
            uint8_t* out = img.ptr<uint8_t>((int)(ps.y0 + y * ps.dy));
            if (ih.color == 0 && ih.depth == 8 && ps.dx == 1) {   // the common case: the row is the pixels
                std::memcpy(out, row, pw);
                continue;
            }
            for (uint32_t x = 0; x < pw; ++x) out[ps.x0 + x * ps.dx] = png_gray(ih, row, x, palette);
        }
        off += (size_t)ph * (stride + 1);
    }
    return img;
}

// ---- binary PGM (P5) / PPM (P6), 8 or 16 bits per sample
auto decode_pnm(std::vector<uint8_t> const& file) -> cv::Mat {
    if (file.size() < 7 || file[0] != 'P' || (file[1] != '5' && file[1] != '6')) return {};
    int const channels = file[1] == '5' ? 1 : 3;
    size_t pos = 2;
    auto next_int = [&]() -> long {
        for (;;) {
            while (pos < file.size() && std::isspace(file[pos])) ++pos;
            if (pos < file.size() && file[pos] == '#') {
                while (pos < file.size() && file[pos] != '\n') ++pos;
                continue;
            }
            break;
        }
        long v = -1;
        while (pos < file.size() && std::isdigit(file[pos])) {
            v = (v < 0 ? 0 : v) * 10 + (file[pos] - '0');
            ++pos;
        }
        return v;
    };
    long const w = next_int(), h = next_int(), maxv = next_int();
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || maxv <= 0 || maxv > 65535 || pos >= file.size()) return {};
    ++pos;   // the single whitespace after maxval
    int const bytes = maxv > 255 ? 2 : 1;
    if (file.size() - pos < (size_t)w * h * channels * bytes) return {};
    cv::Mat img((int)h, (int)w, CV_8UC1);
    uint8_t const* p = &file[pos];
    auto sample = [&](size_t idx) -> int {
        int v = bytes == 1 ? p[idx] : ((p[2 * idx] << 8) | p[2 * idx + 1]);
        return maxv == 255 ? v : (int)((long)v * 255 / maxv);
    };
    for (long y = 0; y < h; ++y) {
        uint8_t* out = img.ptr<uint8_t>((int)y);
        for (long x = 0; x < w; ++x) {
            size_t const i = ((size_t)y * w + x) * channels;
            out[x] = channels == 1 ? (uint8_t)sample(i) : gray_of(sample(i), sample(i + 1), sample(i + 2));
        }
    }
    return img;
}

// "timestamp filename" lines of a TUM rgb.txt ('#' comments) -> absolute path -> timestamp
auto parse_rgb_txt(fs::path const& file_path, fs::path const& base) -> std::unordered_map<std::string, double> {
    std::unordered_map<std::string, double> stamps;
    std::ifstream in(file_path);
    std::string line;
    while (in.is_open() && std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream fields(line);
        double ts = 0.0;
        std::string name;
        if (fields >> ts >> name) stamps[(base / name).string()] = ts;
    }
    return stamps;
}

}  // namespace

auto read_gray_image(std::string const& path) -> cv::Mat {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return {};
    if (file.size() >= 2 && file[0] == 'P') return decode_pnm(file);
    if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) return detail::decode_jpeg_gray(file);
    return decode_png(file);   // anything else comes back empty
}

image_loader::image_loader(std::string image_directory, std::vector<std::string> image_paths, std::vector<double> timestamps)
    : image_directory_(std::move(image_directory)), image_paths_(std::move(image_paths)), timestamps_(std::move(timestamps)) {}

auto image_loader::create(std::string_view image_directory) -> tl::expected<image_loader, std::string> {
    if (!fs::exists(image_directory)) return tl::unexpected("Image directory does not exist: " + std::string(image_directory));
    fs::path const dir(image_directory), parent = dir.parent_path();   // TUM layout: parent/rgb.txt, parent/rgb/*.png
    std::unordered_map<std::string, double> stamps;
    if (fs::exists(parent / "rgb.txt")) stamps = parse_rgb_txt(parent / "rgb.txt", parent);

    std::vector<std::string> paths;
    for (auto const& entry : fs::directory_iterator(dir)) {
        if (!entry.is_regular_file()) continue;
        std::string ext = entry.path().extension().string();
        for (char& ch : ext) ch = (char)std::tolower((unsigned char)ch);
        if (ext == ".png" || ext == ".jpg" || ext == ".jpeg") paths.push_back(entry.path().string());
    }
    std::sort(paths.begin(), paths.end());
    std::vector<double> ts;
    ts.reserve(paths.size());
    for (auto const& p : paths) {
        auto it = stamps.find(p);
        ts.push_back(it == stamps.end() ? 0.0 : it->second);
    }
    return image_loader(std::string(image_directory), std::move(paths), std::move(ts));
}

auto image_loader::load_image(size_t index) const -> tl::expected<cv::Mat, std::string> {
    if (index >= image_paths_.size())
        return tl::unexpected("Image index out of range: " + std::to_string(index) + " >= " + std::to_string(image_paths_.size()));
    cv::Mat img = read_gray_image(image_paths_[index]);
    if (img.empty()) return tl::unexpected("Failed to load image: " + image_paths_[index]);
    return img;
}

auto image_loader::load_image_with_timestamp(size_t index) const -> tl::expected<timestamped_image, std::string> {
    auto img = load_image(index);
    if (!img.has_value()) return tl::unexpected(img.error());
    return timestamped_image{img.value(), get_timestamp(index)};
}

auto image_loader::load_image_pair(size_t index) const -> tl::expected<std::pair<cv::Mat, cv::Mat>, std::string> {
    auto a = load_image(index);
    if (!a.has_value()) return tl::unexpected(a.error());
    auto b = load_image(index + 1);
    if (!b.has_value()) return tl::unexpected(b.error());
    return std::make_pair(a.value(), b.value());
}

auto image_loader::load_image_pair_with_timestamps(size_t index) const -> tl::expected<timestamped_image_pair, std::string> {
    auto a = load_image_with_timestamp(index);
    if (!a.has_value()) return tl::unexpected(a.error());
    auto b = load_image_with_timestamp(index + 1);
    if (!b.has_value()) return tl::unexpected(b.error());
    return timestamped_image_pair{a.value(), b.value()};
}

auto image_loader::next_pair() -> tl::expected<std::pair<cv::Mat, cv::Mat>, std::string> {
    auto pair = load_image_pair(current_index_);
    if (pair.has_value()) ++current_index_;
    return pair;
}

auto image_loader::next_pair_with_timestamps() -> tl::expected<timestamped_image_pair, std::string> {
    auto pair = load_image_pair_with_timestamps(current_index_);
    if (pair.has_value()) ++current_index_;
    return pair;
}

auto image_loader::has_next() const noexcept -> bool { return current_index_ + 1 < image_paths_.size(); }
auto image_loader::reset() noexcept -> void { current_index_ = 0; }
auto image_loader::size() const noexcept -> size_t { return image_paths_.size(); }
auto image_loader::get_timestamp(size_t index) const noexcept -> double { return index < timestamps_.size() ? timestamps_[index] : 0.0; }

auto image_loader::has_timestamps() const noexcept -> bool {
    return std::any_of(timestamps_.begin(), timestamps_.end(), [](double t) { return t != 0.0; });
}

auto image_loader::path(size_t index) const -> std::string const& { return image_paths_.at(index); }

}  // namespace visual_odometry
