// Minimal stand-ins for the handful of OpenCV types the visual_odometry
// interfaces mention (cv::Mat, cv::KeyPoint, cv::DMatch, cv::Point2f).  Used
// only when the real <opencv2/core.hpp> is not on the include path; with
// OpenCV present the real types are used and the GPU library fills them.
#pragma once
#if __has_include(<opencv2/core.hpp>) && !defined(VO_FORCE_COMPAT_TYPES)
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#else
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define VO_COMPAT_CV_TYPES 1
#ifndef CV_8U
#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#endif

namespace cv {

using uchar = unsigned char;

struct Point2f {
    float x{0}, y{0};
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
    bool operator==(Point2f const& o) const { return x == o.x && y == o.y; }
};

struct KeyPoint {
    Point2f pt;
    float size{0};
    float angle{-1};
    float response{0};
    int octave{0};
    int class_id{-1};
};

struct DMatch {
    int queryIdx{-1};
    int trainIdx{-1};
    int imgIdx{-1};
    float distance{0};
};

struct Scalar {
    double v[4]{0, 0, 0, 0};
    Scalar() = default;
    Scalar(double a) { v[0] = a; }
};

// Owning, reference-counted 2-D array (continuous rows).
class Mat {
public:
    int rows{0}, cols{0};
    size_t step{0};
    uchar* data{nullptr};

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, Scalar const& s) {
        create(r, c, type);
        if (depth() == CV_8U) std::memset(data, (int)s.v[0], total_bytes());
        else if (depth() == CV_64F) for (size_t i = 0; i < total_bytes() / 8; ++i) reinterpret_cast<double*>(data)[i] = s.v[0];
        else if (depth() == CV_32F) for (size_t i = 0; i < total_bytes() / 4; ++i) reinterpret_cast<float*>(data)[i] = (float)s.v[0];
    }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elemSize();
        buf_ = std::shared_ptr<uchar[]>(new uchar[std::max<size_t>(total_bytes(), 1)]());
        data = buf_.get();
    }
    [[nodiscard]] bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    [[nodiscard]] int type() const { return type_; }
    [[nodiscard]] int depth() const { return type_ & 7; }
    [[nodiscard]] int channels() const { return (type_ >> 3) + 1; }
    [[nodiscard]] size_t elemSize() const {
        static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        return (size_t)sz[depth()] * channels();
    }
    [[nodiscard]] size_t total_bytes() const { return step * (size_t)rows; }
    [[nodiscard]] bool isContinuous() const { return true; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * (size_t)r); }
    template <typename T> T const* ptr(int r = 0) const { return reinterpret_cast<T const*>(data + step * (size_t)r); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> T const& at(int r, int c) const { return ptr<T>(r)[c]; }

private:
    int type_{0};
    std::shared_ptr<uchar[]> buf_;
};

}  // namespace cv
#endif
