// draw_keypoints / draw_feature_matches of the reference's detector and matcher interfaces
// (ref:src/feature_detector.cpp:29-35, ref:src/feature_matcher.cpp:44-55): visualisation helpers
// outside the hot path, host code.  With real OpenCV they are the same two OpenCV calls the
// reference makes; with the stand-in types the marks are rasterised here (BGR, green).
#include <cmath>
#include <cstdlib>

#include <visual_odometry/feature_detector.hpp>
#include <visual_odometry/feature_matcher.hpp>

namespace visual_odometry {

#ifdef VO_COMPAT_CV_TYPES
namespace {

void put(cv::Mat& m, int x, int y, unsigned char b, unsigned char g, unsigned char r) {
    if (x < 0 || y < 0 || x >= m.cols || y >= m.rows) return;
    unsigned char* p = m.ptr<unsigned char>(y) + 3 * x;
    p[0] = b; p[1] = g; p[2] = r;
}

void line(cv::Mat& m, int x0, int y0, int x1, int y1, unsigned char b, unsigned char g, unsigned char r) {
    int dx = std::abs(x1 - x0), dy = -std::abs(y1 - y0), sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1, err = dx + dy;
    for (;;) {
        put(m, x0, y0, b, g, r);
        if (x0 == x1 && y0 == y1) break;
        int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
    }
}

void circle(cv::Mat& m, int cx, int cy, int rad, unsigned char b, unsigned char g, unsigned char r) {
    int x = rad, y = 0, err = 1 - rad;   // midpoint circle
    while (x >= y) {
        int const px[8] = {x, y, -y, -x, -x, -y, y, x}, py[8] = {y, x, x, y, -y, -x, -x, -y};
        for (int k = 0; k < 8; ++k) put(m, cx + px[k], cy + py[k], b, g, r);
        ++y;
        if (err < 0) err += 2 * y + 1;
        else { --x; err += 2 * (y - x) + 1; }
    }
}

// gray (or already 3-channel) image copied into `out` at column offset x0
void blit(cv::Mat const& src, cv::Mat& out, int x0) {
    for (int y = 0; y < src.rows; ++y) {
        unsigned char const* s = src.ptr<unsigned char>(y);
        unsigned char* d = out.ptr<unsigned char>(y) + 3 * x0;
        for (int x = 0; x < src.cols; ++x)
            for (int ch = 0; ch < 3; ++ch) d[3 * x + ch] = src.channels() == 3 ? s[3 * x + ch] : s[x];
    }
}

}  // namespace

auto draw_keypoints(cv::Mat const& image, std::vector<cv::KeyPoint> const& keypoints) -> cv::Mat {
    cv::Mat out;
    if (image.empty()) return out;
    out.create(image.rows, image.cols, CV_8UC3);
    blit(image, out, 0);
    for (auto const& k : keypoints) {
        int const cx = (int)std::lround(k.pt.x), cy = (int)std::lround(k.pt.y);
        int const rad = std::max(1, (int)std::lround(k.size / 2));
        circle(out, cx, cy, rad, 0, 255, 0);
        if (k.angle >= 0) {   // the radius at the keypoint's orientation
            double const a = k.angle * 3.14159265358979323846 / 180.0;
            line(out, cx, cy, cx + (int)std::lround(rad * std::cos(a)), cy + (int)std::lround(rad * std::sin(a)), 0, 255, 0);
        }
    }
    return out;
}

auto draw_feature_matches(cv::Mat const& image1, std::vector<cv::KeyPoint> const& keypoints1, cv::Mat const& image2,
                          std::vector<cv::KeyPoint> const& keypoints2, std::vector<cv::DMatch> const& matches) -> cv::Mat {
    cv::Mat out;
    if (image1.empty() || image2.empty()) return out;
    out.create(std::max(image1.rows, image2.rows), image1.cols + image2.cols, CV_8UC3);
    blit(image1, out, 0);
    blit(image2, out, image1.cols);
    for (auto const& m : matches) {
        if (m.queryIdx < 0 || m.trainIdx < 0 || (size_t)m.queryIdx >= keypoints1.size() ||
            (size_t)m.trainIdx >= keypoints2.size())
            continue;
        auto const& a = keypoints1[(size_t)m.queryIdx].pt;
        auto const& b = keypoints2[(size_t)m.trainIdx].pt;
        int const ax = (int)std::lround(a.x), ay = (int)std::lround(a.y);
        int const bx = (int)std::lround(b.x) + image1.cols, by = (int)std::lround(b.y);
        circle(out, ax, ay, 3, 0, 255, 0);
        circle(out, bx, by, 3, 0, 255, 0);
        line(out, ax, ay, bx, by, 0, 255, 0);
    }
    return out;
}

#else  // real OpenCV: the reference's own two calls

auto draw_keypoints(cv::Mat const& image, std::vector<cv::KeyPoint> const& keypoints) -> cv::Mat {
    cv::Mat output;
    cv::drawKeypoints(image, keypoints, output, cv::Scalar(0, 255, 0), cv::DrawMatchesFlags::DRAW_RICH_KEYPOINTS);
    return output;
}

auto draw_feature_matches(cv::Mat const& image1, std::vector<cv::KeyPoint> const& keypoints1, cv::Mat const& image2,
                          std::vector<cv::KeyPoint> const& keypoints2, std::vector<cv::DMatch> const& matches) -> cv::Mat {
    cv::Mat output;
    cv::drawMatches(image1, keypoints1, image2, keypoints2, matches, output, cv::Scalar(0, 255, 0), cv::Scalar(255, 0, 0),
                    std::vector<char>(), cv::DrawMatchesFlags::NOT_DRAW_SINGLE_POINTS);
    return output;
}

#endif

}  // namespace visual_odometry
