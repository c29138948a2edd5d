// Host-only checks of the callers either side of the hot path: trajectory algebra and JSON
// (the cases of ref:tests/test_trajectory.cpp:38-322) and the image loader
// (ref:tests/test_image_loader.cpp).  With a directory argument the loader part decodes every
// image there and prints "<file> <rows> <cols> <fnv1a-64 of the pixels>" for the Python test to
// compare with an independent decoder.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>

#include <vector>

#include <visual_odometry/feature_detector.hpp>
#include <visual_odometry/feature_matcher.hpp>
#include <visual_odometry/image_loader.hpp>
#include <visual_odometry/trajectory.hpp>

namespace vo = visual_odometry;

static int failures = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
            ++failures;                                                        \
        }                                                                      \
    } while (0)

static bool is_identity(Eigen::Matrix3d const& m) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (std::fabs(m(r, c) - (r == c ? 1.0 : 0.0)) > 1e-12) return false;
    return true;
}
static bool is_zero(Eigen::Vector3d const& v) { return std::fabs(v(0)) + std::fabs(v(1)) + std::fabs(v(2)) < 1e-12; }

static vo::motion_estimate motion(Eigen::Matrix3d const& R, Eigen::Vector3d const& t, bool valid = true) {
    vo::motion_estimate m{};
    m.rotation = R;
    m.translation = t;
    m.inliers = 100;
    m.valid = valid;
    return m;
}

static void trajectory_cases() {
    auto const I = Eigen::Matrix3d::Identity();
    {   // starts at the origin
        vo::Trajectory t;
        CHECK(t.size() == 1 && t.empty());
        CHECK(is_identity(t.current_pose().rotation) && is_zero(t.current_pose().translation));
    }
    {   // forward motion accumulates
        vo::Trajectory t;
        CHECK(t.add_motion(motion(I, {0, 0, 1})));
        CHECK(t.size() == 2 && !t.empty());
        CHECK(std::fabs(t.current_pose().translation(2) - 1.0) < 1e-9);
        for (int i = 0; i < 4; ++i) CHECK(t.add_motion(motion(I, {0, 0, 1})));
        CHECK(t.size() == 6 && std::fabs(t.current_pose().translation(2) - 5.0) < 1e-9);
    }
    {   // rotation then translation: forward 1, turn 90 deg about y, forward 1 -> (1, 0, 1)
        vo::Trajectory t;
        Eigen::Matrix3d Ry;
        Ry(0, 0) = 0; Ry(0, 1) = 0; Ry(0, 2) = 1;
        Ry(1, 0) = 0; Ry(1, 1) = 1; Ry(1, 2) = 0;
        Ry(2, 0) = -1; Ry(2, 1) = 0; Ry(2, 2) = 0;
        CHECK(t.add_motion(motion(Ry, {0, 0, 1})));
        CHECK(t.add_motion(motion(I, {0, 0, 1})));
        CHECK(std::fabs(t.current_pose().translation(0) - 1.0) < 1e-6);
        CHECK(std::fabs(t.current_pose().translation(2) - 1.0) < 1e-6);
    }
    {   // invalid motions are dropped
        vo::Trajectory t;
        CHECK(!t.add_motion(motion(I, {0, 0, 1}, false)));
        CHECK(t.size() == 1 && t.empty());
    }
    {   // poses(), reset()
        vo::Trajectory t;
        (void)t.add_motion(motion(I, {0, 0, 1}));
        (void)t.add_motion(motion(I, {0, 0, 1}));
        CHECK(t.poses().size() == 3 && is_zero(t.poses()[0].translation));
        CHECK(std::fabs(t.poses()[1].translation(2) - 1.0) < 1e-9 && std::fabs(t.poses()[2].translation(2) - 2.0) < 1e-9);
        t.reset();
        CHECK(t.size() == 1 && t.empty() && is_zero(t.current_pose().translation));
    }
    {   // pose helpers
        CHECK(vo::pose::identity().timestamp == 0.0 && vo::pose::identity(12.5).timestamp == 12.5);
        vo::pose p;
        p.translation = {1, 2, 3};
        auto q = p.compose(motion(I, {0, 0, 0}));
        CHECK(is_identity(q.rotation) && std::fabs(q.translation(0) - 1) + std::fabs(q.translation(1) - 2) + std::fabs(q.translation(2) - 3) < 1e-12);
    }
    {   // timestamps and JSON
        vo::Trajectory t(1305031102.175304);
        CHECK(t.current_pose().timestamp == 1305031102.175304);
        CHECK(t.add_motion(motion(I, {0, 0, 1}), 1305031102.211214));
        CHECK(t.current_pose().timestamp == 1305031102.211214);
        std::string const js = t.to_json();
        for (char const* key : {"\"poses\"", "\"rotation\"", "\"translation\"", "\"timestamp\"", "1305031102.1753", "1305031102.2112"})
            CHECK(js.find(key) != std::string::npos);
        size_t count = 0;
        for (size_t p = js.find("\"timestamp\""); p != std::string::npos; p = js.find("\"timestamp\"", p + 1)) ++count;
        CHECK(count == 2);
        std::string const expect_head = "{\n  \"poses\": [\n    {\n      \"timestamp\": 1305031102.1753039360,\n      \"rotation\": [\n        [1, 0, 0],\n";
        CHECK(js.compare(0, expect_head.size(), expect_head) == 0);
        CHECK(js.find("      \"translation\": [0, 0, 1]\n    }\n  ]\n}\n") != std::string::npos);
        vo::Trajectory u(1000.0);
        for (int i = 1; i <= 3; ++i) (void)u.add_motion(motion(I, {0, 0, 1}), 1000.0 + i);
        CHECK(u.poses().size() == 4 && u.poses()[3].timestamp == 1003.0);
    }
    {   // save_to_json
        auto path = std::filesystem::temp_directory_path() / "vo_test_trajectory.json";
        vo::Trajectory t;
        (void)t.add_motion(motion(I, {0, 0, 1}));
        auto r = t.save_to_json(path);
        CHECK(r.has_value() && std::filesystem::exists(path));
        std::ifstream f(path);
        std::string content((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        CHECK(content == t.to_json());
        std::filesystem::remove(path);
        auto bad = t.save_to_json("/nonexistent/directory/trajectory.json");
        CHECK(!bad.has_value() && bad.error().find("Failed to open") != std::string::npos);
    }
}

static void loader_cases(std::string const& dir) {
    auto missing = vo::image_loader::create(dir + "/does_not_exist");
    CHECK(!missing.has_value() && missing.error().find("does not exist") != std::string::npos);
    auto loader = vo::image_loader::create(dir);
    CHECK(loader.has_value());
    if (!loader.has_value()) return;
    std::printf("count %zu timestamps %d\n", loader->size(), loader->has_timestamps() ? 1 : 0);
    auto out_of_range = loader->load_image(loader->size());
    CHECK(!out_of_range.has_value() && out_of_range.error().find("out of range") != std::string::npos);
    for (size_t i = 0; i < loader->size(); ++i) {
        auto img = loader->load_image_with_timestamp(i);
        std::string const name = std::filesystem::path(loader->path(i)).filename().string();
        if (!img.has_value()) {
            std::printf("image %s error %s\n", name.c_str(), img.error().c_str());
            continue;
        }
        uint64_t h = 1469598103934665603ull;
        for (int r = 0; r < img->image.rows; ++r)
            for (int c = 0; c < img->image.cols; ++c) h = (h ^ img->image.at<uint8_t>(r, c)) * 1099511628211ull;
        std::printf("image %s %d %d %016llx %.4f\n", name.c_str(), img->image.rows, img->image.cols, (unsigned long long)h,
                    img->timestamp);
    }
    // pair iteration
    size_t pairs = 0;
    loader->reset();
    while (loader->has_next()) {
        auto p = loader->next_pair();
        if (!p.has_value()) break;
        ++pairs;
    }
    std::printf("pairs %zu\n", pairs);
}

// ref:tests/test_feature_detector.cpp:67-78 (DrawsKeypoints), ref:tests/test_feature_matcher.cpp:121-135 (DrawsMatches)
static void draw_cases() {
    namespace vo = visual_odometry;
    cv::Mat a(120, 160, CV_8UC1, cv::Scalar(90)), b(120, 200, CV_8UC1, cv::Scalar(40));
    std::vector<cv::KeyPoint> k1(3), k2(3);
    for (int i = 0; i < 3; ++i) {
        k1[(size_t)i].pt = cv::Point2f(30.f + 40.f * (float)i, 50.f);
        k1[(size_t)i].size = 31.f;
        k1[(size_t)i].angle = 45.f * (float)i;
        k2[(size_t)i].pt = cv::Point2f(20.f + 50.f * (float)i, 70.f);
        k2[(size_t)i].size = 7.f;
    }
    cv::Mat const drawn = vo::draw_keypoints(a, k1);
    CHECK(!drawn.empty() && drawn.channels() == 3 && drawn.rows == a.rows && drawn.cols == a.cols);
    int green = 0;
    for (int y = 0; y < drawn.rows; ++y)
        for (int x = 0; x < drawn.cols; ++x) {
            unsigned char const* p = drawn.ptr<unsigned char>(y) + 3 * x;
            green += p[0] == 0 && p[1] == 255 && p[2] == 0;
        }
    CHECK(green > 100);                                      // three circles of radius 15 and their radii
    CHECK(drawn.ptr<unsigned char>(0)[0] == 90);             // untouched pixels keep the gray value
    std::vector<cv::DMatch> m(2);
    m[0].queryIdx = 0; m[0].trainIdx = 2;
    m[1].queryIdx = 2; m[1].trainIdx = 0;
    cv::Mat const side = vo::draw_feature_matches(a, k1, b, k2, m);
    CHECK(!side.empty() && side.channels() == 3 && side.cols == a.cols + b.cols && side.rows == 120);
    CHECK(vo::draw_keypoints(cv::Mat(), k1).empty());
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--draw") {
        draw_cases();
        std::printf(failures ? "FAILED %d\n" : "ALL PASSED\n", failures);
        return failures ? 1 : 0;
    }
    trajectory_cases();
    if (argc > 1) loader_cases(argv[1]);
    std::printf(failures ? "FAILED %d\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
