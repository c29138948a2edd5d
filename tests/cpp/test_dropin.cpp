// The reference's hot-path unit tests (ref:tests/test_feature_detector.cpp,
// ref:tests/test_feature_matcher.cpp, ref:tests/test_motion_estimator.cpp) restated
// against the GPU-backed drop-in headers.  Plain asserts: GoogleTest is not available.
// The checkerboard fixture uses a seeded xorshift noise source where the reference uses
// OpenCV's global RNG (cv::randn), whose stream cannot be reproduced without OpenCV.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include <visual_odometry/feature_detector.hpp>
#include <visual_odometry/feature_matcher.hpp>
#include <visual_odometry/image_matcher.hpp>
#include <visual_odometry/motion_estimator.hpp>

namespace vo = visual_odometry;

static int failures = 0;
#define EXPECT(cond)                                                           \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::printf("FAIL %s:%d  %s  [%s]\n", __FILE__, __LINE__, #cond, vo::last_error()); \
            ++failures;                                                        \
        }                                                                      \
    } while (0)

static cv::Mat checkerboard(int shift, double sigma, unsigned seed) {
    cv::Mat img(480, 640, CV_8UC1);
    std::mt19937 rng(seed);
    std::normal_distribution<double> noise(0.0, sigma);
    for (int y = 0; y < img.rows; ++y)
        for (int x = 0; x < img.cols; ++x) {
            int x2 = (x + shift) % img.cols;
            int v = ((x2 / 32) % 2 == (y / 32) % 2) ? 255 : 0;
            double n = std::max(0.0, std::round(noise(rng)));   // saturating 8-bit noise, as cv::randn into CV_8U
            img.at<unsigned char>(y, x) = (unsigned char)std::min(255.0, v + n);
        }
    return img;
}

static void detector_tests() {
    cv::Mat img = checkerboard(0, 25, 1);
    auto kps = vo::detect_keypoints_only(img, {});
    EXPECT(kps.size() > 0);                                          // DetectsKeypoints
    auto res = vo::detect_features(img, {});
    EXPECT(res.keypoints.size() > 0);                                // DetectsKeypointsAndDescriptors
    EXPECT(!res.descriptors.empty());
    EXPECT((size_t)res.descriptors.rows == res.keypoints.size());
    EXPECT(res.descriptors.cols == 32);
    auto few = vo::detect_keypoints_only(img, vo::feature_detector_config{.max_features = 100});
    EXPECT(few.size() <= 100);                                       // RespectsMaxFeatures
    cv::Mat empty;
    EXPECT(vo::detect_keypoints_only(empty, {}).size() == 0);        // HandlesEmptyImage
    cv::Mat uniform(480, 640, CV_8UC1, cv::Scalar(128));
    EXPECT(vo::detect_keypoints_only(uniform, {}).size() < 10);      // HandlesUniformImage
}

static void matcher_tests() {
    cv::Mat a = checkerboard(0, 15, 2), b = checkerboard(10, 15, 3);
    auto r1 = vo::detect_features(a, {}), r2 = vo::detect_features(b, {});
    auto m = vo::match_features(r1.descriptors, r2.descriptors, r1.keypoints, r2.keypoints, {});
    EXPECT(m.matches.size() > 0);                                    // MatchesSimilarImages
    EXPECT(m.points1.size() == m.matches.size() && m.points2.size() == m.matches.size());
    for (size_t i = 0; i < m.matches.size(); ++i) {                  // MatchedPointsAreConsistent
        auto const& k1 = r1.keypoints[(size_t)m.matches[i].queryIdx];
        auto const& k2 = r2.keypoints[(size_t)m.matches[i].trainIdx];
        EXPECT(m.points1[i] == k1.pt && m.points2[i] == k2.pt);
    }
    auto loose = vo::match_features(r1.descriptors, r2.descriptors, r1.keypoints, r2.keypoints,
                                    vo::feature_matcher_config{.ratio_threshold = 0.9f});
    auto strict = vo::match_features(r1.descriptors, r2.descriptors, r1.keypoints, r2.keypoints,
                                     vo::feature_matcher_config{.ratio_threshold = 0.5f});
    EXPECT(loose.matches.size() >= strict.matches.size());           // StricterRatioReducesMatches
    cv::Mat none;
    EXPECT(vo::match_features(none, r2.descriptors, {}, r2.keypoints, {}).matches.size() == 0);   // HandlesEmptyDescriptors
    EXPECT(vo::match_features(r1.descriptors, none, r1.keypoints, {}, {}).matches.size() == 0);
    vo::orb_matcher om;                                              // matcher_like wrapper
    EXPECT(om.name() == "ORB");
    // the fused pair call gives, element for element, what detect x2 + match give
    auto fused = om.match_images(a, b);
    EXPECT(fused.matches.size() == m.matches.size());
    EXPECT(fused.points1.size() == m.points1.size() && fused.points2.size() == m.points2.size());
    for (size_t i = 0; i < m.matches.size() && i < fused.matches.size(); ++i) {
        EXPECT(fused.matches[i].queryIdx == m.matches[i].queryIdx && fused.matches[i].trainIdx == m.matches[i].trainIdx);
        EXPECT(fused.matches[i].distance == m.matches[i].distance && fused.matches[i].imgIdx == 0);
        EXPECT(fused.points1[i] == m.points1[i] && fused.points2[i] == m.points2[i]);
    }
    cv::Mat none_img;
    EXPECT(om.match_images(none_img, b).matches.size() == 0);      // empty image -> empty result
    cv::Mat small(240, 320, CV_8UC1, cv::Scalar(128));                // different shapes take the three-call route
    EXPECT(om.match_images(a, small).matches.size() == 0);
    bool threw = false;
    try { (void)vo::create_image_matcher("nope"); } catch (std::runtime_error const& e) { threw = std::string(e.what()).find("Unknown matcher") != std::string::npos; }
    EXPECT(threw);
}

struct mat3 { double m[3][3]; };

static void scene(vo::camera_intrinsics const& K, mat3 const& R, double const t[3], int n,
                  std::vector<cv::Point2f>& p1, std::vector<cv::Point2f>& p2) {
    std::mt19937 rng(42);                                            // generate_synthetic_points
    std::uniform_real_distribution<double> dx(100, 1100), dy(50, 300), dz(5, 50);
    for (int i = 0; i < n; ++i) {
        double X[3] = {dx(rng) - K.cx, dy(rng) - K.cy, dz(rng)};
        X[0] *= X[2] / K.fx;
        X[1] *= X[2] / K.fy;
        double Y[3];
        for (int a = 0; a < 3; ++a) Y[a] = R.m[a][0] * X[0] + R.m[a][1] * X[1] + R.m[a][2] * X[2] + t[a];
        p1.emplace_back((float)((K.fx * X[0] + K.cx * X[2]) / X[2]), (float)((K.fy * X[1] + K.cy * X[2]) / X[2]));
        p2.emplace_back((float)((K.fx * Y[0] + K.cx * Y[2]) / Y[2]), (float)((K.fy * Y[1] + K.cy * Y[2]) / Y[2]));
    }
}

static void motion_tests() {
    vo::camera_intrinsics K{.fx = 718.856, .fy = 718.856, .cx = 607.1928, .cy = 185.2157};
    {
        std::vector<cv::Point2f> p1, p2;                             // EstimatesForwardMotion
        mat3 I{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
        double t[3] = {0, 0, 1};
        scene(K, I, t, 100, p1, p2);
        auto r = vo::estimate_motion(p1, p2, K, {});
        EXPECT(r.valid);
        EXPECT(r.inliers > 50);
        EXPECT(r.translation(2) > 0.9);
    }
    {
        std::vector<cv::Point2f> p1, p2;                             // EstimatesRotation
        double a = 0.05;
        mat3 R{{{std::cos(a), 0, std::sin(a)}, {0, 1, 0}, {-std::sin(a), 0, std::cos(a)}}};
        double n = std::sqrt(0.1 * 0.1 + 1.0), t[3] = {0.1 / n, 0, 1.0 / n};
        scene(K, R, t, 100, p1, p2);
        auto r = vo::estimate_motion(p1, p2, K, {});
        EXPECT(r.valid);
        EXPECT(r.inliers > 50);
        EXPECT(std::fabs(r.rotation(0, 2) - std::sin(a)) < 1e-4);   // stronger than the reference: ground truth
    }
    {
        std::vector<cv::Point2f> p1 = {{100, 100}, {200, 200}, {300, 300}};   // FailsWithTooFewPoints
        std::vector<cv::Point2f> p2 = {{110, 100}, {210, 200}, {310, 300}};
        EXPECT(!vo::estimate_motion(p1, p2, K, {}).valid);
    }
    cv::Mat M = K.camera_matrix();                                   // CameraMatrixIsCorrect
    EXPECT(M.at<double>(0, 0) == K.fx && M.at<double>(1, 1) == K.fy && M.at<double>(0, 2) == K.cx &&
           M.at<double>(1, 2) == K.cy && M.at<double>(2, 2) == 1.0);
    auto bad = vo::camera_intrinsics::load_from_yaml("/nonexistent/path.yaml");   // LoadFromYamlReturnsErrorForMissingFile
    EXPECT(!bad.has_value());
    EXPECT(bad.error().find("Could not open") != std::string::npos);
    {
        char const* path = "/tmp/vo_test_camera.yaml";               // LoadsIntrinsicsFromYaml
        FILE* f = std::fopen(path, "w");
        std::fputs("%YAML:1.0\n---\nfx: 718.856\nfy: 718.856\ncx: 607.1928\ncy: 185.2157\n", f);
        std::fclose(f);
        auto okv = vo::camera_intrinsics::load_from_yaml(path);
        EXPECT(okv.has_value());
        EXPECT(std::fabs(okv.value().fx - 718.856) < 1e-3 && std::fabs(okv.value().cy - 185.2157) < 1e-3);
    }
}

int main(int argc, char** argv) {
    bool gpu = !(argc > 1 && std::string(argv[1]) == "--no-gpu");
    if (gpu) {
        detector_tests();
        matcher_tests();
    }
    if (gpu) motion_tests();
    else {
        vo::camera_intrinsics K{.fx = 1, .fy = 2, .cx = 3, .cy = 4};
        EXPECT(K.camera_matrix().at<double>(1, 2) == 4);
        EXPECT(!vo::camera_intrinsics::load_from_yaml("/nonexistent/path.yaml").has_value());
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
