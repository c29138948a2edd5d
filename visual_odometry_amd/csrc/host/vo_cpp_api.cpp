// C++20 shim: the reference's detect_features / match_features /
// estimate_motion / match_images_orb signatures implemented over the C ABI of
// libvo_hip.so.  One lazily created GPU context per host thread.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <visual_odometry/feature_detector.hpp>
#include <visual_odometry/feature_matcher.hpp>
#include <visual_odometry/image_matcher.hpp>
#include <visual_odometry/motion_estimator.hpp>
#include <visual_odometry/sequence_pipeline.hpp>

#include "../../../include/vo_hip.h"

namespace visual_odometry {
namespace {

struct thread_ctx {
    vo_ctx* c{nullptr};
    std::string err;
    // reusable result buffers of match_images_orb
    std::vector<vo_keypoint> k1, k2;
    std::vector<int32_t> qi, ti;
    std::vector<float> dist;
    ~thread_ctx() {
        if (c) vo_ctx_destroy(c);
    }
};

thread_ctx& tls() {
    thread_local thread_ctx t;
    return t;
}

// Context for this thread, or nullptr (reason in last_error()).
vo_ctx* ctx() {
    auto& t = tls();
    if (!t.c) {
        int dev = 0;
        if (char const* e = std::getenv("VO_HIP_DEVICE")) dev = std::atoi(e);
        int rc = vo_ctx_create(dev, &t.c);
        if (rc != VO_OK) {
            t.c = nullptr;
            t.err = "vo_ctx_create failed with code " + std::to_string(rc);
        }
    }
    return t.c;
}

// VO_ERR_CAPACITY from a host-buffer call means that a keypoint list was cut at a table's end
// (thousands of tied scores): the outputs are still written, so the caller gets them - the
// reference returns its (larger) lists in that case too - and the message is kept in last_error().
bool ok(int rc) {
    auto& t = tls();
    if (rc == VO_OK) return true;
    t.err = t.c ? vo_last_error(t.c) : "no GPU context";
    return rc == VO_ERR_CAPACITY;
}

static_assert(sizeof(vo_keypoint) == 28);

void to_cv(vo_keypoint const* k, int n, std::vector<cv::KeyPoint>& out) {
    out.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        cv::KeyPoint& o = out[(size_t)i];
        o.pt.x = k[i].x;
        o.pt.y = k[i].y;
        o.size = k[i].size;
        o.angle = k[i].angle;
        o.response = k[i].response;
        o.octave = k[i].octave;
        o.class_id = k[i].class_id;
    }
}

int detect(cv::Mat const& image, int max_features, bool want_desc, std::vector<cv::KeyPoint>& kps, cv::Mat& desc) {
    kps.clear();
    if (image.empty()) return 0;
    vo_ctx* c = ctx();
    if (!c) return -1;
    int cap = max_features + 8 * 256;
    std::vector<vo_keypoint> k((size_t)cap);
    std::vector<uint8_t> d(want_desc ? (size_t)cap * 32 : 0);
    int n = 0;
    if (!ok(vo_orb_detect(c, image.data, image.rows, image.cols, (int)image.step, max_features, k.data(),
                          want_desc ? d.data() : nullptr, cap, &n)))
        return -1;
    to_cv(k.data(), n, kps);
    if (want_desc && n > 0) {
        desc.create(n, 32, CV_8UC1);
        for (int i = 0; i < n; ++i) std::memcpy(desc.ptr<uint8_t>(i), &d[(size_t)i * 32], 32);
    }
    return n;
}

}  // namespace

auto last_error() -> char const* { return tls().err.c_str(); }

auto detect_features(cv::Mat const& image, feature_detector_config const& config) -> detection_result {
    detection_result r;
    detect(image, config.max_features, true, r.keypoints, r.descriptors);
    return r;
}

auto detect_keypoints_only(cv::Mat const& image, feature_detector_config const& config) -> std::vector<cv::KeyPoint> {
    std::vector<cv::KeyPoint> k;
    cv::Mat none;
    detect(image, config.max_features, false, k, none);
    return k;
}

auto match_features(cv::Mat const& descriptors1, cv::Mat const& descriptors2, std::span<cv::KeyPoint const> keypoints1,
                    std::span<cv::KeyPoint const> keypoints2, feature_matcher_config const& config) -> match_result {
    match_result r;
    if (descriptors1.empty() || descriptors2.empty()) return r;
    vo_ctx* c = ctx();
    if (!c) return r;
    int nq = descriptors1.rows, nt = descriptors2.rows;
    auto pack = [](cv::Mat const& m) {
        std::vector<uint8_t> v((size_t)m.rows * 32);
        for (int i = 0; i < m.rows; ++i) std::memcpy(&v[(size_t)i * 32], m.ptr<uint8_t>(i), 32);
        return v;
    };
    std::vector<uint8_t> q = pack(descriptors1), t = pack(descriptors2);
    std::vector<int32_t> qi((size_t)nq), ti((size_t)nq);
    std::vector<float> d((size_t)nq);
    int m = 0;
    if (!ok(vo_match_ratio(c, q.data(), nq, t.data(), nt, config.ratio_threshold, qi.data(), ti.data(), d.data(), &m)))
        return r;
    r.matches.reserve((size_t)m);
    for (int i = 0; i < m; ++i) {
        cv::DMatch dm;
        dm.queryIdx = qi[(size_t)i];
        dm.trainIdx = ti[(size_t)i];
        dm.imgIdx = 0;
        dm.distance = d[(size_t)i];
        r.matches.push_back(dm);
        r.points1.push_back(keypoints1[(size_t)dm.queryIdx].pt);
        r.points2.push_back(keypoints2[(size_t)dm.trainIdx].pt);
    }
    return r;
}

auto match_images_orb(cv::Mat const& img1, cv::Mat const& img2, feature_detector_config const& detector_config,
                      feature_matcher_config const& matcher_config) -> match_result {
    // One fused call (vo_match_images): the two images go up once, the descriptors never leave
    // the GPU, keypoint sets and matches come back once.  Images of different shape or type take
    // the three-call route the reference spells out.
    if (img1.empty() || img2.empty() || img1.rows != img2.rows || img1.cols != img2.cols || img1.step != img2.step ||
        img1.type() != CV_8UC1 || img2.type() != CV_8UC1 || detector_config.max_features <= 0) {
        auto const det1 = detect_features(img1, detector_config);
        auto const det2 = detect_features(img2, detector_config);
        return match_features(det1.descriptors, det2.descriptors, det1.keypoints, det2.keypoints, matcher_config);
    }
    match_result r;
    vo_ctx* c = ctx();
    if (!c) return r;
    int const cap = detector_config.max_features + 8 * 256;
    auto& t = tls();
    t.k1.resize((size_t)cap);
    t.k2.resize((size_t)cap);
    t.qi.resize((size_t)cap);
    t.ti.resize((size_t)cap);
    t.dist.resize((size_t)cap);
    int n1 = 0, n2 = 0, m = 0;
    if (!ok(vo_match_images(c, img1.data, img2.data, img1.rows, img1.cols, (int)img1.step, detector_config.max_features,
                            matcher_config.ratio_threshold, t.k1.data(), &n1, t.k2.data(), &n2, cap, t.qi.data(),
                            t.ti.data(), t.dist.data(), &m)))
        return r;
    r.matches.reserve((size_t)m);
    r.points1.reserve((size_t)m);
    r.points2.reserve((size_t)m);
    for (int i = 0; i < m; ++i) {
        cv::DMatch dm;
        dm.queryIdx = t.qi[(size_t)i];
        dm.trainIdx = t.ti[(size_t)i];
        dm.imgIdx = 0;
        dm.distance = t.dist[(size_t)i];
        r.matches.push_back(dm);
        vo_keypoint const& a = t.k1[(size_t)dm.queryIdx];
        vo_keypoint const& b = t.k2[(size_t)dm.trainIdx];
        r.points1.emplace_back(a.x, a.y);
        r.points2.emplace_back(b.x, b.y);
    }
    return r;
}

#ifndef VO_WITH_LIGHTGLUE   // (with it, the reference's own src/image_matcher.cpp defines the factory)
auto create_image_matcher(std::string_view name) -> image_matcher {
    if (name == "orb" || name.empty()) return orb_matcher{};
    throw std::runtime_error("Unknown matcher: " + std::string(name));
}
#endif

auto camera_intrinsics::camera_matrix() const -> cv::Mat {
    cv::Mat m(3, 3, CV_64FC1);
    double const v[9] = {fx, 0.0, cx, 0.0, fy, cy, 0.0, 0.0, 1.0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m.at<double>(i, j) = v[i * 3 + j];
    return m;
}

auto camera_intrinsics::load_from_yaml(std::string_view filepath) -> tl::expected<camera_intrinsics, std::string> {
    std::ifstream in{std::string(filepath)};
    if (!in.is_open()) return tl::unexpected("Could not open camera file: " + std::string(filepath));
    camera_intrinsics k{};
    std::string line;
    while (std::getline(in, line)) {
        auto colon = line.find(':');
        if (colon == std::string::npos || line[0] == '%' || line[0] == '#') continue;
        std::string key = line.substr(0, colon);
        key.erase(0, key.find_first_not_of(" \t"));
        key.erase(key.find_last_not_of(" \t") + 1);
        std::istringstream vs(line.substr(colon + 1));
        double v = 0;
        if (!(vs >> v)) continue;
        if (key == "fx") k.fx = v;
        else if (key == "fy") k.fy = v;
        else if (key == "cx") k.cx = v;
        else if (key == "cy") k.cy = v;
    }
    return k;
}

auto estimate_motion(std::span<cv::Point2f const> points1, std::span<cv::Point2f const> points2,
                     camera_intrinsics const& intrinsics, motion_estimator_config const& config) -> motion_estimate {
    motion_estimate r{};
    r.valid = false;
    r.inliers = 0;
    if (points1.size() < (size_t)min_essential_points || points2.size() < (size_t)min_essential_points) return r;
    vo_ctx* c = ctx();
    if (!c) return r;
    static_assert(sizeof(cv::Point2f) == 8);
    int n = (int)std::min(points1.size(), points2.size());
    double const K[4] = {intrinsics.fx, intrinsics.fy, intrinsics.cx, intrinsics.cy};
    double R[9], t[3];
    int inliers = 0, valid = 0;
    if (!ok(vo_estimate_motion(c, reinterpret_cast<float const*>(points1.data()),
                               reinterpret_cast<float const*>(points2.data()), n, K, config.ransac_threshold,
                               config.ransac_confidence, 0, R, t, &inliers, &valid, nullptr)))
        return r;
    if (!valid) return r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.rotation(i, j) = R[i * 3 + j];
        r.translation(i) = t[i];
    }
    r.inliers = inliers;
    r.valid = true;
    return r;
}

namespace {

// Packs the frames and checks them; on failure leaves the reason in last_error().
bool pack_frames(std::span<cv::Mat const> frames, int max_features, std::vector<uint8_t>& packed, int& rows, int& cols) {
    auto& t = tls();
    if (frames.size() < 2) { t.err = "process_sequence: fewer than two frames"; return false; }
    if (max_features <= 0) { t.err = "process_sequence: max_features must be positive"; return false; }
    rows = frames[0].rows;
    cols = frames[0].cols;
    for (size_t i = 0; i < frames.size(); ++i) {
        auto const& f = frames[i];
        if (f.empty() || f.rows != rows || f.cols != cols || f.type() != CV_8UC1) {
            t.err = "process_sequence: frame " + std::to_string(i) + " is empty or differs from frame 0 in size or type";
            return false;
        }
    }
    size_t const image_bytes = (size_t)rows * cols;
    packed.resize(frames.size() * image_bytes);
    for (size_t i = 0; i < frames.size(); ++i)
        for (int r = 0; r < rows; ++r) std::memcpy(&packed[i * image_bytes + (size_t)r * cols], frames[i].ptr<uint8_t>(r), cols);
    return true;
}

auto to_steps(std::vector<vo_pair_result> const& rec) -> std::vector<sequence_step> {
    std::vector<sequence_step> steps(rec.size());
    for (size_t i = 0; i < rec.size(); ++i) {
        sequence_step& s = steps[i];
        s.keypoints1 = rec[i].n_kp1;
        s.keypoints2 = rec[i].n_kp2;
        s.matches = rec[i].n_matches;
        s.motion.valid = rec[i].valid != 0;
        s.motion.inliers = rec[i].inliers;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) s.motion.rotation(a, b) = rec[i].R[a * 3 + b];
            s.motion.translation(a) = rec[i].t[a];
        }
    }
    return steps;
}

// One context per GPU for the multi-device call, created on first use and kept for the life of
// the process; the mutex serialises whole calls (a context serves one host thread at a time).
struct device_pool {
    std::mutex m;
    std::map<int, vo_ctx*> ctx;
    ~device_pool() {
        for (auto& kv : ctx) vo_ctx_destroy(kv.second);
    }
};
device_pool& pool() {
    static device_pool p;
    return p;
}

}  // namespace

auto process_sequence(std::span<cv::Mat const> frames, camera_intrinsics const& intrinsics, int max_features,
                      float ratio_threshold, motion_estimator_config const& config) -> std::vector<sequence_step> {
    std::vector<uint8_t> packed;
    int rows = 0, cols = 0;
    if (!pack_frames(frames, max_features, packed, rows, cols)) return {};
    vo_ctx* c = ctx();
    if (!c) return {};
    std::vector<vo_pair_result> rec(frames.size() - 1);
    double const K[4] = {intrinsics.fx, intrinsics.fy, intrinsics.cx, intrinsics.cy};
    int const rc = vo_pipeline_sequence(c, packed.data(), (int)frames.size(), rows, cols, cols, (size_t)rows * cols,
                                        max_features, ratio_threshold, K, config.ransac_threshold,
                                        config.ransac_confidence, 0, rec.data());
    // a truncated keypoint list (VO_ERR_CAPACITY) still yields every record: keep them, leave the
    // message in last_error()
    if (!ok(rc) && rc != VO_ERR_CAPACITY) return {};
    return to_steps(rec);
}

auto process_sequence(std::span<cv::Mat const> frames, camera_intrinsics const& intrinsics, std::span<int const> devices,
                      int max_features, float ratio_threshold, motion_estimator_config const& config)
    -> std::vector<sequence_step> {
    if (devices.empty()) return process_sequence(frames, intrinsics, max_features, ratio_threshold, config);
    std::vector<uint8_t> packed;
    int rows = 0, cols = 0;
    if (!pack_frames(frames, max_features, packed, rows, cols)) return {};
    auto& p = pool();
    std::lock_guard<std::mutex> lock(p.m);
    std::vector<vo_ctx*> ctxs;
    for (int dev : devices) {
        auto it = p.ctx.find(dev);
        if (it == p.ctx.end()) {
            vo_ctx* c = nullptr;
            int const rc = vo_ctx_create(dev, &c);
            if (rc != VO_OK || !c) {
                tls().err = "vo_ctx_create(device " + std::to_string(dev) + ") failed with code " + std::to_string(rc);
                return {};
            }
            it = p.ctx.emplace(dev, c).first;
        }
        for (vo_ctx* have : ctxs)
            if (have == it->second) {
                tls().err = "process_sequence: device " + std::to_string(dev) + " is listed twice";
                return {};
            }
        ctxs.push_back(it->second);
    }
    std::vector<vo_pair_result> rec(frames.size() - 1);
    double const K[4] = {intrinsics.fx, intrinsics.fy, intrinsics.cx, intrinsics.cy};
    int const rc = vo_pipeline_sequence_multi(ctxs.data(), (int)ctxs.size(), packed.data(), (int)frames.size(), rows, cols,
                                              cols, (size_t)rows * cols, max_features, ratio_threshold, K,
                                              config.ransac_threshold, config.ransac_confidence, 0, rec.data());
    if (rc != VO_OK) tls().err = vo_last_error(ctxs[0]);
    if (rc != VO_OK && rc != VO_ERR_CAPACITY) return {};
    return to_steps(rec);
}

}  // namespace visual_odometry
