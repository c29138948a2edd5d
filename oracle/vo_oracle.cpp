// CPU oracle for the visual-odometry front-end hot path.
//
// TEST INFRASTRUCTURE ONLY.  This file restates, on the CPU and in scalar
// code, the algorithms that the reference reaches through OpenCV 4.12:
//   detect_features        ref:src/feature_detector.cpp:12-18  (cv::ORB::detectAndCompute)
//   detect_keypoints_only  ref:src/feature_detector.cpp:20-27  (cv::ORB::detect)
//   match_features         ref:src/feature_matcher.cpp:15-42   (cv::BFMatcher::knnMatch k=2 + ratio)
//   estimate_motion        ref:src/motion_estimator.cpp:44-95  (cv::findEssentialMat + cv::recoverPose)
// Only tests, the smoke check and the benchmark's cpu_baseline leg may load
// it; the shipped GPU path never links, imports or calls anything here.
//
// PARITY UNPINNED: OpenCV (headers, libraries, sources, cv2) is absent from
// the build environment and the reference's own tests pin no numeric result
// on this path, so every OpenCV-internal detail below (fixed-point resize and
// blur, FAST score, Harris/IC arithmetic order, rBRIEF pattern, cv::RNG
// stream, 5-point solver) is a restatement of the published algorithm that
// could not be diffed against OpenCV itself.  What IS pinned: the behavioural
// contract of the reference's tests (ref:tests/test_feature_detector.cpp,
// ref:tests/test_feature_matcher.cpp, ref:tests/test_motion_estimator.cpp).
//
// Build: g++ -O3 -mavx2 -mpopcnt -std=c++17 -ffp-contract=off -fPIC -shared (see Makefile).
// -ffp-contract=off matters: float expressions must round after every
// operation, as a non-FMA x86-64 OpenCV build does.

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <vector>

// Study switches (oracle_set_variant): the two places where this restatement reads OpenCV one
// way and another reading is defensible.  Defaults are what the parity tests pin the GPU to.
static bool g_trig_float = false;          // descriptor rotation through cosf / sinf
static bool g_five_point_literal = false;  // OpenCV-literal minimal solver: one chart, roots kept only
                                           // with |imag| <= 1e-10, no refinement, no duplicate drop

namespace {

// ---------------------------------------------------------------- helpers
inline int cv_round(double v) { return (int)std::lrint(v); }   // round-half-even
inline int cv_round(float v) { return (int)std::lrintf(v); }
inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_ceil(double v) { return (int)std::ceil(v); }

struct KeyPoint {  // same 28-byte layout as cv::KeyPoint
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

constexpr int kLevels = 8;
constexpr float kScaleFactor = 1.2f;
constexpr int kEdgeThreshold = 31;
constexpr int kPatchSize = 31;
constexpr int kHalfPatch = 15;
constexpr int kFastThreshold = 20;
constexpr int kHarrisBlock = 7;
constexpr float kHarrisK = 0.04f;

const int8_t kPattern[256 * 4] = {
#include "orb_pattern.inc"
};

struct Level {
    int w = 0, h = 0;
    float scale = 1.f;
    std::vector<uint8_t> img;   // w*h, no border
    std::vector<uint8_t> blur;  // filled on demand
};

// ------------------------------------------------- pyramid (ORB_Impl::detectAndCompute)
// layerScale[l] = (float)pow((double)scaleFactor, l); size = cvRound(cols/scale).
void level_geometry(int cols, int rows, int l, int* w, int* h, float* s) {
    float sc = (float)std::pow((double)kScaleFactor, (double)l);
    *s = sc;
    *w = cv_round((float)cols / sc);
    *h = cv_round((float)rows / sc);
}

// cv::resize(..., INTER_LINEAR_EXACT) on 8UC1: fixed-point bilinear
// (resize.cpp: interpolationLinear<uint8_t>::getCoeffs, hlineResize,
// vlineResize).  Coefficients are 8.8 fixed point rounded half-even from
// fval = scale*(d+0.5)-0.5 with scale = 1/((double)dst/src).
struct LinCoeffs {
    std::vector<int> ofs;
    std::vector<uint16_t> c0, c1;
    int minofs, maxofs;
};
LinCoeffs lin_coeffs(int src, int dst) {
    LinCoeffs r;
    r.ofs.assign(dst, 0);
    r.c0.assign(dst, 0);
    r.c1.assign(dst, 0);
    r.minofs = 0;
    r.maxofs = dst;
    double inv_scale = (double)dst / (double)src;
    double scale = 1.0 / inv_scale;
    for (int v = 0; v < dst; ++v) {
        double fval = scale * ((double)v + 0.5) - 0.5;
        int ival = cv_floor(fval);
        if (ival >= 0 && src > 1) {
            if (ival < src - 1) {
                r.ofs[v] = ival;
                int c1 = cv_round((fval - (double)ival) * 256.0);
                r.c1[v] = (uint16_t)c1;
                r.c0[v] = (uint16_t)(256 - c1);
            } else {
                r.ofs[v] = src - 1;
                r.maxofs = std::min(r.maxofs, v);
            }
        } else {
            r.minofs = std::max(r.minofs, v + 1);
        }
    }
    return r;
}

void resize_linear_exact(const uint8_t* src, int sw, int sh, int sstep, uint8_t* dst, int dw,
                         int dh) {
    LinCoeffs cx = lin_coeffs(sw, dw), cy = lin_coeffs(sh, dh);
    // horizontal pass of one source row -> 8.8 fixed point
    auto hline = [&](int sy, std::vector<uint32_t>& out) {
        const uint8_t* s = src + (size_t)sy * sstep;
        out.resize(dw);
        for (int x = 0; x < dw; ++x) {
            if (x < cx.minofs)
                out[x] = (uint32_t)s[0] << 8;
            else if (x >= cx.maxofs)
                out[x] = (uint32_t)s[cx.ofs[dw - 1]] << 8;
            else
                out[x] = (uint32_t)cx.c0[x] * s[cx.ofs[x]] + (uint32_t)cx.c1[x] * s[cx.ofs[x] + 1];
        }
    };
    std::vector<uint32_t> r0, r1;
    for (int y = 0; y < dh; ++y) {
        uint8_t* d = dst + (size_t)y * dw;
        if (y < cy.minofs || y >= cy.maxofs) {
            int sy = (y < cy.minofs) ? 0 : cy.ofs[dh - 1];
            hline(sy, r0);
            for (int x = 0; x < dw; ++x) d[x] = (uint8_t)std::min<uint32_t>((r0[x] + 128) >> 8, 255);
        } else {
            hline(cy.ofs[y], r0);
            hline(cy.ofs[y] + 1, r1);
            uint32_t m0 = cy.c0[y], m1 = cy.c1[y];
            for (int x = 0; x < dw; ++x) {
                uint32_t v = m0 * r0[x] + m1 * r1[x];
                d[x] = (uint8_t)std::min<uint32_t>((v + 32768u) >> 16, 255);
            }
        }
    }
}

std::vector<Level> build_pyramid(const uint8_t* img, int rows, int cols, int step) {
    std::vector<Level> pyr(kLevels);
    for (int l = 0; l < kLevels; ++l) {
        Level& L = pyr[l];
        level_geometry(cols, rows, l, &L.w, &L.h, &L.scale);
        L.img.resize((size_t)L.w * L.h);
        if (l == 0) {
            for (int y = 0; y < rows; ++y) memcpy(&L.img[(size_t)y * cols], img + (size_t)y * step, cols);
        } else {
            const Level& P = pyr[l - 1];
            if (L.w > 0 && L.h > 0 && P.w > 0 && P.h > 0)
                resize_linear_exact(P.img.data(), P.w, P.h, P.w, L.img.data(), L.w, L.h);
        }
    }
    return pyr;
}

inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) as ORB calls it: in place on
// a sub-matrix of the pyramid buffer, which routes to sepFilter2D with the
// float kernel converted tap-by-tap to 8-bit fixed point (sum 257, not 256):
// getGaussianKernel(7, 2) * 256 rounded = {18,34,49,55,49,34,18}.  Rows are
// exact int32 sums; the column stage rounds sum/65536: OpenCV's vector body
// (all columns below width & ~3) rounds half-to-even through float, the
// scalar tail rounds half-up ((sum + 32768) >> 16).  Both saturate to 255.
void gaussian_taps(int taps[7]) {
    double k[7], sum = 0;
    for (int i = 0; i < 7; ++i) {
        double x = i - 3;
        k[i] = std::exp(-x * x / (2.0 * 2.0 * 2.0));
        sum += k[i];
    }
    for (int i = 0; i < 7; ++i) {
        float kf = (float)(k[i] / sum);
        taps[i] = cv_round(kf * 256.f);
    }
}

void blur_level(Level& L) {
    int taps[7];
    gaussian_taps(taps);
    int w = L.w, h = L.h;
    std::vector<int32_t> rows((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int i = -3; i <= 3; ++i) s += taps[i + 3] * L.img[(size_t)y * w + reflect101(x + i, w)];
            rows[(size_t)y * w + x] = s;
        }
    L.blur.resize((size_t)w * h);
    int vec_end = w & ~3;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int64_t s = 0;
            for (int j = -3; j <= 3; ++j) s += (int64_t)taps[j + 3] * rows[(size_t)reflect101(y + j, h) * w + x];
            int64_t q = s >> 16, rem = s & 0xFFFF;
            int64_t r;
            if (x < vec_end)
                r = q + ((rem > 32768) || (rem == 32768 && (q & 1)));  // half-even
            else
                r = (s + 32768) >> 16;  // half-up
            L.blur[(size_t)y * w + x] = (uint8_t)std::min<int64_t>(r, 255);
        }
}

// ------------------------------------------------- FAST-9/16 (fast.cpp, fast_score.cpp)
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// max(threshold, best bright arc, best dark arc) - 1, where an arc is 9
// contiguous ring pixels: identical to cornerScore<16>.  Returns 0 for a
// non-corner (no arc strictly beyond the threshold).
int fast_score(const uint8_t* p, int step, int threshold) {
    int v = p[0];
    int d[25];
    for (int k = 0; k < 16; ++k) d[k] = v - p[kRingDy[k] * step + kRingDx[k]];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    int best_b = INT_MIN, best_d = INT_MIN;
    for (int s = 0; s < 16; ++s) {
        int mn = INT_MAX, mx = INT_MIN;
        for (int j = 0; j < 9; ++j) {
            mn = std::min(mn, d[s + j]);
            mx = std::max(mx, d[s + j]);
        }
        best_b = std::max(best_b, mn);
        best_d = std::max(best_d, -mx);
    }
    int m = std::max(best_b, best_d);
    if (m <= threshold) return 0;
    return m - 1;
}

struct Cand {
    int x, y;
    float response;
};

// FAST with non-max suppression over rows/cols [3, n-3), keypoints in raster
// order, response = score (FAST_t<16> with nonmax_suppression = true).
std::vector<Cand> fast_detect(const uint8_t* img, int w, int h, int step, int threshold) {
    std::vector<Cand> out;
    if (w < 7 || h < 7) return out;
    std::vector<uint8_t> sc((size_t)w * h, 0);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x)
            sc[(size_t)y * w + x] = (uint8_t)fast_score(img + (size_t)y * step + x, step, threshold);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = sc[(size_t)y * w + x];
            if (!s) continue;
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    if (s <= sc[(size_t)(y + dy) * w + x + dx]) { keep = false; break; }
                }
            if (keep) out.push_back({x, y, (float)s});
        }
    return out;
}

// KeyPointsFilter::retainBest.  mode 0: keep every element whose response is
// >= the n-th largest, in the incoming (raster) order.  mode 1: the same set
// in the order libstdc++'s nth_element + partition leave it, which is what
// an OpenCV built against libstdc++ returns.
void retain_best(std::vector<Cand>& k, int n, int mode) {
    if (n < 0 || (int)k.size() <= n) return;
    if (n == 0) { k.clear(); return; }
    if (mode == 1) {
        std::nth_element(k.begin(), k.begin() + n - 1, k.end(),
                         [](const Cand& a, const Cand& b) { return a.response > b.response; });
        float amb = k[n - 1].response;
        auto e = std::partition(k.begin() + n, k.end(), [amb](const Cand& a) { return a.response >= amb; });
        k.resize(e - k.begin());
    } else {
        std::vector<float> r(k.size());
        for (size_t i = 0; i < k.size(); ++i) r[i] = k[i].response;
        std::nth_element(r.begin(), r.begin() + n - 1, r.end(), std::greater<float>());
        float amb = r[n - 1];
        std::vector<Cand> o;
        for (auto& c : k)
            if (c.response >= amb) o.push_back(c);
        k.swap(o);
    }
}

// HarrisResponses (orb.cpp): 7x7 block of Sobel-like gradients, int32 sums,
// float32 response in a fixed operation order.
float harris_response(const uint8_t* img, int step, int x0, int y0) {
    const int r = kHarrisBlock / 2;
    float scale = 1.f / ((1 << 2) * kHarrisBlock * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < kHarrisBlock; ++i)
        for (int j = 0; j < kHarrisBlock; ++j) {
            const uint8_t* p = img + (size_t)(y0 - r + i) * step + (x0 - r + j);
            int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
            int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    float fa = (float)a, fb = (float)b, fc = (float)c;
    float t1 = fa * fb;
    float t2 = fc * fc;
    float s = fa + fb;
    float t3 = kHarrisK * s;
    t3 = t3 * s;
    float v = t1 - t2;
    v = v - t3;
    return v * scale_sq_sq;
}

void compute_umax(int umax[kHalfPatch + 2]) {
    int v, v0, vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
    for (v = 0; v < kHalfPatch + 2; ++v) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)kHalfPatch * kHalfPatch - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// cv::fastAtan2 scalar path (mathfuncs_core: atan_f32), degrees in [0,360).
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float ic_angle(const uint8_t* img, int step, int x0, int y0, const int* umax) {
    const uint8_t* center = img + (size_t)y0 * step + x0;
    int m01 = 0, m10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * step], vm = center[u - v * step];
            vsum += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return fast_atan2((float)m01, (float)m10);
}

void features_per_level(int nfeatures, int n[kLevels]) {
    float factor = (float)(1.0 / (double)kScaleFactor);
    float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kLevels));
    int sum = 0;
    for (int l = 0; l < kLevels - 1; ++l) {
        n[l] = cv_round(nd);
        sum += n[l];
        nd *= factor;
    }
    n[kLevels - 1] = std::max(nfeatures - sum, 0);
}

// computeOrbDescriptors (orb.cpp), WTA_K = 2.
void orb_descriptor(const Level& L, const KeyPoint& kp, uint8_t* desc) {
    float scale = 1.f / L.scale;
    float angle = kp.angle;
    angle *= (float)(M_PI / 180.f);
    // orb.cpp writes (float)cos(angle) with a float argument; whether that resolves to the float
    // or the double libm routine depends on the overload set in scope.  Default: the double
    // routine rounded to float; g_trig_float: libm's cosf / sinf (study switch, see
    // oracle_set_variant).
    float a, b;
    if (g_trig_float) {
        a = cosf(angle);
        b = sinf(angle);
    } else {
        a = (float)std::cos((double)angle);
        b = (float)std::sin((double)angle);
    }
    int cy = cv_round(kp.y * scale), cx = cv_round(kp.x * scale);
    const uint8_t* center = &L.blur[(size_t)cy * L.w + cx];
    int step = L.w;
    auto get = [&](int idx, const int8_t* pat) -> int {
        float px = (float)pat[idx * 2], py = (float)pat[idx * 2 + 1];
        float x = px * a - py * b;
        float y = px * b + py * a;
        int ix = cv_round(x), iy = cv_round(y);
        return center[iy * step + ix];
    };
    const int8_t* pat = kPattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            int t0 = get(2 * k, pat), t1 = get(2 * k + 1, pat);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

// ORB_Impl::detectAndCompute with OpenCV's defaults except nfeatures.
// order_mode 0: per level, raster order (canonical); 1: libstdc++ order.
int orb_detect(const uint8_t* img, int rows, int cols, int step, int nfeatures, int want_desc,
               int order_mode, std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc) {
    kps.clear();
    desc.clear();
    if (!img || rows <= 0 || cols <= 0) return 0;
    std::vector<Level> pyr = build_pyramid(img, rows, cols, step);
    int nper[kLevels];
    features_per_level(nfeatures, nper);
    int umax[kHalfPatch + 2];
    compute_umax(umax);

    std::vector<std::vector<Cand>> per(kLevels);
    for (int l = 0; l < kLevels; ++l) {
        Level& L = pyr[l];
        if (L.w <= 0 || L.h <= 0) continue;
        std::vector<Cand> c = fast_detect(L.img.data(), L.w, L.h, L.w, kFastThreshold);
        // runByImageBorder(edgeThreshold)
        if (L.h <= kEdgeThreshold * 2 || L.w <= kEdgeThreshold * 2)
            c.clear();
        else {
            std::vector<Cand> f;
            for (auto& k : c)
                if (k.x >= kEdgeThreshold && k.x < L.w - kEdgeThreshold && k.y >= kEdgeThreshold &&
                    k.y < L.h - kEdgeThreshold)
                    f.push_back(k);
            c.swap(f);
        }
        retain_best(c, 2 * nper[l], order_mode);
        for (auto& k : c) k.response = harris_response(L.img.data(), L.w, k.x, k.y);
        retain_best(c, nper[l], order_mode);
        per[l] = c;
    }
    for (int l = 0; l < kLevels; ++l) {
        Level& L = pyr[l];
        for (auto& c : per[l]) {
            KeyPoint k;
            k.angle = ic_angle(L.img.data(), L.w, c.x, c.y, umax);
            k.x = (float)c.x * L.scale;
            k.y = (float)c.y * L.scale;
            k.size = kPatchSize * L.scale;
            k.response = c.response;
            k.octave = l;
            k.class_id = -1;
            kps.push_back(k);
        }
    }
    if (want_desc && !kps.empty()) {
        for (int l = 0; l < kLevels; ++l)
            if (!per[l].empty()) blur_level(pyr[l]);
        desc.resize(kps.size() * 32);
        for (size_t i = 0; i < kps.size(); ++i) orb_descriptor(pyr[kps[i].octave], kps[i], &desc[i * 32]);
    }
    return (int)kps.size();
}

// ------------------------------------------------- brute-force Hamming kNN(2)
// cv::batchDistance K>0 path: ascending scan, strict '<' insertion.
void knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2) {
    for (int i = 0; i < nq; ++i) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        const uint64_t* a = (const uint64_t*)(q + (size_t)i * 32);
        uint64_t a0, a1, a2, a3;
        memcpy(&a0, a, 8); memcpy(&a1, a + 1, 8); memcpy(&a2, a + 2, 8); memcpy(&a3, a + 3, 8);
        for (int j = 0; j < nt; ++j) {
            uint64_t b0, b1, b2, b3;
            const uint8_t* bp = t + (size_t)j * 32;
            memcpy(&b0, bp, 8); memcpy(&b1, bp + 8, 8); memcpy(&b2, bp + 16, 8); memcpy(&b3, bp + 24, 8);
            int d = __builtin_popcountll(a0 ^ b0) + __builtin_popcountll(a1 ^ b1) +
                    __builtin_popcountll(a2 ^ b2) + __builtin_popcountll(a3 ^ b3);
            if (d < d1) {
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        idx2[i * 2] = i0; idx2[i * 2 + 1] = i1;
        dist2[i * 2] = i0 < 0 ? -1 : d0; dist2[i * 2 + 1] = i1 < 0 ? -1 : d1;
    }
}

// ------------------------------------------------- small dense linear algebra (fp64)
// One-sided Jacobi SVD of an m x n matrix (m >= n), row-major A.  Returns
// singular values w[n] (descending), U (m x n, columns) and V (n x n,
// columns), both row-major.  Hestenes rotations as in OpenCV's JacobiSVD.
void jacobi_svd(const double* A, int m, int n, double* w, double* U, double* V) {
    std::vector<double> a(A, A + (size_t)m * n);  // columns get orthogonalised
    std::vector<double> v((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    const double eps = DBL_EPSILON * 2;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool changed = false;
        for (int i = 0; i < n - 1; ++i)
            for (int j = i + 1; j < n; ++j) {
                double p = 0, aa = 0, bb = 0;
                for (int k = 0; k < m; ++k) {
                    double x = a[(size_t)k * n + i], y = a[(size_t)k * n + j];
                    p += x * y; aa += x * x; bb += y * y;
                }
                if (std::fabs(p) <= eps * std::sqrt(aa * bb)) continue;
                p *= 2;
                double beta = aa - bb, gamma = std::hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                for (int k = 0; k < m; ++k) {
                    double x = a[(size_t)k * n + i], y = a[(size_t)k * n + j];
                    a[(size_t)k * n + i] = c * x + s * y;
                    a[(size_t)k * n + j] = -s * x + c * y;
                }
                for (int k = 0; k < n; ++k) {
                    double x = v[(size_t)k * n + i], y = v[(size_t)k * n + j];
                    v[(size_t)k * n + i] = c * x + s * y;
                    v[(size_t)k * n + j] = -s * x + c * y;
                }
                changed = true;
            }
        if (!changed) break;
    }
    std::vector<double> sv(n);
    std::vector<int> ord(n);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += a[(size_t)k * n + i] * a[(size_t)k * n + i];
        sv[i] = std::sqrt(s);
        ord[i] = i;
    }
    std::sort(ord.begin(), ord.end(), [&](int x, int y) { return sv[x] > sv[y]; });
    for (int jj = 0; jj < n; ++jj) {
        int j = ord[jj];
        w[jj] = sv[j];
        for (int k = 0; k < m; ++k) U[(size_t)k * n + jj] = sv[j] > 0 ? a[(size_t)k * n + j] / sv[j] : 0.0;
        for (int k = 0; k < n; ++k) V[(size_t)k * n + jj] = v[(size_t)k * n + j];
    }
}

// Complete `have` orthonormal vectors of dimension d (rows of B) to a full
// basis; the new vectors are appended as rows.  Gram-Schmidt (twice) on the
// unit vectors, always taking the candidate with the largest residual.
void complete_basis(std::vector<double>& B, int d, int have) {
    for (int cnt = have; cnt < d; ++cnt) {
        int best = -1;
        double bestn = -1;
        std::vector<double> bestv(d);
        for (int e = 0; e < d; ++e) {
            std::vector<double> x(d, 0.0);
            x[e] = 1.0;
            for (int pass = 0; pass < 2; ++pass)
                for (int r = 0; r < cnt; ++r) {
                    double dot = 0;
                    for (int k = 0; k < d; ++k) dot += x[k] * B[(size_t)r * d + k];
                    for (int k = 0; k < d; ++k) x[k] -= dot * B[(size_t)r * d + k];
                }
            double nn = 0;
            for (int k = 0; k < d; ++k) nn += x[k] * x[k];
            if (nn > bestn) { bestn = nn; best = e; bestv = x; }
        }
        (void)best;
        double inv = 1.0 / std::sqrt(bestn);
        for (int k = 0; k < d; ++k) B.push_back(bestv[k] * inv);
    }
}

// LU solve with partial pivoting: X = A^-1 B (A n x n, B n x m), row-major.
bool lu_solve(std::vector<double> A, int n, std::vector<double>& B, int m) {
    for (int i = 0; i < n; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j)
            if (std::fabs(A[(size_t)j * n + i]) > std::fabs(A[(size_t)k * n + i])) k = j;
        if (std::fabs(A[(size_t)k * n + i]) < DBL_EPSILON) return false;
        if (k != i) {
            for (int j = i; j < n; ++j) std::swap(A[(size_t)i * n + j], A[(size_t)k * n + j]);
            for (int j = 0; j < m; ++j) std::swap(B[(size_t)i * m + j], B[(size_t)k * m + j]);
        }
        double d = -1 / A[(size_t)i * n + i];
        for (int j = i + 1; j < n; ++j) {
            double alpha = A[(size_t)j * n + i] * d;
            for (int c = i + 1; c < n; ++c) A[(size_t)j * n + c] += alpha * A[(size_t)i * n + c];
            for (int c = 0; c < m; ++c) B[(size_t)j * m + c] += alpha * B[(size_t)i * m + c];
        }
    }
    for (int i = n - 1; i >= 0; --i)
        for (int j = 0; j < m; ++j) {
            double s = B[(size_t)i * m + j];
            for (int k = i + 1; k < n; ++k) s -= A[(size_t)i * n + k] * B[(size_t)k * m + j];
            B[(size_t)i * m + j] = s / A[(size_t)i * n + i];
        }
    return true;
}

// cv::solvePoly: Durand-Kerner from the (1+i)^k start, at most 300 sweeps.
// coeffs ascending (coeffs[i] * x^i).
std::vector<std::complex<double>> solve_poly(const double* c0, int deg) {
    typedef std::complex<double> C;
    int n = deg;
    for (; n > 1; --n)
        if (std::fabs(c0[n]) > DBL_EPSILON) break;
    std::vector<C> coeffs(n + 1), roots(n);
    for (int i = 0; i <= n; ++i) coeffs[i] = C(c0[i], 0);
    C p(1, 0), r(1, 1);
    for (int i = 0; i < n; ++i) { roots[i] = p; p = p * r; }
    for (int iter = 0; iter < 300; ++iter) {
        double maxDiff = 0;
        for (int i = 0; i < n; ++i) {
            p = roots[i];
            C num = coeffs[n], denom = coeffs[n];
            for (int j = 0; j < n; ++j) {
                num = num * p + coeffs[n - j - 1];
                if (j != i && p != roots[j]) denom = denom * (p - roots[j]);
            }
            num /= denom;
            roots[i] = p - num;
            maxDiff = std::max(maxDiff, std::abs(num));
        }
        if (maxDiff <= 0) break;
    }
    return roots;
}

// ------------------------------------------------- five-point solver (Nister)
// Polynomials in (x,y,z) of total degree <= 3 stored on the 4x4x4 exponent grid.
struct Poly3 {
    double c[64];
    Poly3() { memset(c, 0, sizeof c); }
    static int id(int i, int j, int k) { return (i * 4 + j) * 4 + k; }
};
Poly3 padd(const Poly3& a, const Poly3& b, double sb = 1.0) {
    Poly3 r;
    for (int i = 0; i < 64; ++i) r.c[i] = a.c[i] + sb * b.c[i];
    return r;
}
Poly3 pmul(const Poly3& a, const Poly3& b) {
    Poly3 r;
    for (int i = 0; i < 64; ++i) {
        if (a.c[i] == 0) continue;
        int ai = i >> 4, aj = (i >> 2) & 3, ak = i & 3;
        for (int j = 0; j < 64; ++j) {
            if (b.c[j] == 0) continue;
            int bi = j >> 4, bj = (j >> 2) & 3, bk = j & 3;
            if (ai + bi > 3 || aj + bj > 3 || ak + bk > 3) continue;
            r.c[Poly3::id(ai + bi, aj + bj, ak + bk)] += a.c[i] * b.c[j];
        }
    }
    return r;
}

// Monomial order of the 10 x 20 constraint matrix: the ten monomials of
// degree >= 2 in (x,y) first, then x*{z^2,z,1}, y*{z^2,z,1}, {z^3,z^2,z,1}.
const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                          {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                          {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

// f(E) = [det E ; 2 E E^T E - trace(E E^T) E] : the ten cubic constraints an
// essential matrix satisfies, and their derivative along a direction D.
void essential_constraints(const double* E, double* f) {
    double EEt[9], T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += E[i * 3 + k] * E[j * 3 + k];
            EEt[i * 3 + j] = s;
        }
    double tr = EEt[0] + EEt[4] + EEt[8];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += EEt[i * 3 + k] * E[k * 3 + j];
            T[i * 3 + j] = 2 * s - tr * E[i * 3 + j];
        }
    f[0] = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
    for (int k = 0; k < 9; ++k) f[1 + k] = T[k];
}
void essential_constraints_dir(const double* E, const double* D, double* df) {
    // d det = trace(adj(E) D); d(2EE^TE - tr(EE^T)E) = 2(DE^TE + ED^TE + EE^TD) - 2 tr(ED^T) E - tr(EE^T) D
    double C[9] = {E[4] * E[8] - E[5] * E[7], E[5] * E[6] - E[3] * E[8], E[3] * E[7] - E[4] * E[6],
                   E[2] * E[7] - E[1] * E[8], E[0] * E[8] - E[2] * E[6], E[1] * E[6] - E[0] * E[7],
                   E[1] * E[5] - E[2] * E[4], E[2] * E[3] - E[0] * E[5], E[0] * E[4] - E[1] * E[3]};  // cofactors
    double dd = 0;
    for (int k = 0; k < 9; ++k) dd += C[k] * D[k];
    df[0] = dd;
    double EtE[9], EEt[9], DtE[9];
    double trEEt = 0, trEDt = 0;
    for (int k = 0; k < 9; ++k) { trEEt += E[k] * E[k]; trEDt += E[k] * D[k]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0, b = 0, c = 0;
            for (int k = 0; k < 3; ++k) {
                a += E[k * 3 + i] * E[k * 3 + j];   // E^T E
                b += E[i * 3 + k] * E[j * 3 + k];   // E E^T
                c += D[k * 3 + i] * E[k * 3 + j];   // D^T E
            }
            EtE[i * 3 + j] = a; EEt[i * 3 + j] = b; DtE[i * 3 + j] = c;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k)
                s += D[i * 3 + k] * EtE[k * 3 + j] + E[i * 3 + k] * DtE[k * 3 + j] + EEt[i * 3 + k] * D[k * 3 + j];
            df[1 + i * 3 + j] = 2 * s - 2 * trEDt * E[i * 3 + j] - trEEt * D[i * 3 + j];
        }
}

// Gauss-Newton refinement of (x,y,z) on the ten constraints for
// E = x*B0 + y*B1 + z*B2 + B3.  OpenCV does not do this: its Durand-Kerner
// root finder stops after 300 sweeps whether converged or not, which leaves a
// few percent of the models accurate to 3-4 digits only.  Refining makes every
// model the exact solution of the five-point problem to rounding, which is the
// quantity any correct solver must agree on.  Returns the largest constraint
// residual of the unit-norm result.
double polish_model(const double* const B[4], double* xyz, double* Eout) {
    double E[9];
    const int kSteps = 12;
    for (int it = 0; it <= kSteps; ++it) {
        for (int k = 0; k < 9; ++k) E[k] = B[0][k] * xyz[0] + B[1][k] * xyz[1] + B[2][k] * xyz[2] + B[3][k];
        if (it == kSteps) break;
        double f[10], J[3][10];
        essential_constraints(E, f);
        for (int c = 0; c < 3; ++c) essential_constraints_dir(E, B[c], J[c]);
        double N[3][3], g[3];
        for (int a = 0; a < 3; ++a) {
            g[a] = 0;
            for (int k = 0; k < 10; ++k) g[a] += J[a][k] * f[k];
            for (int b = 0; b < 3; ++b) {
                double s = 0;
                for (int k = 0; k < 10; ++k) s += J[a][k] * J[b][k];
                N[a][b] = s;
            }
        }
        double lam = 1e-14 * (N[0][0] + N[1][1] + N[2][2]);
        for (int a = 0; a < 3; ++a) N[a][a] += lam;
        double det = N[0][0] * (N[1][1] * N[2][2] - N[1][2] * N[2][1]) - N[0][1] * (N[1][0] * N[2][2] - N[1][2] * N[2][0]) +
                     N[0][2] * (N[1][0] * N[2][1] - N[1][1] * N[2][0]);
        if (!(std::fabs(det) > 0)) break;
        double inv = 1.0 / det;
        double d0 = (g[0] * (N[1][1] * N[2][2] - N[1][2] * N[2][1]) - N[0][1] * (g[1] * N[2][2] - N[1][2] * g[2]) +
                     N[0][2] * (g[1] * N[2][1] - N[1][1] * g[2])) * inv;
        double d1 = (N[0][0] * (g[1] * N[2][2] - N[1][2] * g[2]) - g[0] * (N[1][0] * N[2][2] - N[1][2] * N[2][0]) +
                     N[0][2] * (N[1][0] * g[2] - g[1] * N[2][0])) * inv;
        double d2 = (N[0][0] * (N[1][1] * g[2] - g[1] * N[2][1]) - N[0][1] * (N[1][0] * g[2] - g[1] * N[2][0]) +
                     g[0] * (N[1][0] * N[2][1] - N[1][1] * N[2][0])) * inv;
        xyz[0] -= d0; xyz[1] -= d1; xyz[2] -= d2;
        // converged: one more pass only rebuilds E from the final (x,y,z)
        double big = std::max(1.0, std::max(std::fabs(xyz[0]), std::max(std::fabs(xyz[1]), std::fabs(xyz[2]))));
        if (std::max(std::fabs(d0), std::max(std::fabs(d1), std::fabs(d2))) <= 1e-14 * big) it = kSteps - 1;
    }
    double nrm = 0;
    for (int k = 0; k < 9; ++k) nrm += E[k] * E[k];
    nrm = std::sqrt(nrm);
    if (!(nrm > 0) || !std::isfinite(nrm)) return 1e300;
    for (int k = 0; k < 9; ++k) Eout[k] = E[k] / nrm;
    double f[10], r = 0;
    essential_constraints(Eout, f);
    for (int k = 0; k < 10; ++k) r = std::max(r, std::fabs(f[k]));
    return r;
}

// EMEstimatorCallback::runKernel in ONE chart of the null space: x1,x2 are 5 normalised
// points (x,y).  Writes up to 10 essential matrices (row-major 3x3, unit Frobenius norm).
// chart 0 is OpenCV's: E = x*E0 + y*E1 + z*E2 + E3 on the null-space basis as the SVD
// returns it.  chart k > 0 first reflects that basis by a fixed orthogonal 4x4 matrix
// (I - 2 u u^T): the solutions are the same matrices, but the 10x10 elimination and the
// hidden-variable polynomial are conditioned differently.
int five_point_chart(const double* x1, const double* x2, int chart, double* Es, double* dbg = nullptr) {
    // Q (5 x 9): rows x2^T E x1 = 0 in the order [x1x2, y1x2, x2, x1y2, y1y2, y2, x1, y1, 1]
    double Qt[9 * 5];  // Q transposed, 9 x 5
    for (int i = 0; i < 5; ++i) {
        double a = x1[i * 2], b = x1[i * 2 + 1], c = x2[i * 2], d = x2[i * 2 + 1];
        double row[9] = {a * c, b * c, c, a * d, b * d, d, a, b, 1.0};
        for (int k = 0; k < 9; ++k) Qt[k * 5 + i] = row[k];
    }
    double w[5], U[9 * 5], V[5 * 5];
    jacobi_svd(Qt, 9, 5, w, U, V);
    std::vector<double> B;  // rows: orthonormal vectors of R^9
    for (int j = 0; j < 5; ++j)
        for (int k = 0; k < 9; ++k) B.push_back(U[k * 5 + j]);
    complete_basis(B, 9, 5);
    if (chart > 0) {
        static const double U[3][4] = {{0.5, 0.5, 0.5, 0.5},
                                       {0.1825741858350554, -0.3651483716701107, 0.5477225575051661, -0.7302967433402214},
                                       {0.6708203932499369, 0.2236067977499790, -0.6708203932499369, 0.2236067977499790}};
        const double* u = U[(chart - 1) % 3];
        for (int k = 0; k < 9; ++k) {
            double dot = 0;
            for (int t = 0; t < 4; ++t) dot += u[t] * B[(5 + t) * 9 + k];
            for (int t = 0; t < 4; ++t) B[(5 + t) * 9 + k] -= 2.0 * u[t] * dot;
        }
    }
    const double* EE[4] = {&B[5 * 9], &B[6 * 9], &B[7 * 9], &B[8 * 9]};  // null space

    // E(x,y,z) = x*E0 + y*E1 + z*E2 + E3, entries as polynomials
    Poly3 E[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            E[r][c].c[Poly3::id(1, 0, 0)] = EE[0][r * 3 + c];
            E[r][c].c[Poly3::id(0, 1, 0)] = EE[1][r * 3 + c];
            E[r][c].c[Poly3::id(0, 0, 1)] = EE[2][r * 3 + c];
            E[r][c].c[Poly3::id(0, 0, 0)] = EE[3][r * 3 + c];
        }
    Poly3 eq[10];
    // det(E) = 0
    {
        Poly3 m0 = padd(pmul(E[1][1], E[2][2]), pmul(E[1][2], E[2][1]), -1);
        Poly3 m1 = padd(pmul(E[1][0], E[2][2]), pmul(E[1][2], E[2][0]), -1);
        Poly3 m2 = padd(pmul(E[1][0], E[2][1]), pmul(E[1][1], E[2][0]), -1);
        eq[0] = padd(padd(pmul(E[0][0], m0), pmul(E[0][1], m1), -1), pmul(E[0][2], m2));
    }
    // 2 E E^T E - trace(E E^T) E = 0
    {
        Poly3 EEt[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                Poly3 s;
                for (int k = 0; k < 3; ++k) s = padd(s, pmul(E[r][k], E[c][k]));
                EEt[r][c] = s;
            }
        Poly3 tr = padd(padd(EEt[0][0], EEt[1][1]), EEt[2][2]);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                Poly3 s;
                for (int k = 0; k < 3; ++k) s = padd(s, pmul(EEt[r][k], E[k][c]));
                Poly3 t = pmul(tr, E[r][c]);
                Poly3 e;
                for (int i = 0; i < 64; ++i) e.c[i] = 2 * s.c[i] - t.c[i];
                eq[1 + r * 3 + c] = e;
            }
    }
    std::vector<double> A1(100), A2(100);
    for (int r = 0; r < 10; ++r)
        for (int m = 0; m < 20; ++m) {
            double v = eq[r].c[Poly3::id(kMono[m][0], kMono[m][1], kMono[m][2])];
            if (m < 10) A1[r * 10 + m] = v;
            else A2[r * 10 + (m - 10)] = v;
        }
    if (!lu_solve(A1, 10, A2, 10)) return 0;
    // rows 4..9 of A2 belong to x^2 z, x^2, y^2 z, y^2, xyz, xy.
    // z*row(m) - row(m*z) gives  x*p3(z) + y*q3(z) + r4(z) = 0.
    double b[3][13];  // [x: z^3..1 | y: z^3..1 | 1: z^4..1]
    for (int i = 0; i < 3; ++i) {
        const double* rz = &A2[(4 + 2 * i) * 10];  // leading monomial * z
        const double* r1 = &A2[(5 + 2 * i) * 10];  // leading monomial
        double* o = b[i];
        for (int k = 0; k < 13; ++k) o[k] = 0;
        // r1 * z: x z^2,x z,x -> x z^3,x z^2,x z ; y likewise ; z^3..1 -> z^4..z
        for (int k = 0; k < 3; ++k) { o[k] += r1[k]; o[4 + k] += r1[3 + k]; }
        for (int k = 0; k < 4; ++k) o[8 + k] += r1[6 + k];
        for (int k = 0; k < 3; ++k) { o[1 + k] -= rz[k]; o[5 + k] -= rz[3 + k]; }
        for (int k = 0; k < 4; ++k) o[9 + k] -= rz[6 + k];
    }
    // det of the 3x3 polynomial matrix -> degree 10 in z (ascending coeffs)
    auto pz = [&](int row, int col, double* out, int* deg) {
        // descending storage in b -> ascending out
        if (col < 2) { *deg = 3; for (int k = 0; k < 4; ++k) out[k] = b[row][col * 4 + 3 - k]; }
        else { *deg = 4; for (int k = 0; k < 5; ++k) out[k] = b[row][8 + 4 - k]; }
    };
    auto mulz = [](const double* a, int da, const double* bb, int db, double* o) {
        for (int i = 0; i <= da + db; ++i) o[i] = 0;
        for (int i = 0; i <= da; ++i)
            for (int j = 0; j <= db; ++j) o[i + j] += a[i] * bb[j];
    };
    double c[11] = {0};
    const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
    const double sgn[6] = {1, 1, 1, -1, -1, -1};
    for (int p = 0; p < 6; ++p) {
        double f0[5], f1[5], f2[5], t01[9], t[13];
        int d0, d1, d2;
        pz(0, perm[p][0], f0, &d0);
        pz(1, perm[p][1], f1, &d1);
        pz(2, perm[p][2], f2, &d2);
        mulz(f0, d0, f1, d1, t01);
        mulz(t01, d0 + d1, f2, d2, t);
        for (int i = 0; i <= d0 + d1 + d2; ++i) c[i] += sgn[p] * t[i];
    }
    std::vector<std::complex<double>> roots = solve_poly(c, 10);
    if (dbg) {
        for (int i = 0; i < 11; ++i) dbg[i] = c[i];
        for (size_t i = 0; i < roots.size() && i < 10; ++i) { dbg[11 + 2 * i] = roots[i].real(); dbg[12 + 2 * i] = roots[i].imag(); }
        dbg[31] = (double)roots.size();
    }
    int count = 0;
    for (auto& rt : roots) {
        // OpenCV keeps a root only if |imag| <= 1e-10.  Durand-Kerner is often short of
        // that on clustered roots after its 300 sweeps, so the gate here is loose and the
        // refinement below (which rejects starts that do not settle) decides instead.
        if (g_five_point_literal ? std::fabs(rt.imag()) > 1e-10
                                 : std::fabs(rt.imag()) > 1e-3 * std::max(1.0, std::fabs(rt.real())))
            continue;
        double z1 = rt.real(), z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; ++j) {
            const double* br = b[j];
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        // SVD::solveZ: right singular vector of the smallest singular value
        double sw[3], sU[9], sV[9];
        jacobi_svd(bz, 3, 3, sw, sU, sV);
        double xy1[3] = {sV[0 * 3 + 2], sV[1 * 3 + 2], sV[2 * 3 + 2]};
        if (std::fabs(xy1[2]) < 1e-10) continue;
        double xyz[3] = {xy1[0] / xy1[2], xy1[1] / xy1[2], z1};
        double Ev[9];
        if (g_five_point_literal) {            // E from the root as it stands, unit norm
            double nrm = 0;
            for (int k = 0; k < 9; ++k) {
                Ev[k] = EE[0][k] * xyz[0] + EE[1][k] * xyz[1] + EE[2][k] * xyz[2] + EE[3][k];
                nrm += Ev[k] * Ev[k];
            }
            nrm = std::sqrt(nrm);
            for (int k = 0; k < 9; ++k) Es[count * 9 + k] = Ev[k] / nrm;
            ++count;
            if (count == 10) break;
            continue;
        }
        double res = polish_model(EE, xyz, Ev);
        if (!(res <= 1e-12)) continue;         // did not settle on a solution
        bool dup = false;                      // two starts that settled on the same solution
        for (int m = 0; m < count && !dup; ++m) {
            double dp = 0, dm = 0;
            for (int k = 0; k < 9; ++k) {
                dp = std::max(dp, std::fabs(Es[m * 9 + k] - Ev[k]));
                dm = std::max(dm, std::fabs(Es[m * 9 + k] + Ev[k]));
            }
            dup = std::min(dp, dm) < 1e-7;
        }
        if (dup) continue;
        for (int k = 0; k < 9; ++k) Es[count * 9 + k] = Ev[k];
        ++count;
        if (count == 10) break;
    }
    return count;
}

// Number of null-space charts the minimal solver is run in (1 = OpenCV's own chart only).
// One chart loses true solutions on a few percent of the minimal samples of hard scenes:
// when its 10x10 elimination is ill-conditioned the hidden-variable polynomial is wrong and
// real roots go missing or land too far from a solution for the refinement.  OpenCV loses them
// too, in a way that depends on its exact arithmetic; what any two correct solvers can agree
// on is the complete solution set, so the checker takes the union over several charts
// (duplicates dropped).  The CPU-baseline timing sets this to 1 to keep OpenCV's cost.
static int g_five_point_charts = 4;

int five_point(const double* x1, const double* x2, double* Es, double* dbg = nullptr) {
    int count = five_point_chart(x1, x2, 0, Es, dbg);
    for (int c = 1; c < g_five_point_charts && count < 10 && !g_five_point_literal; ++c) {
        double more[90];
        int n = five_point_chart(x1, x2, c, more);
        for (int i = 0; i < n && count < 10; ++i) {
            bool dup = false;
            for (int m = 0; m < count && !dup; ++m) {
                double dp = 0, dm = 0;
                for (int k = 0; k < 9; ++k) {
                    dp = std::max(dp, std::fabs(Es[m * 9 + k] - more[i * 9 + k]));
                    dm = std::max(dm, std::fabs(Es[m * 9 + k] + more[i * 9 + k]));
                }
                dup = std::min(dp, dm) < 1e-7;
            }
            if (dup) continue;
            for (int k = 0; k < 9; ++k) Es[count * 9 + k] = more[i * 9 + k];
            ++count;
        }
    }
    return count;
}

// EMEstimatorCallback::computeError: Sampson distance, cast to float.
inline float sampson_err(const double* E, double x1, double y1, double x2, double y2) {
    double Ex0 = E[0] * x1 + E[1] * y1 + E[2];
    double Ex1 = E[3] * x1 + E[4] * y1 + E[5];
    double Ex2 = E[6] * x1 + E[7] * y1 + E[8];
    double Et0 = E[0] * x2 + E[3] * y2 + E[6];
    double Et1 = E[1] * x2 + E[4] * y2 + E[7];
    double x2tEx1 = x2 * Ex0 + y2 * Ex1 + Ex2;
    double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

struct CvRng {  // cv::RNG: multiply-with-carry
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffu) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

void normalise_points(const float* p, int n, const double K[4], std::vector<double>& out) {
    // (x - cx)/fx evaluated as x*(1/fx) + (-cx*(1/fx)), as a cv::MatExpr does
    double ifx = 1. / K[0], ify = 1. / K[1];
    double bx = -K[2] * ifx, by = -K[3] * ify;
    out.resize((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        out[i * 2] = (double)p[i * 2] * ifx + bx;
        out[i * 2 + 1] = (double)p[i * 2 + 1] * ify + by;
    }
}

// A total order on essential matrices that does not depend on how they were found: E and -E are
// one model (the sign that makes the first entry of at least half the largest magnitude positive
// is compared), entries are compared in storage order, differences below 1e-9 count as equal.
bool model_before(const double* A, const double* B) {
    auto sign_of = [](const double* E) {
        double m = 0;
        for (int i = 0; i < 9; ++i) m = std::max(m, std::fabs(E[i]));
        for (int i = 0; i < 9; ++i)
            if (std::fabs(E[i]) > 0.5 * m) return E[i] < 0 ? -1.0 : 1.0;
        return 1.0;
    };
    const double sa = sign_of(A), sb = sign_of(B);
    for (int i = 0; i < 9; ++i) {
        const double d = sa * A[i] - sb * B[i];
        if (std::fabs(d) > 1e-9) return d < 0;
    }
    return false;
}

// cv::findEssentialMat(..., RANSAC, prob, threshold, maxIters=1000, mask)
// -> RANSACPointSetRegistrator::run.  fixed_iters > 0 disables the adaptive
// iteration count and evaluates exactly that many hypotheses.
// samples_out (optional, max_iters*5 ints) receives the index sets drawn.
int find_essential(const float* p1, const float* p2, int n, const double K[4], double threshold,
                   double confidence, int max_iters, int fixed_iters, double E[9], uint8_t* mask,
                   int* iters_run, int32_t* samples_out) {
    if (iters_run) *iters_run = 0;
    if (n < 5) return 0;
    std::vector<double> m1, m2;
    normalise_points(p1, n, K, m1);
    normalise_points(p2, n, K, m2);
    threshold /= (K[0] + K[1]) / 2;
    float t = (float)(threshold * threshold);
    int niters = std::max(fixed_iters > 0 ? fixed_iters : max_iters, 1);
    int maxGood = 0;
    std::vector<uint8_t> cur(n), best(n, 0);
    double bestE[9] = {0};
    CvRng rng((uint64_t)-1);
    double Es[90];
    if (n == 5) {
        // count == modelPoints: one pass of the loop; OpenCV returns every model stacked, only a
        // single-model answer is usable downstream.
        if (iters_run) *iters_run = 1;
        if (samples_out) for (int i = 0; i < 5; ++i) samples_out[i] = i;   // the five points as they come, nothing is drawn
        int nm = five_point(m1.data(), m2.data(), Es);
        if (nm != 1) return 0;
        memcpy(E, Es, sizeof(double) * 9);
        if (mask) memset(mask, 1, n);
        return n;
    }
    int iter = 0;
    for (; iter < niters; ++iter) {
        int idx[5];
        double s1[10], s2[10];
        for (int i = 0; i < 5; ++i) {
            int v;
            for (v = rng.uniform(0, n); std::find(idx, idx + i, v) != idx + i; v = rng.uniform(0, n)) {}
            idx[i] = v;
            s1[i * 2] = m1[v * 2]; s1[i * 2 + 1] = m1[v * 2 + 1];
            s2[i * 2] = m2[v * 2]; s2[i * 2 + 1] = m2[v * 2 + 1];
        }
        if (samples_out) for (int i = 0; i < 5; ++i) samples_out[iter * 5 + i] = idx[i];
        int nm = five_point(s1, s2, Es);
        // OpenCV walks the hypothesis' models in the order its polynomial solver happens to emit
        // them and keeps the first with the most inliers.  That order is not reproducible (here the
        // models come from several charts), so among a hypothesis' models with equally many inliers
        // the one that comes first in a chart-independent order (model_before) is taken; the
        // iteration bound only depends on the count.
        int top = -1, top_k = -1;
        std::vector<int> goods(nm);
        for (int k = 0; k < nm; ++k) {
            const double* Ek = Es + k * 9;
            int good = 0;
            for (int i = 0; i < n; ++i)
                good += sampson_err(Ek, m1[i * 2], m1[i * 2 + 1], m2[i * 2], m2[i * 2 + 1]) <= t;
            goods[k] = good;
            if (good > top || (good == top && model_before(Ek, Es + top_k * 9))) {
                top = good;
                top_k = k;
            }
        }
        if (top > std::max(maxGood, 4)) {
            const double* Ek = Es + top_k * 9;
            for (int i = 0; i < n; ++i)
                best[i] = sampson_err(Ek, m1[i * 2], m1[i * 2 + 1], m2[i * 2], m2[i * 2 + 1]) <= t;
            memcpy(bestE, Ek, sizeof bestE);
            maxGood = top;
            if (fixed_iters <= 0)
                niters = ransac_update_iters(confidence, (double)(n - top) / n, 5, niters);
        }
    }
    if (iters_run) *iters_run = iter;
    if (maxGood > 0) {
        memcpy(E, bestE, sizeof bestE);
        if (mask) memcpy(mask, best.data(), n);
    }
    return maxGood;
}

void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}
double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// cv::decomposeEssentialMat
void decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double w[3], U[9], V[9];
    jacobi_svd(E, 3, 3, w, U, V);
    // U's third column: sigma_3 ~ 0 leaves it undefined -> take u1 x u2
    double u1[3] = {U[0], U[3], U[6]}, u2[3] = {U[1], U[4], U[7]};
    double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    U[2] = u3[0]; U[5] = u3[1]; U[8] = u3[2];
    double Vt[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Vt[i * 3 + j] = V[j * 3 + i];
    if (det3(U) < 0) for (double& x : U) x = -x;
    if (det3(Vt) < 0) for (double& x : Vt) x = -x;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mat3_mul(U, W, T); mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T); mat3_mul(T, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

// cv::recoverPose(E, p1, p2, K, R, t, distanceThresh = 50, mask)
int recover_pose(const double* E, const float* p1, const float* p2, int n, const double K[4],
                 uint8_t* mask, double* R, double* t) {
    const double dist = 50.0;
    std::vector<double> m1, m2;
    normalise_points(p1, n, K, m1);
    normalise_points(p2, n, K, m2);
    double R1[9], R2[9], tt[3];
    decompose_essential(E, R1, R2, tt);
    const double* Rs[4] = {R1, R2, R1, R2};
    const double ts[4] = {1, 1, -1, -1};
    std::vector<std::vector<uint8_t>> good(4, std::vector<uint8_t>(n, 0));
    int cnt[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        double P[12];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) P[i * 4 + j] = Rs[c][i * 3 + j];
            P[i * 4 + 3] = ts[c] * tt[i];
        }
        for (int i = 0; i < n; ++i) {
            double x1 = m1[i * 2], y1 = m1[i * 2 + 1], x2 = m2[i * 2], y2 = m2[i * 2 + 1];
            // cv::triangulatePoints: 4x4 DLT, smallest right singular vector
            double A[16];
            const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            for (int k = 0; k < 4; ++k) {
                A[0 * 4 + k] = x1 * P0[8 + k] - P0[k];
                A[1 * 4 + k] = y1 * P0[8 + k] - P0[4 + k];
                A[2 * 4 + k] = x2 * P[8 + k] - P[k];
                A[3 * 4 + k] = y2 * P[8 + k] - P[4 + k];
            }
            double w4[4], U4[16], V4[16];
            jacobi_svd(A, 4, 4, w4, U4, V4);
            double X[4] = {V4[3], V4[7], V4[11], V4[15]};
            bool ok = X[2] * X[3] > 0;
            double Xn[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};
            ok = ok && (Xn[2] < dist);
            double z2 = P[8] * Xn[0] + P[9] * Xn[1] + P[10] * Xn[2] + P[11];
            ok = ok && (z2 > 0) && (z2 < dist);
            if (mask) ok = ok && mask[i];
            good[c][i] = ok;
            cnt[c] += ok;
        }
    }
    int sel;
    if (cnt[0] >= cnt[1] && cnt[0] >= cnt[2] && cnt[0] >= cnt[3]) sel = 0;
    else if (cnt[1] >= cnt[0] && cnt[1] >= cnt[2] && cnt[1] >= cnt[3]) sel = 1;
    else if (cnt[2] >= cnt[0] && cnt[2] >= cnt[1] && cnt[2] >= cnt[3]) sel = 2;
    else sel = 3;
    memcpy(R, Rs[sel], sizeof(double) * 9);
    for (int i = 0; i < 3; ++i) t[i] = ts[sel] * tt[i];
    if (mask) memcpy(mask, good[sel].data(), n);
    return cnt[sel];
}

}  // namespace

// ---------------------------------------------------------------- C entry points
extern "C" {

int oracle_level_geometry(int rows, int cols, int level, int* w, int* h, float* scale) {
    level_geometry(cols, rows, level, w, h, scale);
    return 0;
}

// Writes pyramid level `level` (blurred != 0: after the descriptor blur) into
// out (w*h bytes, tightly packed).
int oracle_pyramid_level(const uint8_t* img, int rows, int cols, int step, int level, int blurred,
                         uint8_t* out) {
    std::vector<Level> pyr = build_pyramid(img, rows, cols, step);
    Level& L = pyr[level];
    if (blurred) { blur_level(L); memcpy(out, L.blur.data(), L.blur.size()); }
    else memcpy(out, L.img.data(), L.img.size());
    return 0;
}

// FAST-9/16 + NMS on one image; out: (x,y,score) int triples in raster order.
int oracle_fast(const uint8_t* img, int rows, int cols, int step, int threshold, int32_t* out, int cap) {
    std::vector<Cand> c = fast_detect(img, cols, rows, step, threshold);
    int n = (int)std::min<size_t>(c.size(), (size_t)cap);
    for (int i = 0; i < n; ++i) { out[i * 3] = c[i].x; out[i * 3 + 1] = c[i].y; out[i * 3 + 2] = (int)c[i].response; }
    return (int)c.size();
}

void oracle_features_per_level(int nfeatures, int* n8) { features_per_level(nfeatures, n8); }

float oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }

int oracle_orb_detect(const uint8_t* img, int rows, int cols, int step, int max_features, int want_desc,
                      int order_mode, void* kps_out, uint8_t* desc_out, int cap) {
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    int n = orb_detect(img, rows, cols, step, max_features, want_desc, order_mode, k, d);
    int m = std::min(n, cap);
    if (kps_out && m > 0) memcpy(kps_out, k.data(), sizeof(KeyPoint) * m);
    if (desc_out && want_desc && m > 0) memcpy(desc_out, d.data(), (size_t)32 * m);
    return n;
}

void oracle_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2) {
    knn2(q, nq, t, nt, idx2, dist2);
}

// match_features (ref:src/feature_matcher.cpp:15-42).  Returns the number of
// matches; query/train index and distance per match in query order.
int oracle_match(const uint8_t* q, int nq, const uint8_t* t, int nt, float ratio, int32_t* qidx,
                 int32_t* tidx, float* dist) {
    if (nq <= 0 || nt <= 0) return 0;
    std::vector<int32_t> i2((size_t)nq * 2), d2((size_t)nq * 2);
    knn2(q, nq, t, nt, i2.data(), d2.data());
    int m = 0;
    for (int i = 0; i < nq; ++i) {
        if (i2[i * 2] < 0 || i2[i * 2 + 1] < 0) continue;
        float a = (float)d2[i * 2], b = (float)d2[i * 2 + 1];
        if (a < ratio * b) { qidx[m] = i; tidx[m] = i2[i * 2]; dist[m] = a; ++m; }
    }
    return m;
}

int oracle_five_point(const double* x1, const double* x2, double* Es) { return five_point(x1, x2, Es); }
void oracle_set_five_point_charts(int n) { g_five_point_charts = n < 1 ? 1 : n; }
// trig_float != 0: descriptor rotation through libm's cosf / sinf; five_point_literal != 0: the
// minimal solver as OpenCV writes it (see the switches' declarations).  Both 0 by default.
void oracle_set_variant(int trig_float, int five_point_literal) {
    g_trig_float = trig_float != 0;
    g_five_point_literal = five_point_literal != 0;
}

// Same, also returning the degree-10 polynomial (dbg[0..10], ascending), the raw
// Durand-Kerner roots (dbg[11..30] as re,im pairs) and their number (dbg[31]).
int oracle_five_point_debug(const double* x1, const double* x2, double* Es, double* dbg) {
    return five_point(x1, x2, Es, dbg);
}

int oracle_find_essential(const float* p1, const float* p2, int n, const double* K, double threshold,
                          double confidence, int max_iters, int fixed_iters, double* E, uint8_t* mask,
                          int* iters_run, int32_t* samples_out) {
    return find_essential(p1, p2, n, K, threshold, confidence, max_iters, fixed_iters, E, mask, iters_run,
                          samples_out);
}

int oracle_recover_pose(const double* E, const float* p1, const float* p2, int n, const double* K,
                        uint8_t* mask, double* R, double* t) {
    return recover_pose(E, p1, p2, n, K, mask, R, t);
}

// estimate_motion (ref:src/motion_estimator.cpp:44-95).  Returns valid (0/1).
int oracle_estimate_motion(const float* p1, const float* p2, int n, const double* K, double threshold,
                           double confidence, int fixed_iters, double* R, double* t, int* inliers,
                           uint8_t* mask_out) {
    *inliers = 0;
    if (n < 5) return 0;
    double E[9];
    std::vector<uint8_t> mask(n, 0);
    int good = find_essential(p1, p2, n, K, threshold, confidence, 1000, fixed_iters, E, mask.data(), nullptr,
                              nullptr);
    if (good <= 0) return 0;
    int cnt = recover_pose(E, p1, p2, n, K, mask.data(), R, t);
    if (mask_out) memcpy(mask_out, mask.data(), n);
    if (cnt < 10) return 0;
    *inliers = cnt;
    return 1;
}

}  // extern "C"
