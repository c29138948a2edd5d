// ORB on gfx950: pyramid (fixed-point bilinear), FAST-9/16 + NMS, score/Harris
// selection, intensity-centroid angle, 7x7 fixed-point blur, rBRIEF.
// Stands in for cv::ORB::detectAndCompute as called by
// ref:src/feature_detector.cpp:12-27.  All integer stages are exact; float
// stages keep OpenCV's operation order (built with -ffp-contract=off).
#include <cfloat>
#include <cmath>

#include "vo_internal.h"

namespace vo {

static const int8_t h_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__device__ int8_t d_pattern[256 * 4];
static bool g_pattern_loaded[64] = {false};

struct SmallTabs {
    int umax[16];
    int taps[7];
};

__device__ __forceinline__ const uint8_t* level_base(const Geom& g, const uint8_t* in, const uint8_t* pyr, int img,
                                                     int l, int* pitch) {
    if (l == 0) {
        *pitch = g.step;
        return in + (size_t)img * g.in_stride;
    }
    *pitch = g.L[l].pitch;
    return pyr + (size_t)img * g.blk_stride + g.L[l].off;
}

// ------------------------------------------------------------------ resize
// Level l from level l-1, INTER_LINEAR_EXACT semantics: 8.8 coefficients,
// out = (m0*(c0*a + c1*b) + m1*(c0*c + c1*d) + 2^15) >> 16.
// Table entry = (source offset << 16) | c1, c0 = 256 - c1.
__global__ __launch_bounds__(256) void resize_kernel(Geom g, int l, const uint8_t* __restrict__ in,
                                                     uint8_t* __restrict__ pyr, const uint32_t* __restrict__ tab) {
    const LevelGeom& D = g.L[l];
    const LevelGeom& S = g.L[l - 1];
    int img = blockIdx.z;
    int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= D.h || x4 >= D.w) return;
    int spitch;
    const uint8_t* src = level_base(g, in, pyr, img, l - 1, &spitch);
    uint8_t* dst = pyr + (size_t)img * g.blk_stride + D.off + (size_t)y * D.pitch;
    uint32_t ty = tab[D.ytab + y];
    int sy = ty >> 16;
    uint32_t m1 = ty & 0xFFFF, m0 = 256 - m1;
    int sy1 = min(sy + 1, S.h - 1);
    const uint8_t* r0 = src + (size_t)sy * spitch;
    const uint8_t* r1 = src + (size_t)sy1 * spitch;
    uint32_t outw = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int x = x4 + k;
        uint32_t v = 0;
        if (x < D.w) {
            uint32_t tx = tab[D.xtab + x];
            int sx = tx >> 16;
            uint32_t c1 = tx & 0xFFFF, c0 = 256 - c1;
            int sx1 = min(sx + 1, S.w - 1);
            uint32_t h0 = c0 * r0[sx] + c1 * r0[sx1];
            uint32_t h1 = c0 * r1[sx] + c1 * r1[sx1];
            v = (m0 * h0 + m1 * h1 + 32768u) >> 16;
            v = min(v, 255u);
        }
        outw |= v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + x4) = outw;  // pitch is a multiple of 64: the store stays inside the row
}

// ------------------------------------------------------------------ FAST
constexpr int FT_W = 64, FT_H = 32;
constexpr int FP_W = 80, FP_H = FT_H + 8;   // pixel tile (dword-aligned origin, 4-px halo)
constexpr int FS_W = 68, FS_H = FT_H + 2;   // score tile (1-px halo)

__device__ __forceinline__ bool has_arc9(unsigned m) {
    m |= m << 16;
    unsigned a = m & (m >> 1);
    a &= a >> 2;
    a &= a >> 4;
    a &= m >> 8;
    return (a & 0xFFFFu) != 0;
}

// FAST-9/16 corner score at LDS pixel p: 0 if no 9-arc is beyond the
// threshold, else max over arcs of the arc's minimum contrast, minus 1.
__device__ __forceinline__ int fast_score_lds(const uint8_t* p) {
    constexpr int P = FP_W;
    const int off[16] = {3 * P,      3 * P + 1,  2 * P + 2,  P + 3,  3,       -P + 3,     -2 * P + 2, -3 * P + 1,
                         -3 * P,     -3 * P - 1, -2 * P - 2, -P - 3, -3,      P - 3,      2 * P - 2,  3 * P - 1};
    int v = p[0];
    int d[16];
    unsigned bright = 0, dark = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        d[k] = v - (int)p[off[k]];
        bright |= (unsigned)(d[k] > kFastThr) << k;
        dark |= (unsigned)(d[k] < -kFastThr) << k;
    }
    bool hb = has_arc9(bright), hd = has_arc9(dark);
    if (!hb && !hd) return 0;
    int mn[16], mx[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        mn[i] = min(d[i], d[(i + 1) & 15]);
        mx[i] = max(d[i], d[(i + 1) & 15]);
    }
    int mn4[16], mx4[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        mn4[i] = min(mn[i], mn[(i + 2) & 15]);
        mx4[i] = max(mx[i], mx[(i + 2) & 15]);
    }
    int A = -1000, B = 1000;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int a8 = min(mn4[i], mn4[(i + 4) & 15]);
        int b8 = max(mx4[i], mx4[(i + 4) & 15]);
        A = max(A, min(a8, d[(i + 8) & 15]));
        B = min(B, max(b8, d[(i + 8) & 15]));
    }
    return max(A, -B) - 1;
}

__global__ __launch_bounds__(256) void fast_kernel(Geom g, const uint8_t* __restrict__ in,
                                                   const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ cand_cnt) {
    __shared__ __attribute__((aligned(16))) uint8_t px[FP_H * FP_W];
    __shared__ uint8_t sc[FS_H * FS_W];
    __shared__ uint32_t list[FT_W * FT_H / 4];
    __shared__ uint32_t nlist, gbase;

    int tile = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    int l = 0;
#pragma unroll
    for (int k = 0; k < kLevels; ++k) {
        int t0 = g.L[k].fast_tile0;
        if (tile >= t0 && tile < t0 + g.L[k].fast_tx * g.L[k].fast_ty) l = k;
    }
    const LevelGeom& L = g.L[l];
    int tl = tile - L.fast_tile0;
    int x0 = kEdge + (tl % L.fast_tx) * FT_W, y0 = kEdge + (tl / L.fast_tx) * FT_H;
    int pitch;
    const uint8_t* base = level_base(g, in, pyr, img, l, &pitch);
    int ax0 = (x0 - 4) & ~3;   // LDS column 0 <-> image column ax0
    if (tid == 0) nlist = 0;

    bool aligned = ((((uintptr_t)base) | (unsigned)pitch) & 3) == 0;
    if (aligned) {
        int valid_dw = (l == 0 ? (g.cols & ~3) : pitch) >> 2;
        for (int i = tid; i < FP_H * (FP_W / 4); i += 256) {
            int r = i / (FP_W / 4), c = i - r * (FP_W / 4);
            int y = min(max(y0 - 4 + r, 0), L.h - 1);
            int dw = min((ax0 >> 2) + c, valid_dw - 1);
            uint32_t v = *reinterpret_cast<const uint32_t*>(base + (size_t)y * pitch + 4 * dw);
            *reinterpret_cast<uint32_t*>(&px[r * FP_W + 4 * c]) = v;
        }
    } else {
        for (int i = tid; i < FP_H * FP_W; i += 256) {
            int r = i / FP_W, c = i - r * FP_W;
            int y = min(max(y0 - 4 + r, 0), L.h - 1);
            int x = min(ax0 + c, L.w - 1);
            px[i] = base[(size_t)y * pitch + x];
        }
    }
    __syncthreads();

    // scores on the (FT_W+2) x (FT_H+2) halo region
    for (int i = tid; i < FS_H * (FT_W + 2); i += 256) {
        int sy = i / (FT_W + 2), sx = i - sy * (FT_W + 2);
        int X = x0 - 1 + sx, Y = y0 - 1 + sy;
        int s = 0;
        if (X <= L.w - kEdge && Y <= L.h - kEdge) s = fast_score_lds(&px[(sy + 3) * FP_W + (X - ax0)]);
        sc[sy * FS_W + sx] = (uint8_t)s;
    }
    __syncthreads();

    // 3x3 non-max suppression + border filter on the interior
    for (int i = tid; i < FT_W * FT_H; i += 256) {
        int iy = i / FT_W, ix = i - iy * FT_W;
        int X = x0 + ix, Y = y0 + iy;
        const uint8_t* s = &sc[(iy + 1) * FS_W + ix + 1];
        int v = s[0];
        bool keep = v > 0 && X < L.w - kEdge && Y < L.h - kEdge;
        if (keep) {
            keep = v > s[-1] && v > s[1] && v > s[-FS_W - 1] && v > s[-FS_W] && v > s[-FS_W + 1] && v > s[FS_W - 1] &&
                   v > s[FS_W] && v > s[FS_W + 1];
        }
        if (keep) {
            uint32_t slot = atomicAdd(&nlist, 1u);
            list[slot] = ((uint32_t)v << 24) | ((uint32_t)Y << 12) | (uint32_t)X;
        }
    }
    __syncthreads();
    uint32_t n = nlist;
    if (n == 0) return;
    if (tid == 0) gbase = atomicAdd(&cand_cnt[img * kLevels + l], n);
    __syncthreads();
    uint32_t* out = cand + (size_t)img * g.cand_stride + L.cand_off;
    for (uint32_t i = tid; i < n; i += 256) {
        uint32_t pos = gbase + i;
        if (pos < (uint32_t)L.cand_cap) out[pos] = list[i];
    }
}

// ------------------------------------------------------------------ select
__device__ __forceinline__ uint32_t f2sortable(float f) {
    if (f == 0.f) f = 0.f;
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ __forceinline__ float harris_at(const uint8_t* img, int pitch, int x0, int y0) {
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = -3; i <= 3; ++i) {
        const uint8_t* rm = img + (size_t)(y0 + i - 1) * pitch + x0;
        const uint8_t* r0 = rm + pitch;
        const uint8_t* rp = r0 + pitch;
#pragma unroll
        for (int j = -3; j <= 3; ++j) {
            int Ix = ((int)r0[j + 1] - (int)r0[j - 1]) * 2 + ((int)rm[j + 1] - (int)rm[j - 1]) +
                     ((int)rp[j + 1] - (int)rp[j - 1]);
            int Iy = ((int)rp[j] - (int)rm[j]) * 2 + ((int)rp[j - 1] - (int)rm[j - 1]) +
                     ((int)rp[j + 1] - (int)rm[j + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    }
    float fa = (float)a, fb = (float)b, fc = (float)c;
    float t1 = fa * fb;
    float t2 = fc * fc;
    float s = fa + fb;
    float t3 = 0.04f * s;
    t3 = t3 * s;
    float v = t1 - t2;
    v = v - t3;
    return v * scale_sq_sq;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y);
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

__device__ __forceinline__ float ic_angle_at(const uint8_t* img, int pitch, int x0, int y0, const int* umax) {
    const uint8_t* center = img + (size_t)y0 * pitch + x0;
    int m01 = 0, m10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int vsum = 0, d = umax[v];
        const uint8_t* rp = center + v * pitch;
        const uint8_t* rm = center - v * pitch;
        for (int u = -d; u <= d; ++u) {
            int vp = rp[u], vm = rm[u];
            vsum += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return fast_atan2_deg((float)m01, (float)m10);
}

template <typename T>
__device__ void bitonic_sort_asc(T* s, int P, int tid, int nt) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nt) {
                int ixj = i ^ j;
                if (ixj > i) {
                    T a = s[i], b = s[ixj];
                    bool asc = (i & k) == 0;
                    if ((a > b) == asc) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
}

// One workgroup per (level, image): retainBest(2n) on the FAST score, raster
// order, Harris response, retainBest(n) on it, angle.  "retainBest(n)" keeps
// every element whose value is >= the n-th largest (ties survive).
template <int CAP>
__global__ __launch_bounds__(512) void select_kernel(Geom g, SmallTabs tabs, const uint8_t* __restrict__ in,
                                                     const uint8_t* __restrict__ pyr,
                                                     const uint32_t* __restrict__ cand,
                                                     const uint32_t* __restrict__ cand_cnt,
                                                     uint16_t* __restrict__ kp_x, uint16_t* __restrict__ kp_y,
                                                     float* __restrict__ kp_resp, float* __restrict__ kp_angle,
                                                     uint32_t* __restrict__ kp_cnt, uint32_t* __restrict__ flags) {
    __shared__ uint32_t keys[CAP];
    __shared__ float resp[CAP];
    __shared__ uint32_t srt[CAP];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_n, s_thr;
    __shared__ int s_umax[16];

    const int l = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    constexpr int NT = 512;
    const LevelGeom& L = g.L[l];
    uint32_t total = min(cand_cnt[img * kLevels + l], (uint32_t)L.cand_cap);
    const uint32_t* cl = cand + (size_t)img * g.cand_stride + L.cand_off;
    const int quota = L.quota;
    if (tid < 256) hist[tid] = 0;
    if (tid < 16) s_umax[tid] = tabs.umax[tid];
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (total == 0 || quota == 0) {
        if (tid == 0) kp_cnt[img * kLevels + l] = 0;
        return;
    }
    // --- retainBest(2*quota) by FAST score
    uint32_t thr_score = 0;
    if (total > (uint32_t)(2 * quota)) {
        for (uint32_t i = tid; i < total; i += NT) atomicAdd(&hist[cl[i] >> 24], 1u);
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            int s = 255;
            for (; s > 0; --s) {
                acc += hist[s];
                if (acc >= (uint32_t)(2 * quota)) break;
            }
            s_thr = (uint32_t)s;
        }
        __syncthreads();
        thr_score = s_thr;
    }
    for (uint32_t i = tid; i < total; i += NT) {
        uint32_t c = cl[i];
        if ((c >> 24) >= thr_score) {
            uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < (uint32_t)CAP) keys[slot] = c & 0x00FFFFFFu;
        }
    }
    __syncthreads();
    uint32_t M = s_n;
    if (M > (uint32_t)CAP) {
        if (tid == 0) atomicOr(flags, 1u);
        M = CAP;
    }
    int P = 1;
    while (P < (int)M) P <<= 1;
    for (int i = M + tid; i < P; i += NT) keys[i] = 0xFFFFFFFFu;
    __syncthreads();
    bitonic_sort_asc(keys, P, tid, NT);   // raster order: key = y << 12 | x

    int pitch;
    const uint8_t* base = level_base(g, in, pyr, img, l, &pitch);
    for (int i = tid; i < (int)M; i += NT) {
        uint32_t k = keys[i];
        resp[i] = harris_at(base, pitch, k & 0xFFF, k >> 12);
    }
    __syncthreads();
    // --- retainBest(quota) by Harris response
    uint32_t inv_thr = 0xFFFFFFFFu;   // keep iff ~sortable(resp) <= inv_thr
    if (M > (uint32_t)quota) {
        for (int i = tid; i < P; i += NT) srt[i] = i < (int)M ? ~f2sortable(resp[i]) : 0xFFFFFFFFu;
        __syncthreads();
        bitonic_sort_asc(srt, P, tid, NT);
        inv_thr = srt[quota - 1];
        __syncthreads();
    }
    // ordered compaction: per-thread contiguous chunk, then a scan of chunk totals
    int chunk = ((int)M + NT - 1) / NT;
    int beg = min(tid * chunk, (int)M), end = min(beg + chunk, (int)M);
    int mine = 0;
    for (int i = beg; i < end; ++i) mine += (~f2sortable(resp[i]) <= inv_thr);
    srt[tid] = mine;   // srt is free again
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < NT; ++i) {
            uint32_t v = srt[i];
            srt[i] = acc;
            acc += v;
        }
        s_n = acc;
    }
    __syncthreads();
    uint32_t F = s_n;
    uint32_t pos = srt[tid];
    __syncthreads();
    if (F > (uint32_t)L.kp_cap) {
        if (tid == 0) atomicOr(flags, 2u);
        F = L.kp_cap;
    }
    // reuse srt as the compacted index list
    for (int i = beg; i < end; ++i)
        if (~f2sortable(resp[i]) <= inv_thr) {
            if (pos < F) srt[pos] = i;
            ++pos;
        }
    __syncthreads();
    size_t ob = (size_t)img * g.kp_stride + L.kp_off;
    for (int j = tid; j < (int)F; j += NT) {
        int i = srt[j];
        uint32_t k = keys[i];
        int x = k & 0xFFF, y = k >> 12;
        kp_x[ob + j] = (uint16_t)x;
        kp_y[ob + j] = (uint16_t)y;
        kp_resp[ob + j] = resp[i];
        kp_angle[ob + j] = ic_angle_at(base, pitch, x, y, s_umax);
    }
    if (tid == 0) kp_cnt[img * kLevels + l] = F;
}

// ------------------------------------------------------------------ blur
// 7x7 separable fixed-point blur (taps sum to 257, 8 fractional bits each),
// BORDER_REFLECT_101.  Column stage rounds sum/65536 half-to-even for columns
// below (w & ~3) and half-up for the rest, saturating to 255.
constexpr int BT_W = 64, BT_H = 16;
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void blur_kernel(Geom g, SmallTabs tabs, const uint8_t* __restrict__ in,
                                                   const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
    __shared__ uint8_t px[(BT_H + 6) * (BT_W + 8)];
    __shared__ uint16_t hs[(BT_H + 6) * BT_W];
    int tile = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    int l = 0;
#pragma unroll
    for (int k = 0; k < kLevels; ++k) {
        int t0 = g.L[k].blur_tile0;
        if (tile >= t0 && tile < t0 + g.L[k].blur_tx * g.L[k].blur_ty) l = k;
    }
    const LevelGeom& L = g.L[l];
    int tl = tile - L.blur_tile0;
    int x0 = (tl % L.blur_tx) * BT_W, y0 = (tl / L.blur_tx) * BT_H;
    int pitch;
    const uint8_t* base = level_base(g, in, pyr, img, l, &pitch);
    constexpr int PW = BT_W + 8;
    for (int i = tid; i < (BT_H + 6) * (BT_W + 6); i += 256) {
        int r = i / (BT_W + 6), c = i - r * (BT_W + 6);
        int y = reflect101(y0 - 3 + r, L.h), x = reflect101(min(x0 - 3 + c, L.w + 2), L.w);
        px[r * PW + c] = base[(size_t)y * pitch + x];
    }
    __syncthreads();
    for (int i = tid; i < (BT_H + 6) * BT_W; i += 256) {
        int r = i / BT_W, c = i - r * BT_W;
        const uint8_t* p = &px[r * PW + c];
        int s = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) s += tabs.taps[k] * p[k];
        hs[i] = (uint16_t)s;
    }
    __syncthreads();
    int vec_end = L.w & ~3;
    uint8_t* out = blur + (size_t)img * g.blk_stride + L.off;   // blur block uses the same offsets
    for (int i = tid; i < BT_H * BT_W; i += 256) {
        int r = i / BT_W, c = i - r * BT_W;
        int X = x0 + c, Y = y0 + r;
        if (X >= L.w || Y >= L.h) continue;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) s += (uint32_t)tabs.taps[k] * hs[(r + k) * BT_W + c];
        uint32_t q = s >> 16, rem = s & 0xFFFFu;
        uint32_t v = X < vec_end ? q + ((rem > 32768u) || (rem == 32768u && (q & 1u))) : (s + 32768u) >> 16;
        out[(size_t)Y * L.pitch + X] = (uint8_t)min(v, 255u);
    }
}

// ------------------------------------------------------------------ describe
// One wavefront per keypoint: lane t evaluates tests t, t+64, t+128, t+192;
// four ballots give the 32 descriptor bytes.  Also writes the final keypoint.
__global__ __launch_bounds__(256) void describe_kernel(Geom g, const uint8_t* __restrict__ blur,
                                                       const uint16_t* __restrict__ kp_x,
                                                       const uint16_t* __restrict__ kp_y,
                                                       const float* __restrict__ kp_resp,
                                                       const float* __restrict__ kp_angle,
                                                       const uint32_t* __restrict__ kp_cnt, vo_keypoint* __restrict__ kps,
                                                       uint8_t* __restrict__ desc, int cap, int32_t* __restrict__ counts,
                                                       uint32_t* __restrict__ flags) {
    int l = blockIdx.y, img = blockIdx.z;
    int lane = threadIdx.x & 63;
    int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t* cnt = kp_cnt + img * kLevels;
    int base_idx = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kLevels; ++k) {
        int c = cnt[k];
        if (k < l) base_idx += c;
        total += c;
    }
    if (l == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        counts[img] = min(total, cap);
        if (total > cap) atomicOr(flags, 4u);
    }
    if (slot >= (int)cnt[l]) return;
    int oi = base_idx + slot;
    if (oi >= cap) return;
    const LevelGeom& L = g.L[l];
    size_t ib = (size_t)img * g.kp_stride + L.kp_off + slot;
    int x = kp_x[ib], y = kp_y[ib];
    float angle = kp_angle[ib];
    float fx = (float)x * L.scale, fy = (float)y * L.scale;
    if (lane == 0) {
        vo_keypoint k;
        k.x = fx;
        k.y = fy;
        k.size = 31 * L.scale;
        k.angle = angle;
        k.response = kp_resp[ib];
        k.octave = l;
        k.class_id = -1;
        kps[(size_t)img * cap + oi] = k;
    }
    if (!desc) return;
    float inv = 1.f / L.scale;
    int cx = __float2int_rn(fx * inv), cy = __float2int_rn(fy * inv);
    float ar = angle * (float)(M_PI / 180.f);
    float a = (float)cos((double)ar), b = (float)sin((double)ar);
    const uint8_t* center = blur + (size_t)img * g.blk_stride + L.off + (size_t)cy * L.pitch + cx;
    const int pitch = L.pitch;
    unsigned long long bits[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int8_t* p = &d_pattern[(k * 64 + lane) * 4];
        float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
        float rx0 = x0 * a - y0 * b, ry0 = x0 * b + y0 * a;
        float rx1 = x1 * a - y1 * b, ry1 = x1 * b + y1 * a;
        int t0 = center[__float2int_rn(ry0) * pitch + __float2int_rn(rx0)];
        int t1 = center[__float2int_rn(ry1) * pitch + __float2int_rn(rx1)];
        bits[k] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        unsigned long long v = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        *reinterpret_cast<unsigned long long*>(desc + ((size_t)img * cap + oi) * 32 + lane * 8) = v;
    }
}

// ------------------------------------------------------------------ host side
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

static void lin_table(int src, int dst, uint32_t* out) {
    double inv_scale = (double)dst / (double)src;
    double scale = 1.0 / inv_scale;
    int maxofs = dst;
    for (int v = 0; v < dst; ++v) {
        double fval = scale * ((double)v + 0.5) - 0.5;
        int ival = (int)std::floor(fval);
        uint32_t e;
        if (ival >= 0 && src > 1) {
            if (ival < src - 1) {
                int c1 = cv_round_d((fval - (double)ival) * 256.0);
                e = ((uint32_t)ival << 16) | (uint32_t)c1;
            } else {
                e = (uint32_t)(src - 1) << 16;
                maxofs = std::min(maxofs, v);
            }
        } else {
            e = 0;   // before the first source sample: source 0 with weight 1
        }
        out[v] = e;
    }
    // beyond maxofs every entry reads the offset of the last column
    for (int v = maxofs; v < dst; ++v) out[v] = out[dst - 1] & 0xFFFF0000u;
}

static SmallTabs make_tabs() {
    SmallTabs t{};
    int v, v0, vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round_d(std::sqrt((double)kHalfPatch * kHalfPatch - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    double k[7], sum = 0;
    for (int i = 0; i < 7; ++i) {
        double x = i - 3;
        k[i] = std::exp(-x * x / 8.0);
        sum += k[i];
    }
    for (int i = 0; i < 7; ++i) t.taps[i] = cv_round_f((float)(k[i] / sum) * 256.f);
    return t;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int orb_prepare(vo_ctx* c, int n_images, int rows, int cols, int step, size_t in_stride, int max_features) {
    Geom& g = c->g;
    bool same = c->geom_valid && g.rows == rows && g.cols == cols && g.step == step && g.in_stride == in_stride &&
                g.max_features == max_features;
    if (!same) {
        if (cols >= 4096 || rows >= 4096) return fail(c, VO_ERR_INVALID, "image dimensions must be below 4096");
        Geom n{};
        n.rows = rows; n.cols = cols; n.step = step; n.in_stride = in_stride; n.max_features = max_features;
        // per-level feature quota (ORB_Impl: nfeaturesPerLevel)
        float factor = (float)(1.0 / (double)1.2f);
        float nd = max_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kLevels));
        int sum = 0, maxq = 0;
        for (int l = 0; l < kLevels - 1; ++l) {
            n.L[l].quota = cv_round_f(nd);
            sum += n.L[l].quota;
            nd *= factor;
        }
        n.L[kLevels - 1].quota = std::max(max_features - sum, 0);
        size_t off = 0;
        int cand_off = 0, kp_off = 0, ftile = 0, btile = 0;
        std::vector<uint32_t> tab;
        for (int l = 0; l < kLevels; ++l) {
            LevelGeom& L = n.L[l];
            L.scale = (float)std::pow((double)1.2f, (double)l);
            L.w = cv_round_f((float)cols / L.scale);
            L.h = cv_round_f((float)rows / L.scale);
            if (L.w < 1 || L.h < 1) return fail(c, VO_ERR_INVALID, "image too small for an 8-level pyramid");
            L.pitch = (int)align_up(L.w, 64);
            L.off = (uint32_t)off;
            off += align_up((size_t)L.pitch * L.h, 256);
            maxq = std::max(maxq, L.quota);
            int iw = L.w - 2 * kEdge, ih = L.h - 2 * kEdge;
            bool has = iw > 0 && ih > 0;
            L.cand_off = cand_off;
            L.cand_cap = has ? ((iw + 1) / 2 + 1) * ((ih + 1) / 2 + 1) : 0;
            cand_off += L.cand_cap;
            L.kp_off = kp_off;
            L.kp_cap = L.quota + 256;
            kp_off += L.kp_cap;
            L.fast_tx = has ? (iw + FT_W - 1) / FT_W : 0;
            L.fast_ty = has ? (ih + FT_H - 1) / FT_H : 0;
            L.fast_tile0 = ftile;
            ftile += L.fast_tx * L.fast_ty;
            L.blur_tx = (L.w + BT_W - 1) / BT_W;
            L.blur_ty = (L.h + BT_H - 1) / BT_H;
            L.blur_tile0 = btile;
            btile += L.blur_tx * L.blur_ty;
            if (l > 0) {
                L.xtab = (uint32_t)tab.size();
                tab.resize(tab.size() + L.w);
                lin_table(n.L[l - 1].w, L.w, &tab[L.xtab]);
                L.ytab = (uint32_t)tab.size();
                tab.resize(tab.size() + L.h);
                lin_table(n.L[l - 1].h, L.h, &tab[L.ytab]);
            }
        }
        n.blk_stride = off;
        n.cand_stride = cand_off;
        n.kp_stride = kp_off;
        n.fast_tiles = ftile;
        n.blur_tiles = btile;
        int need = 2 * maxq + 512;
        n.sel_cap = need <= 2048 ? 2048 : need <= 4096 ? 4096 : 8192;
        if (need > 8192) return fail(c, VO_ERR_INVALID, "max_features too large for the selection kernel");
        c->h_tab = tab;
        if (int rc = ensure(c, c->tab, std::max<size_t>(tab.size(), 1) * 4)) return rc;
        VO_HIP(c, hipMemcpyAsync(c->tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
        g = n;
        c->geom_valid = true;
    }
    g.n_images = n_images;
    if (!g_pattern_loaded[c->device]) {
        VO_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(d_pattern), h_pattern, sizeof(h_pattern)));
        g_pattern_loaded[c->device] = true;
    }
    size_t N = (size_t)n_images;
    int rc;
    if ((rc = ensure(c, c->pyr, N * g.blk_stride))) return rc;
    if ((rc = ensure(c, c->blur, N * g.blk_stride))) return rc;
    if ((rc = ensure(c, c->cand, N * std::max(g.cand_stride, 1) * 4))) return rc;
    if ((rc = ensure(c, c->cand_cnt, N * kLevels * 4))) return rc;
    if ((rc = ensure(c, c->kp_x, N * g.kp_stride * 2))) return rc;
    if ((rc = ensure(c, c->kp_y, N * g.kp_stride * 2))) return rc;
    if ((rc = ensure(c, c->kp_resp, N * g.kp_stride * 4))) return rc;
    if ((rc = ensure(c, c->kp_angle, N * g.kp_stride * 4))) return rc;
    if ((rc = ensure(c, c->kp_cnt, N * kLevels * 4))) return rc;
    if ((rc = ensure(c, c->flags, 256))) return rc;
    return VO_OK;
}

int orb_run(vo_ctx* c, const uint8_t* d_imgs, vo_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts) {
    const Geom& g = c->g;
    static const SmallTabs tabs = make_tabs();
    hipStream_t s = c->stream;
    int N = g.n_images;
    c->last_imgs = d_imgs;
    const uint8_t* pyr = (const uint8_t*)c->pyr.p;
    VO_HIP(c, hipMemsetAsync(c->cand_cnt.p, 0, (size_t)N * kLevels * 4, s));
    VO_HIP(c, hipMemsetAsync(c->flags.p, 0, 256, s));
    t_begin(c, VO_T_RESIZE);
    for (int l = 1; l < kLevels; ++l) {
        dim3 grid((g.L[l].w + 255) / 256, (g.L[l].h + 3) / 4, N);
        hipLaunchKernelGGL(resize_kernel, grid, dim3(256), 0, s, g, l, d_imgs, (uint8_t*)c->pyr.p,
                           (const uint32_t*)c->tab.p);
    }
    t_end(c, VO_T_RESIZE);
    if (g.fast_tiles > 0) {
        t_begin(c, VO_T_FAST);
        hipLaunchKernelGGL(fast_kernel, dim3(g.fast_tiles, N), dim3(256), 0, s, g, d_imgs, pyr, (uint32_t*)c->cand.p,
                           (uint32_t*)c->cand_cnt.p);
        t_end(c, VO_T_FAST);
    }
    t_begin(c, VO_T_SELECT);
#define VO_SELECT(CAPV)                                                                                              \
    hipLaunchKernelGGL(select_kernel<CAPV>, dim3(kLevels, N), dim3(512), 0, s, g, tabs, d_imgs, pyr,                \
                       (const uint32_t*)c->cand.p, (const uint32_t*)c->cand_cnt.p, (uint16_t*)c->kp_x.p,            \
                       (uint16_t*)c->kp_y.p, (float*)c->kp_resp.p, (float*)c->kp_angle.p, (uint32_t*)c->kp_cnt.p,   \
                       (uint32_t*)c->flags.p)
    if (g.sel_cap == 2048) VO_SELECT(2048);
    else if (g.sel_cap == 4096) VO_SELECT(4096);
    else VO_SELECT(8192);
#undef VO_SELECT
    t_end(c, VO_T_SELECT);
    if (d_desc) {
        t_begin(c, VO_T_BLUR);
        hipLaunchKernelGGL(blur_kernel, dim3(g.blur_tiles, N), dim3(256), 0, s, g, tabs, d_imgs, pyr,
                           (uint8_t*)c->blur.p);
        t_end(c, VO_T_BLUR);
    }
    t_begin(c, VO_T_DESCRIBE);
    int max_kp = 0;
    for (int l = 0; l < kLevels; ++l) max_kp = std::max(max_kp, g.L[l].kp_cap);
    hipLaunchKernelGGL(describe_kernel, dim3((max_kp + 3) / 4, kLevels, N), dim3(256), 0, s, g,
                       (const uint8_t*)c->blur.p, (const uint16_t*)c->kp_x.p, (const uint16_t*)c->kp_y.p,
                       (const float*)c->kp_resp.p, (const float*)c->kp_angle.p, (const uint32_t*)c->kp_cnt.p, d_kps,
                       d_desc, cap, d_counts, (uint32_t*)c->flags.p);
    t_end(c, VO_T_DESCRIBE);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

}  // namespace vo

using namespace vo;

extern "C" int vo_orb_detect_batch_dev(vo_ctx* c, const uint8_t* d_imgs, int n_images, int rows, int cols, int step,
                                       size_t image_stride, int max_features, vo_keypoint* d_kps, uint8_t* d_desc,
                                       int cap, int32_t* d_counts) {
    if (!c) return VO_ERR_INVALID;
    if (n_images <= 0 || rows <= 0 || cols <= 0 || step < cols || max_features < 0 || cap <= 0 || !d_imgs || !d_kps ||
        !d_counts)
        return fail(c, VO_ERR_INVALID, "vo_orb_detect_batch_dev: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    if (int rc = orb_prepare(c, n_images, rows, cols, step, image_stride, max_features)) return rc;
    return orb_run(c, d_imgs, d_kps, d_desc, cap, d_counts);
}

extern "C" int vo_orb_detect(vo_ctx* c, const uint8_t* img, int rows, int cols, int step, int max_features,
                             vo_keypoint* kps_out, uint8_t* desc_out, int cap, int* n_out) {
    if (!c) return VO_ERR_INVALID;
    if (!n_out) return fail(c, VO_ERR_INVALID, "vo_orb_detect: n_out is NULL");
    *n_out = 0;
    if (!img || rows <= 0 || cols <= 0) return VO_OK;   // empty image -> no keypoints
    if (step < cols || cap <= 0 || !kps_out || max_features < 0)
        return fail(c, VO_ERR_INVALID, "vo_orb_detect: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    size_t bytes = (size_t)rows * step;
    int rc;
    if ((rc = ensure(c, c->st_img, bytes))) return rc;
    if ((rc = ensure(c, c->st_kps, (size_t)cap * sizeof(vo_keypoint)))) return rc;
    if ((rc = ensure(c, c->st_desc, (size_t)cap * 32))) return rc;
    if ((rc = ensure(c, c->st_cnt, 64))) return rc;
    VO_HIP(c, hipMemcpyAsync(c->st_img.p, img, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = orb_prepare(c, 1, rows, cols, step, bytes, max_features))) return rc;
    if ((rc = orb_run(c, (const uint8_t*)c->st_img.p, (vo_keypoint*)c->st_kps.p,
                      desc_out ? (uint8_t*)c->st_desc.p : nullptr, cap, (int32_t*)c->st_cnt.p)))
        return rc;
    int32_t n = 0;
    uint32_t flags = 0;
    VO_HIP(c, hipMemcpyAsync(&n, c->st_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(&flags, c->flags.p, 4, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    if (flags) return fail(c, VO_ERR_CAPACITY, "vo_orb_detect: list overflow (flags " + std::to_string(flags) + ")");
    if (n > 0) {
        VO_HIP(c, hipMemcpyAsync(kps_out, c->st_kps.p, (size_t)n * sizeof(vo_keypoint), hipMemcpyDeviceToHost,
                                 c->stream));
        if (desc_out)
            VO_HIP(c, hipMemcpyAsync(desc_out, c->st_desc.p, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    return VO_OK;
}

extern "C" int vo_debug_level(vo_ctx* c, int image, int level, int blurred, uint8_t* out, int* w, int* h) {
    if (!c || !c->geom_valid || level < 0 || level >= kLevels || image < 0 || image >= c->g.n_images || !out)
        return c ? fail(c, VO_ERR_INVALID, "vo_debug_level: bad argument") : VO_ERR_INVALID;
    const Geom& g = c->g;
    const LevelGeom& L = g.L[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    const uint8_t* src;
    size_t pitch;
    if (blurred) {
        src = (const uint8_t*)c->blur.p + (size_t)image * g.blk_stride + L.off;
        pitch = L.pitch;
    } else if (level == 0) {
        src = c->last_imgs + (size_t)image * g.in_stride;
        pitch = g.step;
    } else {
        src = (const uint8_t*)c->pyr.p + (size_t)image * g.blk_stride + L.off;
        pitch = L.pitch;
    }
    VO_HIP(c, hipStreamSynchronize(c->stream));
    VO_HIP(c, hipMemcpy2D(out, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return VO_OK;
}

extern "C" int vo_debug_candidates(vo_ctx* c, int image, int level, int32_t* out, int cap, int* n_out) {
    if (!c || !c->geom_valid || level < 0 || level >= kLevels || image < 0 || image >= c->g.n_images || !out || !n_out)
        return c ? fail(c, VO_ERR_INVALID, "vo_debug_candidates: bad argument") : VO_ERR_INVALID;
    const Geom& g = c->g;
    const LevelGeom& L = g.L[level];
    VO_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t n = 0;
    VO_HIP(c, hipMemcpy(&n, (const uint32_t*)c->cand_cnt.p + image * kLevels + level, 4, hipMemcpyDeviceToHost));
    n = std::min<uint32_t>(n, L.cand_cap);
    std::vector<uint32_t> tmp(n);
    if (n)
        VO_HIP(c, hipMemcpy(tmp.data(), (const uint32_t*)c->cand.p + (size_t)image * g.cand_stride + L.cand_off,
                            (size_t)n * 4, hipMemcpyDeviceToHost));
    int m = std::min<int>((int)n, cap);
    for (int i = 0; i < m; ++i) {
        out[i * 3] = tmp[i] & 0xFFF;
        out[i * 3 + 1] = (tmp[i] >> 12) & 0xFFF;
        out[i * 3 + 2] = tmp[i] >> 24;
    }
    *n_out = (int)n;
    return VO_OK;
}
