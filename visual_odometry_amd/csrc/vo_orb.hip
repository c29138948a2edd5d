// ORB on gfx950: pyramid (fixed-point bilinear), FAST-9/16 + NMS, score/Harris
// selection, intensity-centroid angle, 7x7 fixed-point blur, rBRIEF.
// Stands in for cv::ORB::detectAndCompute as called by
// ref:src/feature_detector.cpp:12-27.  All integer stages are exact; float
// stages keep OpenCV's operation order (built with -ffp-contract=off).
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "vo_internal.h"

namespace vo {

static const int8_t h_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__device__ __attribute__((aligned(16))) int8_t d_pattern[256 * 4];
// Disc masks of the intensity-centroid pass: entry [t][hl] is the byte mask of the dword
// that half-wave lane hl (row hl >> 3 of a four-row group, dword hl & 7 of the 32-byte row
// window starting at x - 15) loads in step t (patch row 4t - 15 + (hl >> 3)).
__device__ uint32_t d_angle_mask[8 * 32];
static bool g_pattern_loaded[64] = {false};   // per GPU; guarded by g_pattern_mutex (contexts live on several host threads)
static std::mutex g_pattern_mutex;

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

// Workgroups are handed to the eight XCDs round-robin by flat workgroup id, and every XCD
// has its own L2.  This permutation gives XCD x the contiguous range
// [x * ceil(total/8), (x+1) * ceil(total/8)) of logical work items, so neighbouring tiles and
// the keypoints of one image share an L2 (halo rows and partially used 128-byte lines are
// then fetched from memory once instead of once per XCD).  The launch uses
// xcd_grid(total) workgroups; returns false for the padding ones.
__device__ __forceinline__ bool xcd_remap(uint32_t total, uint32_t* logical) {
    const uint32_t per = (total + 7u) >> 3;
    *logical = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    return (blockIdx.x >> 3) < per && *logical < total;
}
static inline uint32_t xcd_grid(uint32_t total) { return ((total + 7u) >> 3) << 3; }

// work = img * tiles + tile with magic = floor(2^32 / tiles): the high product is the quotient
// or one below it.  All operands are wave-uniform, so this stays on the scalar unit.
__device__ __forceinline__ void split_work(uint32_t work, uint32_t tiles, uint32_t magic, int* img, int* tile) {
    uint32_t q = __umulhi(work, magic), r = work - q * tiles;
    if (r >= tiles) {
        ++q;
        r -= tiles;
    }
    *img = (int)q;
    *tile = (int)r;
}

// Four bytes at an arbitrary byte address (one global_load_dword; gfx950 global loads need
// no alignment).
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 va, vb;
    __builtin_memcpy(&va, &a, 4);
    __builtin_memcpy(&vb, &b, 4);
    return __builtin_amdgcn_udot2(va, vb, c, false);
}

// ------------------------------------------------------------------ resize
// Level l from level l-1, INTER_LINEAR_EXACT semantics: 8.8 coefficients,
// out = (m0*(c0*a + c1*b) + m1*(c0*c + c1*d) + 2^15) >> 16  (never above 255, the weights
// sum to 2^16).  Row/column table entry = (source offset << 16) | c1, c0 = 256 - c1.
// A lane produces 4 columns x RS_ROWS rows.  Its four columns read at most eight
// consecutive source bytes per row (two unaligned dwords); the group table holds, per
// column, a v_perm selector that places the two neighbours in the halves of a dword and
// the coefficient pair (c0 | c1 << 16), so a horizontal interpolation is one v_perm and
// one v_dot2_u32_u16 and the vertical one two 24-bit multiply-adds.
constexpr int RS_ROWS = 4;
__global__ __launch_bounds__(256) void resize_kernel(Geom g, int l, const uint8_t* __restrict__ in,
                                                     uint8_t* __restrict__ pyr, const uint32_t* __restrict__ tab) {
    const LevelGeom& D = g.L[l];
    const LevelGeom& S = g.L[l - 1];
    const int img = blockIdx.y;
    const uint32_t item = blockIdx.x * 256 + threadIdx.x;
    const uint32_t rg = __umulhi(item, D.ncg_magic), cg = item - rg * D.ncg;
    const int ybase = (int)rg * RS_ROWS, x4 = (int)cg * 4;
    if (ybase >= D.h) return;
    int spitch;
    const uint8_t* src = level_base(g, in, pyr, img, l - 1, &spitch);
    uint8_t* dst = pyr + (size_t)img * g.blk_stride + D.off;
    const uint32_t* gt = tab + D.gtab + 12 * cg;
    const uint32_t head = gt[0];
    const int sx0 = (int)(head & 0xFFFFu);
    uint32_t ty[RS_ROWS];
#pragma unroll
    for (int j = 0; j < RS_ROWS; ++j) ty[j] = tab[D.ytab + min(ybase + j, D.h - 1)];
    if (!(head >> 31)) {
        const uint4 sel = *reinterpret_cast<const uint4*>(gt + 4), cf = *reinterpret_cast<const uint4*>(gt + 8);
        uint32_t a[RS_ROWS][2], b[RS_ROWS][2];
        uint32_t prev_sy1 = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < RS_ROWS; ++j) {
            // 32-bit offsets from the wave-uniform image base (24-bit multiplies: rows and
            // pitches are below 2^13)
            uint32_t sy = ty[j] >> 16, sy1 = min(sy + 1u, (uint32_t)(S.h - 1));
            uint32_t o0 = __umul24(sy, (uint32_t)spitch) + (uint32_t)sx0;
            uint32_t o1 = __umul24(sy1, (uint32_t)spitch) + (uint32_t)sx0;
            // With a scale of 1.2 four of five output rows start at the source row the previous one
            // ended on: that row is in registers already.  (The kernel is bound by the texture
            // addresser: every load instruction it does not issue counts.)
            if (sy == prev_sy1) {
                a[j][0] = b[j > 0 ? j - 1 : 0][0];
                a[j][1] = b[j > 0 ? j - 1 : 0][1];
            } else {
                a[j][0] = load_u32_unaligned(src + o0);
                a[j][1] = load_u32_unaligned(src + (o0 + 4u));
            }
            b[j][0] = load_u32_unaligned(src + o1);
            b[j][1] = load_u32_unaligned(src + (o1 + 4u));
            prev_sy1 = sy1;
        }
        const uint32_t sl[4] = {sel.x, sel.y, sel.z, sel.w}, cc[4] = {cf.x, cf.y, cf.z, cf.w};
#pragma unroll
        for (int j = 0; j < RS_ROWS; ++j) {
            const uint32_t m1 = ty[j] & 0xFFFFu, m0 = 256u - m1;
            uint32_t sum[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t h0 = udot2_u16(__builtin_amdgcn_perm(a[j][1], a[j][0], sl[k]), cc[k], 0u);
                uint32_t h1 = udot2_u16(__builtin_amdgcn_perm(b[j][1], b[j][0], sl[k]), cc[k], 0u);
                sum[k] = __umul24(m1, h1) + (__umul24(m0, h0) + 32768u);
            }
            uint32_t outw = __builtin_amdgcn_perm(sum[1], sum[0], 0x0C0C0602u) | __builtin_amdgcn_perm(sum[3], sum[2], 0x06020C0Cu);
            // pitch is a multiple of 64: the dword store stays inside the row
            if (ybase + j < D.h)
                *reinterpret_cast<uint32_t*>(dst + (__umul24((uint32_t)(ybase + j), (uint32_t)D.pitch) + (uint32_t)x4)) = outw;
        }
    } else {
        // groups at the right edge (or with an unusual column spacing): byte loads
        uint32_t tx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) tx[k] = tab[D.xtab + min(x4 + k, D.w - 1)];
#pragma unroll
        for (int j = 0; j < RS_ROWS; ++j) {
            if (ybase + j >= D.h) break;
            int sy = ty[j] >> 16, sy1 = min(sy + 1, S.h - 1);
            const uint32_t m1 = ty[j] & 0xFFFFu, m0 = 256u - m1;
            const uint8_t* r0 = src + (size_t)sy * spitch;
            const uint8_t* r1 = src + (size_t)sy1 * spitch;
            uint32_t outw = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t v = 0;
                if (x4 + k < D.w) {
                    int sx = tx[k] >> 16;
                    uint32_t c1 = tx[k] & 0xFFFF, c0 = 256 - c1;
                    int sx1 = min(sx + 1, S.w - 1);
                    uint32_t h0 = c0 * r0[sx] + c1 * r0[sx1];
                    uint32_t h1 = c0 * r1[sx] + c1 * r1[sx1];
                    v = (m0 * h0 + m1 * h1 + 32768u) >> 16;
                }
                outw |= v << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(dst + (size_t)(ybase + j) * D.pitch + x4) = outw;
        }
    }
}

// ------------------------------------------------------------------ FAST
constexpr int FT_W = 64, FT_H = 32;
constexpr int FP_W = 80, FP_H = FT_H + 8;   // pixel tile (dword-aligned origin, 4-px halo)
constexpr int FS_W = FP_W, FS_H = FT_H + 2;   // score tile (1-px halo), logical width FT_W + 2; the pixel tile's pitch,
                                              // so that a queue entry (byte offset into the pixel tile) also addresses the score
constexpr int FQ_N = FS_H * (FT_W + 2);     // positions whose score is needed

constexpr int FP_N = FP_H * FP_W;            // bytes of the pixel tile
// Offset of the tile's inverted copy (255 - p): 16 LDS banks away from the plain pixel, so that a
// queue group's darker-ring entries (which read the copy) and the brighter-ring entries of the
// pixels next to them do not meet on one bank.
constexpr int FI_OFF = FP_N + 64;
// Capacity of the first queue.  A position can enter it once per polarity, 2 * FQ_N entries for the
// whole tile, but hardly any tile needs half of that, and the full size costs the eighth
// workgroup per compute unit (22.8 against 18.4 KB of LDS; the kernel is bound by the tiles it has
// in flight).  It holds what half a tile can produce at most; a tile whose pre-test passes more
// (pixel, polarity) pairs than fit is worked through in two halves of its rows.
constexpr int FQ1_N = 2 * ((FS_H + 1) / 2) * (FT_W + 2) + 60;
enum { FAST_PLAIN = 0, FAST_RAISED = 1, FAST_REDO = 2 };   // fast_kernel's mode

// Largest minimum over the sixteen 9-arcs of the ring (Bresenham circle of radius 3) around
// p, minus the centre: > thr iff the pixel is a FAST-9/16 corner whose ring is brighter, and
// then cornerScore<16> is that value minus 1.  Read from the inverted copy of the tile the
// same code answers for a ring that is darker (255 - r against 255 - centre).  An arc's
// minimum is min3 of three 3-runs: 16 + 16 three-operand minima and 8 three-operand maxima.
__device__ __forceinline__ int ring_arc_contrast(const uint8_t* p) {
    constexpr int P = FP_W;
    const int off[16] = {3 * P,      3 * P + 1,  2 * P + 2,  P + 3,  3,       -P + 3,     -2 * P + 2, -3 * P + 1,
                         -3 * P,     -3 * P - 1, -2 * P - 2, -P - 3, -3,      P - 3,      2 * P - 2,  3 * P - 1};
    int r[16], m3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = (int)p[off[k]];
    const int v = (int)p[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) m3[i] = min(min(r[i], r[(i + 1) & 15]), r[(i + 2) & 15]);
    int best = 0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const int a = min(min(m3[i], m3[(i + 3) & 15]), m3[(i + 6) & 15]);
        const int b = min(min(m3[i + 1], m3[(i + 4) & 15]), m3[(i + 7) & 15]);
        best = max(max(best, a), b);
    }
    return best - v;
}

// LDS fetch-and-add issued as written.  (atomicAdd under `if (lane == 0)` is rewritten by the
// compiler's atomic optimiser into a count of the active lanes, a multiply and a broadcast:
// some ten vector instructions for a single-lane add.)
__device__ __forceinline__ uint32_t lds_fetch_add(uint32_t* p, uint32_t v) {
    uint32_t r;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a), "v"(v) : "memory");
    return r;
}

// Inclusive prefix sum over the 64 lanes of a wavefront (all lanes active): four shifts inside
// the rows of 16 lanes, then the totals of rows 0 and 2 into rows 1 and 3 and that of the
// lower half into the upper one.  Six DPP adds; written out because each must wait two
// states for the previous one's result.
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return x;
}

// Appends `val` for the lanes with pass == true to an LDS queue: one atomic per wavefront.
__device__ __forceinline__ void wave_push(bool pass, uint16_t val, uint16_t* q, uint32_t* cnt, int lane) {
    unsigned long long m = __ballot(pass);
    if (m == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = lds_fetch_add(cnt, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (pass) q[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base))] = val;
}

// One launch over every level's 64x32 tiles of the 31-px interior.  Two filters, the first
// feeding the second through a wavefront-compacted LDS queue so that the expensive one runs
// on dense wavefronts:
//   1. four cardinal ring pixels, per polarity: a 9-arc always contains two adjacent cardinals;
//   2. the largest 9-arc minimum of the ring in the polarity the pre-test passed (both in
//      the rare case that both did): it decides the corner test and is the corner score, so
//      the ring is read once; scores go to an LDS score tile, corners to a second queue;
// then 3x3 non-max suppression, the border filter and one global atomic per tile.
//
// MODE (see orb_run and fast_raise_kernel): FAST_PLAIN - threshold kFastThr, every candidate
// reported; FAST_RAISED - the raised threshold of the tile's (image, level); FAST_REDO - a fixed
// grid that walks the unsampled tiles of the (image, level)s on the redo list (none, as a rule),
// threshold kFastThr again, reporting the candidates the raised pass left out.
template <int MODE>
__global__ __launch_bounds__(256) void fast_kernel(Geom g, const uint8_t* __restrict__ in,
                                                   const uint8_t* __restrict__ pyr, const uint32_t* __restrict__ tab,
                                                   uint32_t tile_first, uint32_t ntiles, uint32_t tile_magic,
                                                   uint32_t* __restrict__ cand, uint32_t* __restrict__ cand_cnt,
                                                   unsigned long long* __restrict__ stats,
                                                   const uint32_t* __restrict__ thr_tab,
                                                   const uint32_t* __restrict__ redo_list, uint32_t redo_span) {
    __shared__ __attribute__((aligned(16))) uint8_t px[FI_OFF + FP_N];   // the tile, then 255 - tile at FI_OFF
    __shared__ __attribute__((aligned(16))) uint8_t sc[FS_H * FS_W];
    __shared__ __attribute__((aligned(16))) uint16_t q1[FQ1_N];
    __shared__ uint16_t q2[FQ_N];
    uint32_t* const list = reinterpret_cast<uint32_t*>(q1);   // the first queue is dead when the list is built
    static_assert(sizeof(uint16_t) * FQ1_N >= sizeof(uint32_t) * (FT_W * FT_H / 4), "the candidate list fits the first queue");
    __shared__ uint32_t n1, n2, nlist, gbase;
    __shared__ uint32_t colmask[32];

    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t item = blockIdx.x;   // (FAST_REDO: position in the redo list x tiles of a level)
    for (;;) {
    int img, tile;
    if constexpr (MODE == FAST_REDO) {
        // redo_list[0] = entries, redo_list[1 + f] = image * kLevels + level; entry f owns the items
        // [f * redo_span, (f + 1) * redo_span), of which the first (unsampled tiles of its level) are tiles
        const uint32_t n_items = min(redo_list[0], (uint32_t)(g.n_images * kLevels)) * redo_span;
        if (item >= n_items) return;
        const uint32_t f = item / redo_span, k = item - f * redo_span;
        item += gridDim.x;
        const uint32_t e = redo_list[1 + f];
        const LevelGeom& Lr = g.L[e % kLevels];
        if (k >= (uint32_t)(Lr.fast_tx * Lr.fast_ty - Lr.fast_ns)) continue;
        img = (int)(e / kLevels);
        tile = Lr.fast_u0 + (int)k;
    } else {
        uint32_t work;
        if (!xcd_remap(ntiles * (uint32_t)g.n_images, &work)) return;
        split_work(work, ntiles, tile_magic, &img, &tile);
        tile += (int)tile_first;   // this launch covers tiles [tile_first, tile_first + ntiles) of every image
    }
    const uint32_t te = tab[g.fast_ttab + tile];
    const int l = (int)(te >> 24);
    const LevelGeom& L = g.L[l];
    // corner threshold of this tile and the scores it reports, workgroup-uniform
    uint32_t thr = (uint32_t)kFastThr, report_below = 256u;
    if constexpr (MODE == FAST_RAISED) thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)thr_tab[img * kLevels + l]);
    if constexpr (MODE == FAST_REDO) report_below = thr_tab[img * kLevels + l];
    int x0 = kEdge + (int)(te & 0xFFFu) * FT_W, y0 = kEdge + (int)((te >> 12) & 0xFFFu) * FT_H;
    int pitch;
    const uint8_t* base = level_base(g, in, pyr, img, l, &pitch);
    int ax0 = (x0 - 4) & ~3;   // LDS column 0 <-> image column ax0
    if (tid == 0) {
        n1 = 0;
        n2 = 0;
        nlist = 0;
    }
    for (int i = tid; i < FS_H * FS_W / 4; i += 256) reinterpret_cast<uint32_t*>(sc)[i] = 0;
    // LDS column of score column 0: x0 = kEdge + 64k, so (x0 - 4) & ~3 = x0 - 7 and x0 - 1 - ax0 = 6
    static_assert(kEdge % 4 == 3 && FT_W % 4 == 0, "the score region starts at LDS column 6");
    constexpr int cx0 = 6;
    constexpr int g0 = cx0 >> 2, ng = ((cx0 + FT_W + 1) >> 2) - g0 + 1;   // dword groups that hold score columns
    const int max_sx = L.w - kEdge - (x0 - 1), max_sy = L.h - kEdge - (y0 - 1);   // inclusive bounds
    if (tid < FP_W / 4) {
        // pre-test result bits (7, 15, 23, 31: pixels 0..3) of a dword column's pixels that are score
        // columns 0 .. lim_x: the same for every row, so looked up instead of recomputed per group
        // (zero for the dword columns left and right of the score region)
        const int lim_x = min(FT_W + 1, max_sx);
        const int sx0 = 4 * tid - cx0;
        const int lo = max(-sx0, 0), hi = min(lim_x - sx0 + 1, 4);
        const uint32_t nib = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        colmask[tid] = ((nib & 1u) << 7) | ((nib & 2u) << 14) | ((nib & 4u) << 21) | ((nib & 8u) << 28);
    }

    bool aligned = ((((uintptr_t)base) | (unsigned)pitch) & 3) == 0;
    if (aligned) {
        // 12 rows of 20 dwords per pass (240 of the 256 lanes), row and column of a lane fixed
        int valid_dw = (l == 0 ? (g.cols & ~3) : pitch) >> 2;
        constexpr int RD = FP_W / 4, RPP = 256 / RD;
        const int r0 = tid / RD, c = tid - r0 * RD;
        const uint32_t col = 4u * (uint32_t)min((ax0 >> 2) + c, valid_dw - 1);
        if (r0 < RPP) {
#pragma unroll
            for (int r = r0; r < FP_H; r += RPP) {
                int y = min(max(y0 - 4 + r, 0), L.h - 1);
                uint32_t v = *reinterpret_cast<const uint32_t*>(base + ((uint32_t)__mul24(y, pitch) + col));
                *reinterpret_cast<uint32_t*>(&px[r * FP_W + 4 * c]) = v;
                *reinterpret_cast<uint32_t*>(&px[FI_OFF + r * FP_W + 4 * c]) = ~v;
            }
        }
    } else {
        for (int i = tid; i < FP_H * FP_W; i += 256) {
            int r = i / FP_W, c = i - r * FP_W;
            int y = min(max(y0 - 4 + r, 0), L.h - 1);
            int x = min(ax0 + c, L.w - 1);
            const uint8_t v = base[(size_t)y * pitch + x];
            px[i] = v;
            px[FI_OFF + i] = (uint8_t)~v;
        }
    }
    __syncthreads();

    // Stages 1 and 2 run over score rows [row_lo, row_hi): the whole tile, unless its pre-test
    // passes more entries than the first queue holds - then the rows are halved (half a tile
    // always fits) and the parts follow one another; scores and corners accumulate.
    constexpr int sc_shift = 3 * FP_W + cx0;
    static_assert(FS_W == FP_W, "one offset addresses both tiles");
    int row_lo = 0, row_hi = FS_H;
    uint32_t n1_total = 0;   // pre-test passes of all parts (scene statistics)
    static_assert(FQ1_N >= 2 * ((FS_H + 1) / 2) * (FT_W + 2), "half a tile's entries fit the first queue");
    for (;;) {
    // stage 1: cardinal pre-test on the (FT_W+2) x (FT_H+2) region whose scores are needed,
    // four horizontally adjacent pixels per lane, one per byte.  v_lerp_u8 averages bytes
    // without carries between them: x = (a + (255 - b)) >> 1 (255 - b read from the inverted
    // copy) and then (x + 128 - thr / 2) >> 1 has bit 7 set iff a - b > thr.  A 9-arc contains two
    // adjacent cardinals, i.e. (S | N) & (E | W) for one polarity.  A queue entry is the byte
    // offset of the pixel in the tile copy its polarity is scored on: the plain one for a
    // brighter ring, the inverted one (+ FI_OFF) for a darker ring.
    {
        static_assert(kFastThr % 2 == 0 && kFastThr < 254, "the two-step average tests a - b >= thr + 1 exactly for even thr");
        // (a raised threshold is even and at most kFastThrMax by construction: fast_raise_kernel)
        constexpr int PD = FP_W / 4;   // dwords per LDS row
        const uint32_t* px32 = reinterpret_cast<const uint32_t*>(px);
        const uint32_t* ix32 = px32 + FI_OFF / 4;
        const int lim_y = min(FS_H - 1, max_sy);
        // One first average x = (C + 255 - b) >> 1 per cardinal b serves both polarities.  Darker
        // ring, C - b > thr: bit 7 of (x + 128 - thr / 2) >> 1, exact.  Brighter ring, b - C > thr,
        // i.e. C + 255 - b <= 254 - thr: x <= 127 - thr / 2, i.e. bit 7 of (x + 128 + thr / 2) >> 1
        // clear - which also lets b - C == thr through (odd sums round down); the ring stage
        // decides exactly, so the pre-test may pass a few pixels too many.
        const uint32_t KT = (128u - (thr >> 1)) * 0x01010101u;
        const uint32_t KB = (128u + (thr >> 1)) * 0x01010101u;
        auto avg = [](uint32_t a, uint32_t b) -> uint32_t { return __builtin_amdgcn_lerp(a, b, 0u); };
        // A lane takes the dword at tile offset 3 * PD + i: consecutive lanes read consecutive
        // dwords whatever the row, so no read of this stage meets a bank conflict (the three
        // dword columns per row outside the score region ride along with an empty column mask).
        static_assert(g0 >= 1 && g0 + ng < PD, "the left and right neighbour dwords stay inside the row");
        const int row_end = min(row_hi - 1, lim_y);   // last score row of this part
        for (int i0 = PD * row_lo; i0 < PD * row_hi; i0 += 256) {
            const int i = i0 + tid;
            const int sy = i / PD, gi = i - sy * PD;
            uint32_t TD = 0, TB = 0;   // bits 7, 15, 23, 31: pixels 0..3 of the group pass (darker / brighter ring)
            if (sy <= row_end) {
                const int idx = 3 * PD + i;
                const uint32_t C = px32[idx];
                const uint32_t Ci = ix32[idx], Li = ix32[idx - 1], Ri = ix32[idx + 1];
                const uint32_t Si = ix32[idx + 3 * PD], Ni = ix32[idx - 3 * PD];
                const uint32_t Wi = __builtin_amdgcn_alignbyte(Ci, Li, 1), Ei = __builtin_amdgcn_alignbyte(Ri, Ci, 3);
                const uint32_t xS = avg(C, Si), xN = avg(C, Ni), xE = avg(C, Ei), xW = avg(C, Wi);
                const uint32_t cm = colmask[gi];
                TD = (avg(xS, KT) | avg(xN, KT)) & (avg(xE, KT) | avg(xW, KT)) & cm;
                TB = ~((avg(xS, KB) & avg(xN, KB)) | (avg(xE, KB) & avg(xW, KB))) & cm;
            }
            const uint32_t T = TD | TB;
            // Entries go to the queue in pixel order (a lane's up to four entries side by side, its
            // first slot from a prefix sum over the lanes' entry counts): a group of 32 consecutive
            // entries then spans little more than one tile row, so the byte reads of the ring
            // stage fall into distinct LDS banks or into the same dword.  (With one ballot per
            // byte position a group gathered every fourth pixel of four or five rows, whose
            // 20-dword pitch folds them onto the same banks.)
            const uint32_t pc = (uint32_t)__popc(T & 0x80808080u);
            const uint32_t upto = wave_prefix_sum(pc);   // entries of this lane and the lanes below
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
            if (tot == 0) continue;   // wave-uniform
            uint32_t qb = 0;
            if (lane == 0) qb = lds_fetch_add(&n1, tot);
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            uint32_t sl = (qb + upto - pc) << 1;   // byte offset of this lane's first entry
            const uint32_t pos = (uint32_t)((sy + 3) * FP_W + 4 * gi);
            auto put = [&](uint32_t entry) {   // (entries beyond the queue are dropped: the part is then run again in halves)
                if (sl < 2u * (uint32_t)FQ1_N) *reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(q1) + sl) = (uint16_t)entry;
                sl += 2u;
            };
            // one entry per passing pixel: the darker-ring copy if that polarity passed, else the plain one
            if (T & 0x80u) put(__umul24(__builtin_amdgcn_ubfe(TD, 7, 1), FI_OFF) + pos);
            if (T & 0x8000u) put(__umul24(__builtin_amdgcn_ubfe(TD, 15, 1), FI_OFF) + (pos + 1));
            if (T & 0x800000u) put(__umul24(__builtin_amdgcn_ubfe(TD, 23, 1), FI_OFF) + (pos + 2));
            if (T & 0x80000000u) put(__umul24(TD >> 31, FI_OFF) + (pos + 3));
            // pixels that passed in both polarities (saddle-like neighbourhoods, about one in a
            // hundred entries): a second entry for the brighter-ring polarity
            const uint32_t both = TD & TB;
            if (__ballot(both != 0)) {   // (some lane of three wavefront passes in four on the dense scene)
                // the one or two lanes concerned append for themselves: a per-lane LDS add, where
                // four more ballot-and-compact rounds cost five times the instructions
                if (both) {
                    uint32_t s = lds_fetch_add(&n1, (uint32_t)__popc(both));
                    auto put2 = [&](uint32_t entry) {
                        if (s < (uint32_t)FQ1_N) q1[s] = (uint16_t)entry;
                        ++s;
                    };
                    if (both & 0x80u) put2(pos);
                    if (both & 0x8000u) put2(pos + 1);
                    if (both & 0x800000u) put2(pos + 2);
                    if (both & 0x80000000u) put2(pos + 3);
                }
            }
        }
    }
    __syncthreads();

    // stage 2: the ring's largest 9-arc minimum against the centre, in the entry's polarity:
    // corner test and corner score at once (score (sy, sx) lives at the pixel's offset in the
    // plain tile minus sc_shift; at most one polarity of a pixel is a corner)
    const uint32_t c1 = n1;
    if (c1 > (uint32_t)FQ1_N) {   // (workgroup-uniform) the part did not fit: its first half, then the rest
        __syncthreads();
        if (tid == 0) n1 = 0;
        row_hi = row_lo + (row_hi - row_lo + 1) / 2;
        __syncthreads();
        continue;
    }
    for (uint32_t i0 = 0; i0 < c1; i0 += 256) {
        const uint32_t i = i0 + tid;
        bool pass = false;
        uint32_t pos = 0;
        if (i < c1) {
            const uint32_t e = q1[i];
            const int best = ring_arc_contrast(&px[e]);
            pos = e >= (uint32_t)FP_N ? e - (uint32_t)FI_OFF : e;
            pass = best > (int)thr;
            if (pass) sc[pos - sc_shift] = (uint8_t)(best - 1);
        }
        wave_push(pass, (uint16_t)pos, q2, &n2, lane);
    }
    __syncthreads();
    n1_total += c1;
    if (row_hi >= FS_H) break;   // (workgroup-uniform)
    row_lo = row_hi;
    row_hi = FS_H;
    if (tid == 0) n1 = 0;
    __syncthreads();
    }
    const uint32_t c2 = n2;

    // 3x3 non-max suppression + border filter on the interior (only corners need the look-up)
    const uint32_t nms_w = (uint32_t)max(min(FT_W, L.w - kEdge - x0), 0), nms_h = (uint32_t)max(min(FT_H, L.h - kEdge - y0), 0);
    for (uint32_t i = tid; i < c2; i += 256) {
        const uint16_t pos = q2[i];
        const int sy = pos / FP_W - 3, sx = pos - (sy + 3) * FP_W - cx0;
        const int ix = sx - 1, iy = sy - 1;
        const int X = x0 + ix, Y = y0 + iy;
        // inside the tile and inside the level's 31-px border: one unsigned compare per axis
        if ((uint32_t)ix >= nms_w || (uint32_t)iy >= nms_h) continue;
        const uint8_t* s = &sc[pos - sc_shift];
        // all nine reads issued together (a short-circuit chain is nine LDS round trips in a row)
        const int v = s[0], n0 = s[-1], n1 = s[1], n2 = s[-FS_W - 1], n3 = s[-FS_W], n4 = s[-FS_W + 1], n5 = s[FS_W - 1],
                  n6 = s[FS_W], n7 = s[FS_W + 1];
        const bool keep = v > max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
        if (keep && (uint32_t)v < report_below) {
            uint32_t slot = atomicAdd(&nlist, 1u);
            list[slot] = ((uint32_t)v << 24) | ((uint32_t)Y << 12) | (uint32_t)X;
        }
    }
    __syncthreads();
    uint32_t n = nlist;
    if (stats && tid == 0) {   // scene statistics (vo_debug_fast_stats), off in normal runs
        const int lim_x = min(FT_W + 1, max_sx), lim_y = min(FS_H - 1, max_sy);
        atomicAdd(&stats[0], (unsigned long long)((lim_x + 1) * (lim_y + 1)));
        atomicAdd(&stats[1], (unsigned long long)n1_total);
        atomicAdd(&stats[2], (unsigned long long)c2);
        atomicAdd(&stats[3], (unsigned long long)n);
    }
    if (n != 0) {   // (workgroup-uniform)
        if (tid == 0) gbase = atomicAdd(&cand_cnt[img * kLevels + l], n);
        __syncthreads();
        uint32_t* out = cand + (size_t)img * g.cand_stride + L.cand_off;
        for (uint32_t i = tid; i < n; i += 256) {
            uint32_t pos = gbase + i;
            if (pos < (uint32_t)L.cand_cap) out[pos] = list[i];
        }
    }
    if constexpr (MODE != FAST_REDO) return;
    __syncthreads();   // the next tile reuses the queues and counters
    }
}

// Raised FAST threshold of an (image, level) from the candidates of its sampled tiles.
//
// Only the candidates among the 2 * quota best scores of a level (and their ties) survive the
// first cut of the selection, and a candidate's score - the largest threshold at which it is
// still a corner - and its fate in the non-maximum suppression do not depend on corners that
// score less than it does.  FAST with a threshold t' >= kFastThr therefore reports exactly the
// candidates of the plain run that score >= t', and as soon as there are 2 * quota of those the
// selection sees the same survivors.  On a densely textured level most of the plain run's work -
// ring tests and scores of tens of thousands of corners - goes into candidates far below the
// cut, so: the sampled tiles (every eighth, spread over the level) run first with the plain
// threshold; this kernel takes the largest even t' at which their candidates, scaled to the
// whole level, still promise 1.5 times the 2 * quota needed; the other tiles run with t';
// fast_check_kernel counts what came of it, and where the promise did not hold the tiles run
// again with the plain threshold and add what was left out, so the list is complete either way.
// A workgroup per (image, level).
constexpr int kFastThrMax = 240;
constexpr int kRaiseMarginNum = 3, kRaiseMarginDen = 2;
__global__ __launch_bounds__(256) void fast_raise_kernel(Geom g, int l_first, const uint32_t* __restrict__ cand,
                                                         const uint32_t* __restrict__ cand_cnt,
                                                         uint32_t* __restrict__ thr_tab) {
    __shared__ uint32_t ge[256];
    __shared__ uint32_t best;
    const int img = blockIdx.x, l = l_first + blockIdx.y, tid = threadIdx.x;
    const LevelGeom& L = g.L[l];
    const int nt = L.fast_tx * L.fast_ty, ns = L.fast_ns;
    if (ns >= nt || L.quota == 0) {   // (no tile left to run with a raised threshold)
        if (tid == 0) thr_tab[img * kLevels + l] = (uint32_t)kFastThr;
        return;
    }
    ge[tid] = 0;
    if (tid == 0) best = (uint32_t)kFastThr;
    __syncthreads();
    const uint32_t total = min(cand_cnt[img * kLevels + l], (uint32_t)L.cand_cap);
    const uint32_t* cl = cand + (size_t)img * g.cand_stride + L.cand_off;
    for (uint32_t i = tid; i < total; i += 256) atomicAdd(&ge[cl[i] >> 24], 1u);
    __syncthreads();
    // suffix sums: ge[s] = candidates of the sampled tiles with a score >= s
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = tid + o < 256 ? ge[tid + o] : 0u;
        __syncthreads();
        ge[tid] += v;
        __syncthreads();
    }
    // ge[t] * nt / ns >= kRaiseMargin (as a ratio of two integers) * 2 * quota
    const unsigned long long have = (unsigned long long)kRaiseMarginDen * ge[tid] * (unsigned long long)nt;
    const unsigned long long want = (unsigned long long)kRaiseMarginNum * (unsigned long long)(2 * L.quota) * (unsigned long long)ns;
    if ((tid & 1) == 0 && tid > kFastThr && tid <= kFastThrMax && have >= want) atomicMax(&best, (uint32_t)tid);
    __syncthreads();
    if (tid == 0) thr_tab[img * kLevels + l] = best;
}

// After the raised pass: does the (image, level) hold the 2 * quota candidates scoring at least
// its raised threshold that make its list complete for the selection?  If not, it joins the
// redo list (redo_list[0] = entries, then image * kLevels + level each) that the redo pass walks.
__global__ __launch_bounds__(256) void fast_check_kernel(Geom g, int l_first, const uint32_t* __restrict__ cand,
                                                         const uint32_t* __restrict__ cand_cnt,
                                                         const uint32_t* __restrict__ thr_tab,
                                                         uint32_t* __restrict__ redo_list) {
    __shared__ uint32_t n_ge;
    const int img = blockIdx.x, l = l_first + blockIdx.y, tid = threadIdx.x;
    const LevelGeom& L = g.L[l];
    const uint32_t raised = thr_tab[img * kLevels + l];
    if (raised <= (uint32_t)kFastThr) return;   // (the other tiles ran with the plain threshold)
    if (tid == 0) n_ge = 0;
    __syncthreads();
    const uint32_t total = min(cand_cnt[img * kLevels + l], (uint32_t)L.cand_cap);
    const uint32_t* cl = cand + (size_t)img * g.cand_stride + L.cand_off;
    uint32_t mine = 0;
    for (uint32_t i = tid; i < total; i += 256) mine += (cl[i] >> 24) >= raised ? 1u : 0u;
    if (mine) atomicAdd(&n_ge, mine);
    __syncthreads();
    if (tid == 0 && n_ge < (uint32_t)(2 * L.quota)) redo_list[1u + atomicAdd(&redo_list[0], 1u)] = (uint32_t)(img * kLevels + l);
}

// ------------------------------------------------------------------ select
__device__ __forceinline__ uint32_t f2sortable(float f) {
    if (f == 0.f) f = 0.f;
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// HarrisResponses: 7x7 block of Sobel-like gradients around (x0, y0), int32 sums, float32
// response in OpenCV's operation order.  The 9x9 support is fetched as 9 rows of three dwords
// from the 4-byte boundary at or below the window (a byte-misaligned 12-byte load costs the
// texture addresser about twice an aligned one), shifted into place, and walked with a
// three-row register window.
__device__ __forceinline__ float harris_at(const uint8_t* img, int pitch, int x0, int y0) {
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    uint32_t w[9][3];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        // one 12-byte load per row: the lanes' windows are scattered, so every load instruction
        // costs a cache-line look-up per lane whatever its width
        const uintptr_t addr = reinterpret_cast<uintptr_t>(img + (size_t)(y0 - 4 + r) * pitch + (x0 - 4));
        const uint32_t skew = (uint32_t)(addr & 3u);   // the nine bytes wanted start here; 9 + 3 <= 12
        uint32_t t[3];
        __builtin_memcpy(t, __builtin_assume_aligned(reinterpret_cast<const void*>(addr & ~(uintptr_t)3), 4), 12);
        w[r][0] = __builtin_amdgcn_alignbyte(t[1], t[0], skew);
        w[r][1] = __builtin_amdgcn_alignbyte(t[2], t[1], skew);
        w[r][2] = __builtin_amdgcn_alignbyte(0u, t[2], skew);
    }
    auto px = [&](int r, int c) -> int { return (int)((w[r][c >> 2] >> (8 * (c & 3))) & 0xFFu); };
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int i = 1; i <= 7; ++i)
#pragma unroll
        for (int j = 1; j <= 7; ++j) {
            int Ix = (px(i, j + 1) - px(i, j - 1)) * 2 + (px(i - 1, j + 1) - px(i - 1, j - 1)) +
                     (px(i + 1, j + 1) - px(i + 1, j - 1));
            int Iy = (px(i + 1, j) - px(i - 1, j)) * 2 + (px(i + 1, j - 1) - px(i - 1, j - 1)) +
                     (px(i + 1, j + 1) - px(i - 1, j + 1));
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
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

// Bitonic sorting network for the 512-thread selection kernels with the elements in registers:
// thread t holds elements E*t .. E*t + E - 1.  Compare-exchange distances below E stay inside a thread,
// distances below 64*E are lane shuffles inside a wavefront, and only the larger ones go through
// LDS with a workgroup barrier each - 9 barriers instead of 55 for 1024 elements.
template <int E>
__device__ void bitonic_sort_regs(uint32_t* s, int P, int tid) {
    constexpr int NT = 512;
    const bool act = E * tid < P;
    uint32_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = act ? s[E * tid + e] : 0xFFFFFFFFu;
    for (int k = 2; k <= P; k <<= 1) {
        int j = k >> 1;
        if (j >= 64 * E) {
            if (act) {
#pragma unroll
                for (int e = 0; e < E; ++e) s[E * tid + e] = v[e];
            }
            __syncthreads();
            for (; j >= 64 * E; j >>= 1) {
                for (int i = tid; i < P; i += NT) {
                    int ixj = i ^ j;
                    if (ixj > i) {
                        uint32_t a = s[i], b = s[ixj];
                        bool asc = (i & k) == 0;
                        if ((a > b) == asc) {
                            s[i] = b;
                            s[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
            if (act) {
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = s[E * tid + e];
            }
        }
        for (; j >= E; j >>= 1) {   // partner in another lane of the wavefront
            const int lx = j / E;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = E * tid + e;
                const uint32_t w = (uint32_t)__shfl_xor((int)v[e], lx);
                const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                v[e] = keep_min ? min(v[e], w) : max(v[e], w);
            }
        }
#pragma unroll
        for (int jj = E >> 1; jj >= 1; jj >>= 1) {   // partner in the same thread
            if (jj > j) continue;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (e & jj) continue;
                const bool asc = ((E * tid + e) & k) == 0;
                const uint32_t lo = min(v[e], v[e | jj]), hi = max(v[e], v[e | jj]);
                v[e] = asc ? lo : hi;
                v[e | jj] = asc ? hi : lo;
            }
        }
    }
    if (act) {
#pragma unroll
        for (int e = 0; e < E; ++e) s[E * tid + e] = v[e];
    }
    __syncthreads();
}

// P (a power of two, at most CAP) elements of s ascending, 512 threads.
template <int CAP>
__device__ void sort_asc_512(uint32_t* s, int P, int tid) {
    if (P <= 512) bitonic_sort_regs<1>(s, P, tid);
    else if (P == 1024) bitonic_sort_regs<2>(s, P, tid);
    else if (P == 2048) bitonic_sort_regs<4>(s, P, tid);
    else if (CAP >= 4096 && P == 4096) bitonic_sort_regs<8>(s, P, tid);
    else if (CAP >= 8192) bitonic_sort_regs<16>(s, P, tid);
}

// Selection runs as three kernels so that the per-candidate image reads (Harris) are
// spread over the whole batch instead of being serialised inside one workgroup:
//   select_a: per (level, image) workgroup - retainBest(2n) on the FAST score via a
//             256-bin LDS histogram, then raster order by an LDS bitonic sort;
//   harris:   one lane per surviving candidate, batch-wide;
//   select_b: per (level, image) workgroup - retainBest(n) on the Harris response,
//             ordered compaction.
// "retainBest(n)" keeps every element whose value is >= the n-th largest (ties survive).
template <int CAP>
__global__ __launch_bounds__(512) void select_a_kernel(Geom g, const uint32_t* __restrict__ cand,
                                                       const uint32_t* __restrict__ cand_cnt,
                                                       uint32_t* __restrict__ sel_keys, uint32_t* __restrict__ sel_cnt,
                                                       uint32_t* __restrict__ flags, uint32_t* __restrict__ img_status) {
    __shared__ uint32_t keys[CAP];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_n, s_thr, s_ge, s_cnt;

    // image index fastest: consecutive workgroups go to different XCDs, and with the level
    // fastest one XCD would get every image's (largest) level-0 list
    const int l = blockIdx.y, img = blockIdx.x, tid = threadIdx.x;
    constexpr int NT = 512;
    const LevelGeom& L = g.L[l];
    uint32_t total = min(cand_cnt[img * kLevels + l], (uint32_t)L.cand_cap);
    const uint32_t* cl = cand + (size_t)img * g.cand_stride + L.cand_off;
    const int quota = L.quota;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (total == 0 || quota == 0) {
        if (tid == 0) sel_cnt[img * kLevels + l] = 0;
        return;
    }
    uint32_t thr_score = 0;
    if (total > (uint32_t)(2 * quota)) {
        for (uint32_t i = tid; i < total; i += NT) atomicAdd(&hist[cl[i] >> 24], 1u);
        __syncthreads();
        if (tid < 64) {
            // suffix sums of the histogram, four bins per lane, then the largest score s
            // with count(score >= s) >= 2*quota
            uint32_t h0 = hist[tid * 4], h1 = hist[tid * 4 + 1], h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
            uint32_t mine = h0 + h1 + h2 + h3, suf = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t v = __shfl_down(suf, o);
                if (tid + o < 64) suf += v;
            }
            uint32_t above = suf - mine;   // bins of higher lanes
            uint32_t need = (uint32_t)(2 * quota);
            int s = -1;
            if (above + h3 >= need) s = tid * 4 + 3;
            else if (above + h3 + h2 >= need) s = tid * 4 + 2;
            else if (above + h3 + h2 + h1 >= need) s = tid * 4 + 1;
            else if (suf >= need) s = tid * 4;
            unsigned long long m = __ballot(s >= 0);
            int top = 63 - __builtin_clzll(m);   // total > 2*quota guarantees m != 0
            if (tid == top) {
                s_thr = (uint32_t)s;
                // number of candidates with a score >= s
                const int k = s - tid * 4;
                s_ge = above + (k <= 3 ? h3 : 0u) + (k <= 2 ? h2 : 0u) + (k <= 1 ? h1 : 0u) + (k <= 0 ? h0 : 0u);
            }
        }
        __syncthreads();
        thr_score = s_thr;
    }
    // Scores are 8-bit, so the cut can land in a large group of equal scores; all of them stay
    // (retainBest keeps ties).  When that makes the list longer than the LDS table, the call is
    // flagged as truncated and the table is filled with everything above the cut score plus the
    // first cut-score candidates in raster order - the same set on every run, not whichever
    // lanes won a slot: key_cut = the largest K with |{score == cut, key < K}| <= room, found
    // bit by bit (keys are unique inside a level).
    uint32_t key_cut = 0x01000000u;
    if (thr_score > 0 && s_ge > (uint32_t)CAP) {
        const uint32_t room = (uint32_t)CAP - (s_ge - hist[thr_score]);   // fewer than 2 * quota lie above the cut
        uint32_t K = 0;
        for (int bit = 23; bit >= 0; --bit) {
            const uint32_t trial = K | (1u << bit);
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            uint32_t mine = 0;
            for (uint32_t i = tid; i < total; i += NT) {
                const uint32_t c = cl[i];
                mine += ((c >> 24) == thr_score && (c & 0x00FFFFFFu) < trial) ? 1u : 0u;
            }
            if (mine) atomicAdd(&s_cnt, mine);
            __syncthreads();
            if (s_cnt <= room) K = trial;
            __syncthreads();
        }
        key_cut = K;
        if (tid == 0) {
            atomicOr(flags, (uint32_t)VO_TRUNC_SCORE_LIST);
            atomicOr(&img_status[img], (uint32_t)VO_TRUNC_SCORE_LIST);
        }
    }
    for (uint32_t i = tid; i < total; i += NT) {
        uint32_t c = cl[i];
        const uint32_t sc = c >> 24, key = c & 0x00FFFFFFu;
        if (sc > thr_score || (sc == thr_score && key < key_cut)) {
            uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < (uint32_t)CAP) keys[slot] = key;
        }
    }
    __syncthreads();
    const uint32_t M = min(s_n, (uint32_t)CAP);   // (s_n <= CAP by construction)
    int P = 1;
    while (P < (int)M) P <<= 1;
    for (int i = M + tid; i < P; i += NT) keys[i] = 0xFFFFFFFFu;
    __syncthreads();
    sort_asc_512<CAP>(keys, P, tid);   // raster order: key = y << 12 | x
    uint32_t* out = sel_keys + ((size_t)img * kLevels + l) * CAP;
    for (int i = tid; i < (int)M; i += NT) out[i] = keys[i];
    if (tid == 0) sel_cnt[img * kLevels + l] = M;
}

// One lane per surviving candidate.  A level gets as many 256-lane blocks as its list after the
// first cut normally needs (2 * quota plus a margin; hb.first[l] .. hb.first[l + 1]) and the
// blocks stride over the list, so ties at the cut that make it longer are still covered; sizing
// every level for the longest possible list launched four times as many workgroups as had work,
// and the kernel ran at the rate workgroups can be dispatched.
struct HarrisBlocks {
    int first[kLevels + 1];
};
__global__ __launch_bounds__(256) void harris_kernel(Geom g, int cap, HarrisBlocks hb, const uint8_t* __restrict__ in,
                                                     const uint8_t* __restrict__ pyr,
                                                     const uint32_t* __restrict__ sel_keys,
                                                     const uint32_t* __restrict__ sel_cnt, float* __restrict__ sel_resp) {
    const int per_img = hb.first[kLevels];
    uint32_t work;
    if (!xcd_remap((uint32_t)per_img * (uint32_t)g.n_images, &work)) return;
    const int img = (int)(work / (uint32_t)per_img);
    const int rem = (int)(work - (uint32_t)img * (uint32_t)per_img);
    int l = 0;
#pragma unroll
    for (int k = 1; k < kLevels; ++k) l += (rem >= hb.first[k]);
    const int nblk = hb.first[l + 1] - hb.first[l];
    const int count = (int)sel_cnt[img * kLevels + l];
    int pitch;
    const uint8_t* base = level_base(g, in, pyr, img, l, &pitch);
    for (int i = (rem - hb.first[l]) * 256 + (int)threadIdx.x; i < count; i += nblk * 256) {
        size_t o = ((size_t)img * kLevels + l) * cap + i;
        uint32_t k = sel_keys[o];
        sel_resp[o] = harris_at(base, pitch, k & 0xFFF, k >> 12);
    }
}

template <int CAP>
__global__ __launch_bounds__(512) void select_b_kernel(Geom g, const uint32_t* __restrict__ sel_keys,
                                                       const uint32_t* __restrict__ sel_cnt,
                                                       const float* __restrict__ sel_resp, uint16_t* __restrict__ kp_x,
                                                       uint16_t* __restrict__ kp_y, float* __restrict__ kp_resp,
                                                       uint32_t* __restrict__ kp_cnt, uint32_t* __restrict__ flags,
                                                       uint32_t* __restrict__ img_status) {
    __shared__ uint32_t inv[CAP];    // ~sortable(response): ascending == best first
    __shared__ uint32_t srt[CAP];
    __shared__ uint32_t wsum[8];
    __shared__ uint32_t s_tot;
    const int l = blockIdx.y, img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = 512;
    const LevelGeom& L = g.L[l];
    const int quota = L.quota;
    const int M = (int)sel_cnt[img * kLevels + l];
    if (M == 0) {
        if (tid == 0) kp_cnt[img * kLevels + l] = 0;
        return;
    }
    const size_t ib = ((size_t)img * kLevels + l) * CAP;
    int P = 1;
    while (P < M) P <<= 1;
    for (int i = tid; i < P; i += NT) {
        uint32_t v = i < M ? ~f2sortable(sel_resp[ib + i]) : 0xFFFFFFFFu;
        inv[i] = v;
        srt[i] = v;
    }
    __syncthreads();
    uint32_t inv_thr = 0xFFFFFFFFu;   // keep iff inv <= inv_thr
    if (M > quota) {
        sort_asc_512<CAP>(srt, P, tid);
        inv_thr = srt[quota - 1];
        __syncthreads();
    }
    // ordered compaction: ballot prefix inside a wave, wave totals through LDS
    uint32_t base_out = 0;
    uint32_t F_cap = (uint32_t)L.kp_cap;
    size_t ob = (size_t)img * g.kp_stride + L.kp_off;
    for (int i0 = 0; i0 < M; i0 += NT) {
        int i = i0 + tid;
        bool keep = i < M && inv[i] <= inv_thr;
        unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = base_out;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            uint32_t pos = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < F_cap) {
                uint32_t k = sel_keys[ib + i];
                kp_x[ob + pos] = (uint16_t)(k & 0xFFF);
                kp_y[ob + pos] = (uint16_t)(k >> 12);
                kp_resp[ob + pos] = sel_resp[ib + i];
            }
        }
        if (tid == 0) {
            uint32_t t = 0;
            for (int w = 0; w < 8; ++w) t += wsum[w];
            s_tot = t;
        }
        __syncthreads();
        base_out += s_tot;
        __syncthreads();
    }
    if (tid == 0) {
        if (base_out > F_cap) {   // the first F_cap in raster order were kept
            atomicOr(flags, (uint32_t)VO_TRUNC_KEYPOINT_LIST);
            atomicOr(&img_status[img], (uint32_t)VO_TRUNC_KEYPOINT_LIST);
            base_out = F_cap;
        }
        kp_cnt[img * kLevels + l] = base_out;
    }
}

// ------------------------------------------------------------------ blur
// 7x7 separable fixed-point blur (taps sum to 257, 8 fractional bits each),
// BORDER_REFLECT_101.  Column stage rounds sum/65536 half-to-even for columns
// below (w & ~3) and half-up for the rest, saturating to 255.
constexpr int BT_W = 64, BT_H = 58;
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ uint32_t sat_pk_u8_i16(uint32_t v) {   // two i16 -> two saturated u8 (low half)
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// Tile = 64 x 58 outputs from 64 source rows (3-row halo), 256 lanes.
// Row pass: a lane takes two vertically adjacent source rows and four columns; each row
// sum is two v_dot4_u32_u8 on byte-aligned windows (v_alignbyte) of three source dwords,
// read straight from global memory on interior tiles and from a reflected LDS copy on
// tiles that touch the border.  The two rows' u16 sums of a column share one LDS dword.
// Column pass: a lane takes two adjacent output rows and four columns; both rows read
// the same four row-pair dwords, so a 7-tap column sum is four v_dot2_u32_u16 with the
// taps paired (t0,t1)(t2,t3)(t4,t5)(t6,0) for the even row and (0,t0)(t1,t2)(t3,t4)(t5,t6)
// for the odd one.
__global__ __launch_bounds__(256) void blur_kernel(Geom g, SmallTabs tabs, const uint8_t* __restrict__ in,
                                                   const uint8_t* __restrict__ pyr, const uint32_t* __restrict__ tab,
                                                   uint8_t* __restrict__ blur) {
    constexpr int PW = BT_W + 8, PH = BT_H + 6;          // LDS column 0 <-> image column x0 - 4
    static_assert(PH == 64, "row pass covers 32 row pairs in two sweeps of 256 lanes");
    __shared__ __attribute__((aligned(16))) uint8_t px[PH * PW];
    __shared__ __attribute__((aligned(16))) uint32_t hs[(PH / 2) * BT_W];
    uint32_t work;
    if (!xcd_remap((uint32_t)g.blur_tiles * (uint32_t)g.n_images, &work)) return;
    int img, tile;
    split_work(work, (uint32_t)g.blur_tiles, g.blur_magic, &img, &tile);
    const int tid = threadIdx.x;
    const uint32_t te = tab[g.blur_ttab + tile];
    const int l = (int)(te >> 24);
    const LevelGeom& L = g.L[l];
    int x0 = (int)(te & 0xFFFu) * BT_W, y0 = (int)((te >> 12) & 0xFFFu) * BT_H;
    int pitch;
    const uint8_t* base = level_base(g, in, pyr, img, l, &pitch);
    const bool aligned = ((((uintptr_t)base) | (unsigned)pitch) & 3) == 0;
    // Every tile of a level wider than one tile reads its source rows straight from global
    // memory: rows beyond the top / bottom border are reflected per lane, at the left border the
    // four columns -4..-1 are the mirror image of columns 4..1 (one v_perm), and in a level's
    // last column of tiles (`edge`) the one or two dwords of a row that reach past the last
    // column are picked, with one v_perm, out of the row's last eight bytes - which hold every
    // column that BORDER_REFLECT_101 can ask for - by the lanes that own them.  (Staging the
    // whole window of those tiles in LDS byte by byte made them cost 3.5 times a direct tile:
    // a quarter of the kernel's vector instructions on a seventh of its tiles.)
    const bool edge = x0 + BT_W + 4 > L.w;
    const bool direct = !edge || x0 > 0;

    const uint32_t t0123 = (uint32_t)tabs.taps[0] | ((uint32_t)tabs.taps[1] << 8) | ((uint32_t)tabs.taps[2] << 16) |
                           ((uint32_t)tabs.taps[3] << 24);
    const uint32_t t456 = (uint32_t)tabs.taps[4] | ((uint32_t)tabs.taps[5] << 8) | ((uint32_t)tabs.taps[6] << 16);
    // four row sums (outputs 4*gx .. 4*gx+3 read source columns 4*gx-3+j .. 4*gx+3+j,
    // i.e. bytes 1+j .. 7+j of the three dwords that start at column 4*gx-4)
    auto row4 = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t* o) {
        uint32_t a0 = __builtin_amdgcn_alignbyte(w1, w0, 1), b0 = __builtin_amdgcn_alignbyte(w2, w1, 1);
        uint32_t a1 = __builtin_amdgcn_alignbyte(w1, w0, 2), b1 = __builtin_amdgcn_alignbyte(w2, w1, 2);
        uint32_t a2 = __builtin_amdgcn_alignbyte(w1, w0, 3), b2 = __builtin_amdgcn_alignbyte(w2, w1, 3);
        o[0] = __builtin_amdgcn_udot4(a0, t0123, __builtin_amdgcn_udot4(b0, t456, 0u, false), false);
        o[1] = __builtin_amdgcn_udot4(a1, t0123, __builtin_amdgcn_udot4(b1, t456, 0u, false), false);
        o[2] = __builtin_amdgcn_udot4(a2, t0123, __builtin_amdgcn_udot4(b2, t456, 0u, false), false);
        o[3] = __builtin_amdgcn_udot4(w1, t0123, __builtin_amdgcn_udot4(w2, t456, 0u, false), false);
    };
    auto store_pair = [&](int rp, int gx, const uint32_t* e, const uint32_t* o) {
        uint4 v;
        v.x = e[0] | (o[0] << 16);
        v.y = e[1] | (o[1] << 16);
        v.z = e[2] | (o[2] << 16);
        v.w = e[3] | (o[3] << 16);
        *reinterpret_cast<uint4*>(&hs[rp * BT_W + 4 * gx]) = v;
    };

    if (direct) {
        const bool inner_rows = y0 >= 3 && y0 + BT_H + 3 <= L.h;   // source rows y0 - 3 .. y0 + 60 exist
        // Edge tiles: v_perm selector of the dword at column x out of the row's bytes w - 8 .. w - 1
        // (selector bytes 0..3: first dword, 4..7: second), and whether the dword needs it at all.
        // Columns beyond w + 2 feed no output inside the image; their selector is clamped.
        const int gx_l = tid & 15;
        auto edge_sel = [&](int x, bool* fix) -> uint32_t {
            *fix = edge && x + 3 >= L.w;
            uint32_t s = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xs = x + b;
                const int src = xs < L.w ? xs : 2 * (L.w - 1) - xs;
                s |= (uint32_t)min(max(src - (L.w - 8), 0), 7) << (8 * b);
            }
            return s;
        };
        bool fix_c = false, fix_h = false;
        uint32_t sel_c = 0, sel_h = 0;
        if (edge) {   // workgroup-uniform
            sel_c = edge_sel(x0 + 4 * gx_l, &fix_c);
            sel_h = edge_sel(x0 + 4 * gx_l + 4, &fix_h);   // (only lane 15 of a row uses it)
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int item = tid + 256 * it, rp = item >> 4, gx = item & 15;
            // A lane loads only its own dword of each row (columns x0 + 4*gx ..); the dwords left
            // and right of it are its neighbours' (DPP row shifts inside the 16 lanes of a tile
            // row), and the two lanes at the ends of the row fetch the halo dword instead: a third
            // of the load traffic of three dwords per lane, on a kernel that keeps the texture
            // data path busier than the vector ALUs.
            const bool left = x0 == 0 && gx == 0;   // no columns -4..-1: mirrored below
            int re = y0 - 3 + 2 * rp, ro = re + 1;
            if (!inner_rows) {   // workgroup-uniform: only tiles at the top or bottom border reflect
                re = reflect101(re, L.h);
                ro = reflect101(ro, L.h);
            }
            const uint32_t ye = (uint32_t)__mul24(re, pitch), yo = (uint32_t)__mul24(ro, pitch);
            const uint32_t cx = (uint32_t)(x0 + 4 * gx);
            // (a dword that reaches past the last column is loaded from the row's last four bytes
            // instead - a valid address - and replaced below)
            uint32_t e1 = load_u32_unaligned(base + ye + (fix_c ? (uint32_t)(L.w - 4) : cx));
            uint32_t o1 = load_u32_unaligned(base + yo + (fix_c ? (uint32_t)(L.w - 4) : cx));
            uint32_t he = 0, ho = 0;
            if ((gx == 0 && !left) || gx == 15) {
                const uint32_t hx = gx == 0 ? cx - 4 : (fix_h ? (uint32_t)(L.w - 4) : cx + 4);
                he = load_u32_unaligned(base + ye + hx);
                ho = load_u32_unaligned(base + yo + hx);
            }
            if (edge) {   // workgroup-uniform
                const uint32_t ae = load_u32_unaligned(base + ye + (uint32_t)(L.w - 8)), be = load_u32_unaligned(base + ye + (uint32_t)(L.w - 4));
                const uint32_t ao = load_u32_unaligned(base + yo + (uint32_t)(L.w - 8)), bo = load_u32_unaligned(base + yo + (uint32_t)(L.w - 4));
                if (fix_c) {
                    e1 = __builtin_amdgcn_perm(be, ae, sel_c);
                    o1 = __builtin_amdgcn_perm(bo, ao, sel_c);
                }
                if (fix_h && gx == 15) {
                    he = __builtin_amdgcn_perm(be, ae, sel_h);
                    ho = __builtin_amdgcn_perm(bo, ao, sel_h);
                }
            }
            // row_shr:1 / row_shl:1; the lane without a neighbour keeps the halo value
            uint32_t e0 = (uint32_t)__builtin_amdgcn_update_dpp((int)he, (int)e1, 0x111, 0xF, 0xF, false);
            uint32_t e2 = (uint32_t)__builtin_amdgcn_update_dpp((int)he, (int)e1, 0x101, 0xF, 0xF, false);
            uint32_t o0 = (uint32_t)__builtin_amdgcn_update_dpp((int)ho, (int)o1, 0x111, 0xF, 0xF, false);
            uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp((int)ho, (int)o1, 0x101, 0xF, 0xF, false);
            if (left) {   // columns -4..-1 are the mirror image of columns 4..1
                e0 = __builtin_amdgcn_perm(e2, e1, 0x01020304u);
                o0 = __builtin_amdgcn_perm(o2, o1, 0x01020304u);
            }
            uint32_t se[4], so[4];
            row4(e0, e1, e2, se);
            row4(o0, o1, o2, so);
            store_pair(rp, gx, se, so);
        }
    } else {
        // reflected copy of the source window; whole dwords where they lie inside the row
        for (int i = tid; i < PH * (PW / 4); i += 256) {
            int r = i / (PW / 4), c = i - r * (PW / 4);
            int y = reflect101(y0 - 3 + r, L.h);
            int xs = x0 - 4 + 4 * c;
            const uint8_t* row = base + (size_t)y * pitch;
            uint32_t v;
            if (aligned && xs >= 0 && xs + 3 < L.w) {
                v = *reinterpret_cast<const uint32_t*>(row + xs);
            } else {
                v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int x = reflect101(min(xs + b, L.w + 3), L.w);
                    v |= (uint32_t)row[x] << (8 * b);
                }
            }
            *reinterpret_cast<uint32_t*>(&px[r * PW + 4 * c]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int item = tid + 256 * it, rp = item >> 4, gx = item & 15;
            const uint32_t* pe = reinterpret_cast<const uint32_t*>(&px[(2 * rp) * PW + 4 * gx]);
            const uint32_t* po = reinterpret_cast<const uint32_t*>(&px[(2 * rp + 1) * PW + 4 * gx]);
            uint32_t se[4], so[4];
            row4(pe[0], pe[1], pe[2], se);
            row4(po[0], po[1], po[2], so);
            store_pair(rp, gx, se, so);
        }
    }
    __syncthreads();

    const uint32_t T0 = (uint32_t)tabs.taps[0], T1 = (uint32_t)tabs.taps[1], T2 = (uint32_t)tabs.taps[2], T3 = (uint32_t)tabs.taps[3],
                   T4 = (uint32_t)tabs.taps[4], T5 = (uint32_t)tabs.taps[5], T6 = (uint32_t)tabs.taps[6];
    const uint32_t ke0 = T0 | (T1 << 16), ke1 = T2 | (T3 << 16), ke2 = T4 | (T5 << 16), ke3 = T6;
    const uint32_t ko0 = T0 << 16, ko1 = T1 | (T2 << 16), ko2 = T3 | (T4 << 16), ko3 = T5 | (T6 << 16);
    const int vec_end = L.w & ~3;
    const bool all_even = x0 + BT_W <= vec_end;
    uint8_t* out = blur + (size_t)img * g.blk_stride + L.off;   // blur block uses the same offsets
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int item = tid + 256 * it, op = item >> 4, gx = item & 15;
        int X = x0 + 4 * gx, Y = y0 + 2 * op;
        if (op >= BT_H / 2 || X >= L.w || Y >= L.h) continue;
        uint4 p0 = *reinterpret_cast<const uint4*>(&hs[(op + 0) * BT_W + 4 * gx]);
        uint4 p1 = *reinterpret_cast<const uint4*>(&hs[(op + 1) * BT_W + 4 * gx]);
        uint4 p2 = *reinterpret_cast<const uint4*>(&hs[(op + 2) * BT_W + 4 * gx]);
        uint4 p3 = *reinterpret_cast<const uint4*>(&hs[(op + 3) * BT_W + 4 * gx]);
        const uint32_t c0[4] = {p0.x, p0.y, p0.z, p0.w}, c1[4] = {p1.x, p1.y, p1.z, p1.w};
        const uint32_t c2[4] = {p2.x, p2.y, p2.z, p2.w}, c3[4] = {p3.x, p3.y, p3.z, p3.w};
        uint32_t se[4], so[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            se[j] = udot2_u16(c0[j], ke0, udot2_u16(c1[j], ke1, udot2_u16(c2[j], ke2, udot2_u16(c3[j], ke3, 0u))));
            so[j] = udot2_u16(c0[j], ko0, udot2_u16(c1[j], ko1, udot2_u16(c2[j], ko2, udot2_u16(c3[j], ko3, 0u))));
            // sum / 65536 rounded half-to-even below vec_end, half-up from there on
            if (all_even) {   // workgroup-uniform: every tile but a level's last column of tiles
                se[j] += 32767u + ((se[j] >> 16) & 1u);
                so[j] += 32767u + ((so[j] >> 16) & 1u);
            } else {
                uint32_t ev = (X + j) < vec_end ? 1u : 0u;
                se[j] += 32768u - ev + (ev & (se[j] >> 16));
                so[j] += 32768u - ev + (ev & (so[j] >> 16));
            }
        }
        // high halves -> i16 pairs -> saturated bytes
        uint32_t e01 = sat_pk_u8_i16(__builtin_amdgcn_perm(se[1], se[0], 0x07060302u));
        uint32_t e23 = sat_pk_u8_i16(__builtin_amdgcn_perm(se[3], se[2], 0x07060302u));
        uint32_t o01 = sat_pk_u8_i16(__builtin_amdgcn_perm(so[1], so[0], 0x07060302u));
        uint32_t o23 = sat_pk_u8_i16(__builtin_amdgcn_perm(so[3], so[2], 0x07060302u));
        // the row pitch is a multiple of 64, so the dword store stays inside the row even when
        // the level width is not a multiple of 4 (the surplus bytes are never read as pixels)
        *reinterpret_cast<uint32_t*>(out + (size_t)Y * L.pitch + X) = e01 | (e23 << 16);
        if (Y + 1 < L.h) *reinterpret_cast<uint32_t*>(out + (size_t)(Y + 1) * L.pitch + X) = o01 | (o23 << 16);
    }
}

// ------------------------------------------------------------------ describe
// (float)cos / (float)sin of a float angle in [0, 2*pi], evaluated in double so that the
// rounded float equals the one a double libm gives: Cody-Waite reduction by pi/2 and the
// classic degree-13/14 minimax kernels on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_f64(double x, double* sn, double* cs) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    double kd = rint(x * two_over_pi);
    int k = (int)kd;
    double r = (x - kd * pio2_hi) - kd * pio2_lo;
    double z = r * r;
    double ps = 8.33333333332248946124e-03 +
                z * (-1.98412698298579493134e-04 +
                     z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    double pc = z * (4.16666666666666019037e-02 +
                     z * (-1.38888888888741095749e-03 +
                          z * (2.48015872894767294178e-05 +
                               z * (-2.75573143513906633035e-07 +
                                    z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    double c = 1.0 - (0.5 * z - z * pc);
    switch (k & 3) {
        case 0: *sn = s; *cs = c; break;
        case 1: *sn = c; *cs = -s; break;
        case 2: *sn = -s; *cs = -c; break;
        default: *sn = -c; *cs = s; break;
    }
}

// Orientation and keypoint records.  Half a wavefront (32 lanes) per keypoint, two keypoints
// per wave, grid-stride over an image's keypoints.  The 31 rows x 48 bytes that hold the 31x31
// disc (from the 16-byte boundary at or below column x - 15) go to LDS with three 16-byte loads
// per lane - a load instruction costs the texture addresser, which bounded this kernel, the
// same for 16 bytes per lane as for 4, and less the fewer rows it touches.  Intensity-centroid
// angle: the half-wave sweeps the disc in eight steps of four rows, lane = (row of the step,
// dword of the 32-byte row window); a step is two aligned LDS dwords shifted into place
// (v_alignbyte), the disc byte mask and two v_dot4_u32_u8 (sum and column-weighted sum), the row
// weight is applied to the row sum.  The first lane of the half writes the final keypoint.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void orient_kernel(Geom g, int bx, const uint8_t* __restrict__ in,
                                                     const uint8_t* __restrict__ pyr,
                                                     const uint16_t* __restrict__ kp_x,
                                                     const uint16_t* __restrict__ kp_y,
                                                     const float* __restrict__ kp_resp,
                                                     const uint32_t* __restrict__ kp_cnt, vo_keypoint* __restrict__ kps,
                                                     int cap, int32_t* __restrict__ counts, uint32_t* __restrict__ flags,
                                                     uint32_t* __restrict__ img_status) {
    constexpr int OW_ROWS = 2 * kHalfPatch + 1, OW_PITCH = 48;
    __shared__ __attribute__((aligned(16))) uint8_t win[8][OW_ROWS * OW_PITCH];
    uint32_t work;
    if (!xcd_remap((uint32_t)bx * (uint32_t)g.n_images, &work)) return;
    const int img = (int)(work / (uint32_t)bx), blk = (int)(work - (uint32_t)img * (uint32_t)bx);
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const uint32_t* cnt = kp_cnt + img * kLevels;
    int start[kLevels + 1];
    start[0] = 0;
#pragma unroll
    for (int k = 0; k < kLevels; ++k) start[k + 1] = start[k] + (int)cnt[k];
    const int total = start[kLevels];
    if (blk == 0 && threadIdx.x == 0) {
        counts[img] = min(total, cap);
        if (total > cap) {   // the first `cap` in level-major raster order are returned
            atomicOr(flags, (uint32_t)VO_TRUNC_OUTPUT);
            atomicOr(&img_status[img], (uint32_t)VO_TRUNC_OUTPUT);
        }
    }
    const int n = min(total, cap);
    // this lane's disc masks, the same for every keypoint, are read from LDS in the loop: eight
    // registers less are a seventh wavefront per SIMD, and the kernel waits for loads, not for LDS
    __shared__ uint32_t amask_l[8 * 32];
    amask_l[threadIdx.x] = d_angle_mask[threadIdx.x];
    const int rr = hl >> 3, dj = hl & 7;
    const uint32_t wcol = 0x03020100u + 0x04040404u * (uint32_t)dj;   // window byte index = u + 15
    // staging: element i = 32 * t + hl of the 31 x 3 sixteen-byte pieces, row i / 3, piece i % 3
    int srow[3], sseg[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int i = 32 * t + hl;
        srow[t] = i / 3;
        sseg[t] = i - 3 * srow[t];
    }
    uint8_t* my_win = win[(threadIdx.x >> 6) * 2 + half];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // The levels' geometry, looked up per keypoint, comes from LDS: fetched from the kernel
    // arguments with a per-lane level it is three dependent memory round trips per keypoint
    // (candidate offset -> coordinates; level offset; pitch), and the kernel's time is the length of
    // that chain.  For the same reason the next keypoint's coordinates and response are requested
    // while the current window is on its way.
    __shared__ u32x4 lvl[kLevels];   // candidate offset, byte offset of the level, pitch, scale
    if (threadIdx.x < kLevels) {
        const int k = threadIdx.x;
        lvl[k] = u32x4{(uint32_t)g.L[k].kp_off, k ? (uint32_t)g.L[k].off : 0u, k ? (uint32_t)g.L[k].pitch : (uint32_t)g.step,
                       __float_as_uint(g.L[k].scale)};
    }
    __syncthreads();
    const uint8_t* base0 = in + (size_t)img * g.in_stride;
    const uint8_t* base1 = pyr + (size_t)img * g.blk_stride;
    const size_t ib0 = (size_t)img * g.kp_stride;
    int l_nx = 0, oi_nx = 0, x_nx = 0, y_nx = 0;
    bool live_nx = false;
    float resp_nx = 0.f;
    auto locate = [&](int o2) {
        live_nx = o2 + half < n;              // an odd count leaves the last second half idle
        oi_nx = live_nx ? o2 + half : o2;     // (it repeats the first half's work, writes nothing)
        int l = 0, lstart = 0;
#pragma unroll
        for (int k = 1; k < kLevels; ++k) {
            l += (oi_nx >= start[k]);
            lstart = (oi_nx >= start[k]) ? start[k] : lstart;
        }
        l_nx = l;
        const size_t ib = ib0 + lvl[l].x + (uint32_t)(oi_nx - lstart);
        x_nx = kp_x[ib];
        y_nx = kp_y[ib];
        resp_nx = kp_resp[ib];
    };
    const int o2_first = (blk * 4 + (threadIdx.x >> 6)) * 2;
    if (o2_first < n) locate(o2_first);
    for (int o2 = o2_first; o2 < n; o2 += bx * 8) {
        const bool live = live_nx;
        const int oi = oi_nx, l = l_nx, x = x_nx, y = y_nx;
        const float resp = resp_nx;
        const u32x4 lv = lvl[l];
        const int lpitch = (int)lv.z;
        const uint8_t* lbase = (l == 0 ? base0 : base1) + lv.y;
        // every address within 16 px of a keypoint is inside the level (keypoints keep a 31-px
        // margin); the 48 bytes of a row end at most one byte past column x + 16 + 15, still
        // inside the image buffer because the rows below the window exist
        const uint8_t* p0 = lbase + (size_t)(y - kHalfPatch) * lpitch + (x - kHalfPatch);
        const int skew = (int)((uintptr_t)p0 & 15);   // the same for every row when the pitch is a multiple of 16
        p0 -= skew;
        u32x4 sv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
            __builtin_memcpy(&sv[t], p0 + (size_t)min(srow[t], OW_ROWS - 1) * lpitch + 16 * sseg[t], 16);
        locate(o2 + bx * 8 < n ? o2 + bx * 8 : o2);   // (after the last keypoint its own once more: no branch,
                                                      // so that the window loads can be waited for by count)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous keypoint's reads are done
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (srow[t] < OW_ROWS) *reinterpret_cast<u32x4*>(&my_win[srow[t] * OW_PITCH + 16 * sseg[t]]) = sv[t];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // this lane's dword of step t starts skew + 4 * dj bytes into row 4 * t + rr
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(my_win) + rr * (OW_PITCH / 4) + ((skew >> 2) + dj);
        const uint32_t sh = (uint32_t)skew & 3u;
        uint32_t S = 0, X = 0;
        int m01 = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // (rows 31 of the last step do not exist: their mask is empty, the address is clamped)
            const uint32_t* q = wp + min(4 * t, OW_ROWS - 1 - rr) * (OW_PITCH / 4);
            uint32_t val = __builtin_amdgcn_alignbyte(q[1], q[0], sh) & amask_l[t * 32 + hl];
            uint32_t rs = __builtin_amdgcn_udot4(val, 0x01010101u, 0u, false);
            X = __builtin_amdgcn_udot4(val, wcol, X, false);
            S += rs;
            m01 += (4 * t - kHalfPatch + rr) * (int)rs;
        }
        int m10 = (int)X - kHalfPatch * (int)S;
        // sum over the 32 lanes of the half: four DPP steps inside a row of 16 lanes (pairs, quads,
        // half-row mirror, row mirror - no LDS round trip), one shuffle across the two rows
        auto row_sum = [](int x) -> int {
            x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
            x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
            x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);   // row_half_mirror
            x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);   // row_mirror
            return x;
        };
        m10 = row_sum(m10);
        m01 = row_sum(m01);
        m10 += __shfl_xor(m10, 16);
        m01 += __shfl_xor(m01, 16);
        if (hl == 0 && live) {
            const float scale = __uint_as_float(lv.w);
            vo_keypoint k;
            k.x = (float)x * scale;
            k.y = (float)y * scale;
            k.size = 31 * scale;
            k.angle = fast_atan2_deg((float)m01, (float)m10);
            k.response = resp;
            k.octave = l;
            k.class_id = -1;
            kps[(size_t)img * cap + oi] = k;
        }
    }
}

// rBRIEF descriptors from the finished keypoint records.  A wavefront takes `kpw` consecutive
// keypoints (at most 64).  First every lane prepares one of them - window origin, row pitch, and
// the (float) cosine and sine of its angle, evaluated in double: work that is the same for all
// lanes of a keypoint and used to be repeated by each of them.  Then half a wavefront per keypoint: the
// 37x37 blurred window around it goes to LDS with five 16-byte loads per lane (sweeps of
// eight 64-byte rows; the columns wanted are inside the level thanks to the 31-px
// keypoint margin - gathering the 512 test points from global memory instead costs one L1 tag
// look-up per lane and test), lane t evaluates tests t, t+32, ..., t+224 on it - both
// coordinates of a rotated test point in one packed single-precision instruction per step - and
// eight ballots give the 32 descriptor bytes (low word: first keypoint, high word: second).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 5) void describe_kernel(Geom g, int bx, int kpw, const uint8_t* __restrict__ blur,
                                                       const vo_keypoint* __restrict__ kps,
                                                       const int32_t* __restrict__ counts, uint8_t* __restrict__ desc,
                                                       int cap) {
    uint32_t work;
    if (!xcd_remap((uint32_t)bx * (uint32_t)g.n_images, &work)) return;
    const int img = (int)(work / (uint32_t)bx), blk = (int)(work - (uint32_t)img * (uint32_t)bx);
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = counts[img];
    const int r8 = hl >> 2, q16 = hl & 3;   // row of an eight-row sweep, 16-byte quarter of the 64-byte row
    const uint8_t* blur_img = blur + (size_t)img * g.blk_stride;
    // blurred window of +-18 px around the keypoint (the rotated test points stay inside:
    // the pattern's largest radius is 18.4), one per half-wave; a row holds the 64 bytes fetched
    // for it, pitch 80
    constexpr int DP_R = 18, DP_ROWS = 2 * DP_R + 1, DP_PITCH = 80;
    __shared__ __attribute__((aligned(16))) uint8_t patch[8][DP_ROWS * DP_PITCH];
    uint8_t* my_patch = patch[wave * 2 + half];
    uint8_t* my_rows = my_patch + (r8 * DP_PITCH + 16 * q16);   // this lane's piece of sweep t: + 8 * t * DP_PITCH
    const int row4 = min(32 + r8, DP_ROWS - 1);   // (the fifth sweep's rows 37..39 do not exist: clamped, not stored)
    for (int o_base = (blk * 4 + wave) * kpw; o_base < n; o_base += bx * 4 * kpw) {
        const int cnt = min(kpw, n - o_base);
        // ---- one keypoint per lane
        // (its values go to the 16 bytes that a 64-byte window row leaves unused in the 80-byte pitch:
        // keypoint 2 i + h into row i of half h's window, where the window copies never write)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            uint32_t k_a, k_b, k_org, k_pitch;
            const vo_keypoint kp = kps[(size_t)img * cap + o_base + min(lane, cnt - 1)];
            const LevelGeom& L = g.L[kp.octave];
            const float inv = 1.f / L.scale;
            const int cx = __float2int_rn(kp.x * inv), cy = __float2int_rn(kp.y * inv);
            const int pitch = L.pitch;
            // A load instruction costs the texture addresser about the same for 16 bytes per lane as
            // for 4, and what it costs grows with the number of rows it touches: a row is fetched as
            // four aligned 16-byte quarters (64 bytes from the 16-byte boundary at or below column
            // cx - 18; rows of the blur block are 64-byte aligned), eight rows per half-wave and
            // instruction, five instructions instead of thirteen dword sweeps.  The 37 columns
            // wanted begin `skew` = (cx - 18) & 15 bytes into the 64: the low four bits of the window's
            // offset, because level offsets and pitches are multiples of 64.
            k_org = L.off + (uint32_t)(__mul24(cy - DP_R, pitch) + cx - DP_R);
            k_pitch = (uint32_t)pitch;
            const float ar = kp.angle * (float)(M_PI / 180.f);
            double sd, cd;
            sincos_f64((double)ar, &sd, &cd);
            k_a = __float_as_uint((float)cd);
            k_b = __float_as_uint((float)sd);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4*>(&patch[wave * 2 + (lane & 1)][(lane >> 1) * DP_PITCH + 64]) = u32x4{k_a, k_b, k_org, k_pitch};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // This lane's eight tests, the same for every keypoint, are fetched here and not once per kernel
        // (the opaque pointer keeps the compiler from moving the loads up): 32 registers that, held
        // across the preparation above with its double-precision polynomials, exceed the budget of six
        // wavefronts per SIMD and would be spilled for the whole kernel.
        f32x2 p0[8], p1[8];
        {
            const int* patp = reinterpret_cast<const int*>(d_pattern);
            asm volatile("" : "+s"(patp));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pat = patp[k * 32 + hl];
                p0[k] = f32x2{(float)(int8_t)(pat & 0xFF), (float)(int8_t)((pat >> 8) & 0xFF)};
                p1[k] = f32x2{(float)(int8_t)((pat >> 16) & 0xFF), (float)(int8_t)(pat >> 24)};
            }
        }
        // ---- half a wavefront per keypoint.  The loads of the next keypoint's window are issued as soon
        // as the current one's have been written to LDS, so that they are in flight during its tests.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 pw[5];
        float a_nx, b_nx;
        uint32_t skew_nx;
        auto fetch = [&](int i) {
            // (an idle second half repeats the first half's work and writes nothing)
            const u32x4 kv = *reinterpret_cast<const u32x4*>(&patch[wave * 2 + (2 * i + half < cnt ? half : 0)][i * DP_PITCH + 64]);
            a_nx = __uint_as_float(kv.x);
            b_nx = __uint_as_float(kv.y);
            const uint32_t orgu = kv.z, pitch = kv.w;
            skew_nx = orgu & 15u;
            const uint32_t org0 = (orgu & ~15u) + 16u * (uint32_t)q16, pitch8 = 8u * pitch;
            const uint32_t orgr = org0 + __umul24((uint32_t)r8, pitch);
#pragma unroll
            for (int t = 0; t < 5; ++t)
                pw[t] = *reinterpret_cast<const u32x4*>(
                    blur_img + (t < 4 ? orgr + (uint32_t)t * pitch8 : org0 + __umul24((uint32_t)row4, pitch)));
        };
        const int last = (cnt - 1) >> 1;
        fetch(0);
        for (int i = 0; i <= last; ++i) {
            const bool live = 2 * i + half < cnt;     // an odd count leaves the last second half idle
            const float a = a_nx, b = b_nx;
            const uint32_t skew = skew_nx;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // previous keypoint's reads are done
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (8 * t + r8 < DP_ROWS) *reinterpret_cast<u32x4*>(my_rows + 8 * t * DP_PITCH) = pw[t];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            fetch(min(i + 1, last));   // (the last keypoint's once more: harmless, and no branch)
            // cvRound of a rotated coordinate v, |v| < 2^22: the float sum v + 1.5 * 2^23 is rounded to
            // an integer half-to-even like rint, and its bit pattern is 0x4B400000 + rint(v) - one
            // fast-class add instead of v_rndne + v_cvt; the bias is folded into the window offset
            // (the 24-bit multiply sees 0x400000 + row).
            // (x a - y b, x b + y a) as (x, x) * (a, b) - (y, y) * (b, -a): the same products and the same
            // sum or difference, rounded the same way, two coordinates per instruction
            const f32x2 ab = {a, b}, nba = {b, -a};
            const f32x2 kround = {12582912.f, 12582912.f};
            const uint32_t cbase = (uint32_t)(DP_R * DP_PITCH + DP_R) + skew - 0x400000u * (uint32_t)DP_PITCH - 0x4B400000u;
            // (the empty statements make the coordinate pairs opaque, so that the compiler selects its halves
            // in the multiply - op_sel - instead of keeping (x, x) and (y, y) copies of every test point
            // in registers across the loop, 64 instead of 32)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("" : "+v"(p0[k]));
                asm volatile("" : "+v"(p1[k]));
            }
            auto tap = [&](f32x2 pt) -> int {
                const f32x2 xx = __builtin_shufflevector(pt, pt, 0, 0), yy = __builtin_shufflevector(pt, pt, 1, 1);
                const f32x2 r = (xx * ab - yy * nba) + kround;
                return my_patch[__umul24(__float_as_uint(r.y), (uint32_t)DP_PITCH) + __float_as_uint(r.x) + cbase];
            };
            int t0[8], t1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                t0[k] = tap(p0[k]);
                t1[k] = tap(p1[k]);
            }
            uint32_t mine = 0;   // lanes 0..7 of each half keep one ballot word each
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned long long m = __ballot(t0[k] < t1[k]);
                uint32_t w = half ? (uint32_t)(m >> 32) : (uint32_t)m;
                if (hl == k) mine = w;
            }
            if (hl < 8 && live)
                *reinterpret_cast<uint32_t*>(desc + ((size_t)img * cap + (o_base + 2 * i + half)) * 32 + hl * 4) = mine;
        }
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
                tab.resize(align_up(tab.size(), 4));
                L.ncg = (uint32_t)(L.w + 3) / 4;
                L.ncg_magic = (uint32_t)((0x100000000ull + L.ncg - 1) / L.ncg);
                L.gtab = (uint32_t)tab.size();
                tab.resize(tab.size() + 12 * (size_t)L.ncg, 0u);
                for (uint32_t cg = 0; cg < L.ncg; ++cg) {
                    // eight source bytes starting at `base` cover the group's four columns; at the
                    // right edge the window is pulled back inside the row (a neighbour that would
                    // fall outside it always has weight 0), columns past the level width repeat
                    // the last one (they land in the row padding)
                    uint32_t* e = &tab[L.gtab + 12 * (size_t)cg];
                    const int sw = n.L[l - 1].w;
                    int base = std::min((int)(tab[L.xtab + 4 * cg] >> 16), sw - 8);
                    bool fast = sw >= 8;
                    for (int k = 0; k < 4 && fast; ++k) {
                        uint32_t t = tab[L.xtab + std::min<int>(4 * cg + k, L.w - 1)];
                        int o = (int)(t >> 16) - base;
                        uint32_t c1 = t & 0xFFFFu;
                        if (o < 0 || o > 7 || (o == 7 && c1 != 0)) { fast = false; break; }
                        e[4 + k] = 0x0C000C00u | (uint32_t)o | ((uint32_t)std::min(o + 1, 7) << 16);
                        e[8 + k] = (256u - c1) | (c1 << 16);
                    }
                    e[0] = (uint32_t)std::max(base, 0) | (fast ? 0u : 0x80000000u);
                }
            }
        }
        // FAST tiles: level 0, then the other levels (two launch groups); inside a group the sampled
        // tiles first (fast_raise_kernel: every eighth tile of a level, in horizontal pairs - a lone
        // 64-pixel tile uses half of the 128-byte lines it touches - on a diagonal pattern so that
        // they spread over rows and columns; a level of fewer than 24 tiles is sampled whole)
        n.fast_ttab = (uint32_t)tab.size();
        auto sampled = [&](int l, int tx, int ty) {
            return n.L[l].fast_tx * n.L[l].fast_ty < 24 || ((tx >> 1) + 3 * ty) % 8 == 0;
        };
        for (int l = 0; l < kLevels; ++l) {
            n.L[l].fast_ns = 0;
            for (int ty = 0; ty < n.L[l].fast_ty; ++ty)
                for (int tx = 0; tx < n.L[l].fast_tx; ++tx) n.L[l].fast_ns += sampled(l, tx, ty) ? 1 : 0;
        }
        n.fast_s0 = n.L[0].fast_ns;
        n.fast_s1 = 0;
        for (int l = 1; l < kLevels; ++l) n.fast_s1 += n.L[l].fast_ns;
        for (int group = 0; group < 2; ++group)
            for (int pass = 0; pass < 2; ++pass)   // 0: the sampled tiles, 1: the others
                for (int l = group ? 1 : 0; l < (group ? kLevels : 1); ++l) {
                    if (pass == 1) n.L[l].fast_u0 = (int)(tab.size() - n.fast_ttab);
                    for (int ty = 0; ty < n.L[l].fast_ty; ++ty)
                        for (int tx = 0; tx < n.L[l].fast_tx; ++tx)
                            if (sampled(l, tx, ty) == (pass == 0))
                                tab.push_back(((uint32_t)l << 24) | ((uint32_t)ty << 12) | (uint32_t)tx);
                }
        n.blur_ttab = (uint32_t)tab.size();
        for (int l = 0; l < kLevels; ++l)
            for (int ty = 0; ty < n.L[l].blur_ty; ++ty)
                for (int tx = 0; tx < n.L[l].blur_tx; ++tx) tab.push_back(((uint32_t)l << 24) | ((uint32_t)ty << 12) | (uint32_t)tx);
        n.fast_magic = ftile > 0 ? (uint32_t)std::min<unsigned long long>(0x100000000ull / (uint32_t)ftile, 0xFFFFFFFFull) : 0u;
        n.blur_magic = btile > 0 ? (uint32_t)std::min<unsigned long long>(0x100000000ull / (uint32_t)btile, 0xFFFFFFFFull) : 0u;
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
    if (c->device < 0 || c->device >= 64) return fail(c, VO_ERR_INVALID, "device index beyond 63");
    {
    std::lock_guard<std::mutex> pattern_lock(g_pattern_mutex);
    if (!g_pattern_loaded[c->device]) {
        VO_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(d_pattern), h_pattern, sizeof(h_pattern)));
        static const SmallTabs tabs = make_tabs();
        std::vector<uint32_t> am(8 * 32, 0u);
        for (int t = 0; t < 8; ++t)
            for (int hl = 0; hl < 32; ++hl) {
                int r = 4 * t - kHalfPatch + (hl >> 3);
                for (int b = 0; b < 4 && r <= kHalfPatch; ++b) {
                    int u = 4 * (hl & 7) + b - kHalfPatch;
                    if (std::abs(u) <= tabs.umax[std::abs(r)]) am[t * 32 + hl] |= 0xFFu << (8 * b);
                }
            }
        VO_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(d_angle_mask), am.data(), am.size() * 4));
        g_pattern_loaded[c->device] = true;
    }
    }
    size_t N = (size_t)n_images;
    int rc;
    if ((rc = ensure(c, c->pyr, N * g.blk_stride))) return rc;
    // (the descriptor kernel fetches whole 64-byte pieces of rows: up to 45 bytes past the last
    // column it needs)
    if ((rc = ensure(c, c->blur, N * g.blk_stride + 256))) return rc;
    if ((rc = ensure(c, c->cand, N * std::max(g.cand_stride, 1) * 4))) return rc;
    if ((rc = ensure(c, c->cand_cnt, N * kLevels * 4))) return rc;
    if ((rc = ensure(c, c->fast_thr, N * kLevels * 4))) return rc;
    if ((rc = ensure(c, c->fast_redo, 2 * (1 + N * kLevels) * 4))) return rc;   // a redo list per launch group
    if ((rc = ensure(c, c->kp_x, N * g.kp_stride * 2))) return rc;
    if ((rc = ensure(c, c->kp_y, N * g.kp_stride * 2))) return rc;
    if ((rc = ensure(c, c->kp_resp, N * g.kp_stride * 4))) return rc;
    if ((rc = ensure(c, c->sel_keys, N * kLevels * (size_t)g.sel_cap * 4))) return rc;
    if ((rc = ensure(c, c->sel_resp, N * kLevels * (size_t)g.sel_cap * 4))) return rc;
    if ((rc = ensure(c, c->sel_cnt, N * kLevels * 4))) return rc;
    if ((rc = ensure(c, c->kp_cnt, N * kLevels * 4))) return rc;
    if ((rc = ensure(c, c->img_status, 2 * N * 4))) return rc;
    if (!c->flags.p) {
        if ((rc = ensure(c, c->flags, 256))) return rc;
        VO_HIP(c, hipMemsetAsync(c->flags.p, 0, 256, c->stream));
    }
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
    // per-image truncation bits of this call; c->flags is the context's sticky word and is only
    // cleared by whoever reports it (take_truncation)
    // (two halves used alternately: the previous call's words may still be read by its tail on the
    // side stream; the call before that is done with them, because the previous call's orientation
    // kernel was ordered behind that tail)
    c->status_phase ^= 1;
    c->img_status_cur = (uint32_t*)c->img_status.p + (size_t)c->status_phase * (c->img_status.bytes / 8);
    VO_HIP(c, hipMemsetAsync(c->img_status_cur, 0, (size_t)N * 4, s));
    // level 0 (needs only the input) and the other levels are separate FAST launches
    const uint32_t tiles0 = (uint32_t)(kLevels > 1 ? g.L[1].fast_tile0 : g.fast_tiles);
    const uint32_t tiles_up = (uint32_t)g.fast_tiles - tiles0;
    unsigned long long* const stats = c->fast_stats_on ? (unsigned long long*)c->fast_stats.p : nullptr;
    auto launch_fast_tiles = [&](hipStream_t fs, uint32_t first, uint32_t count, bool raised) {
        const uint32_t magic = (uint32_t)std::min<unsigned long long>(0x100000000ull / count, 0xFFFFFFFFull);
        if (raised)
            hipLaunchKernelGGL(fast_kernel<FAST_RAISED>, dim3(xcd_grid(count * (uint32_t)N)), dim3(256), 0, fs, g, d_imgs, pyr,
                               (const uint32_t*)c->tab.p, first, count, magic, (uint32_t*)c->cand.p,
                               (uint32_t*)c->cand_cnt.p, stats, (const uint32_t*)c->fast_thr.p, nullptr, 0u);
        else
            hipLaunchKernelGGL(fast_kernel<FAST_PLAIN>, dim3(xcd_grid(count * (uint32_t)N)), dim3(256), 0, fs, g, d_imgs, pyr,
                               (const uint32_t*)c->tab.p, first, count, magic, (uint32_t*)c->cand.p,
                               (uint32_t*)c->cand_cnt.p, stats, nullptr, nullptr, 0u);
    };
    // Raised thresholds (fast_raise_kernel) pay on batches: the group becomes five launches, three
    // of them FAST passes, which a call on a few images only feels as launch latency.  The scene
    // counters describe the plain run, so they switch it off as well.  VO_OPT_FAST_RAISE = 0 / 1
    // switches it off / on whatever the batch (measurement and test aid).
    bool raise = (size_t)N * (size_t)g.fast_tiles >= 4096 && !c->fast_stats_on;
    if (c->opt[VO_OPT_FAST_RAISE] >= 0) raise = c->opt[VO_OPT_FAST_RAISE] == 1 && !c->fast_stats_on;
    c->fast_raised = raise;
    if (raise) VO_HIP(c, hipMemsetAsync(c->fast_redo.p, 0, c->fast_redo.bytes, s));   // (before the fork below)
    // one launch group: tiles [first, first + count) of the table, the first `ns` of them sampled,
    // levels [l_first, l_first + n_lev)
    auto launch_fast = [&](hipStream_t fs, uint32_t first, uint32_t count, uint32_t ns, int l_first, int n_lev) {
        if (!raise || ns >= count) {
            launch_fast_tiles(fs, first, count, false);
            return;
        }
        uint32_t* const redo_list = (uint32_t*)c->fast_redo.p + (l_first ? 1 + (size_t)N * kLevels : 0);
        uint32_t span = 0;   // unsampled tiles of the group's largest level
        for (int l = l_first; l < l_first + n_lev; ++l)
            span = std::max(span, (uint32_t)(g.L[l].fast_tx * g.L[l].fast_ty - g.L[l].fast_ns));
        if (ns > 0) launch_fast_tiles(fs, first, ns, false);
        hipLaunchKernelGGL(fast_raise_kernel, dim3(N, n_lev), dim3(256), 0, fs, g, l_first, (const uint32_t*)c->cand.p,
                           (const uint32_t*)c->cand_cnt.p, (uint32_t*)c->fast_thr.p);
        launch_fast_tiles(fs, first + ns, count - ns, true);
        hipLaunchKernelGGL(fast_check_kernel, dim3(N, n_lev), dim3(256), 0, fs, g, l_first, (const uint32_t*)c->cand.p,
                           (const uint32_t*)c->cand_cnt.p, (const uint32_t*)c->fast_thr.p, redo_list);
        // (a grid that fills the GPU a few times over; it returns at once when the list is empty)
        const uint32_t redo_grid = (uint32_t)std::min<unsigned long long>((unsigned long long)(count - ns) * N, 4096ull);
        hipLaunchKernelGGL(fast_kernel<FAST_REDO>, dim3(redo_grid), dim3(256), 0, fs, g, d_imgs, pyr,
                           (const uint32_t*)c->tab.p, 0u, 0u, 0u, (uint32_t*)c->cand.p, (uint32_t*)c->cand_cnt.p, stats,
                           (const uint32_t*)c->fast_thr.p, (const uint32_t*)redo_list, span);
    };
    // The resize chain is seven dependent, bandwidth- and latency-bound launches that leave the
    // vector ALUs mostly idle, FAST is bound by them: level 0's FAST runs beside the chain on the
    // auxiliary stream.  With event brackets on the resize group everything stays on one stream,
    // so that the group's time is that of its kernels alone.
    const bool fast0_aside = tiles0 > 0 && !((c->timing >> VO_T_RESIZE) & 1u);
    if (fast0_aside) {
        VO_HIP(c, hipEventRecord(c->ev_fork, s));
        VO_HIP(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        t_begin_on(c, VO_T_FAST, c->aux);
        launch_fast(c->aux, 0, tiles0, (uint32_t)g.fast_s0, 0, 1);
        t_end_on(c, VO_T_FAST, c->aux);
        VO_HIP(c, hipEventRecord(c->ev_fast0, c->aux));
    }
    t_begin(c, VO_T_RESIZE);
    for (int l = 1; l < kLevels; ++l) {
        uint32_t items = g.L[l].ncg * (uint32_t)((g.L[l].h + RS_ROWS - 1) / RS_ROWS);
        dim3 grid((items + 255) / 256, N);
        hipLaunchKernelGGL(resize_kernel, grid, dim3(256), 0, s, g, l, d_imgs, (uint8_t*)c->pyr.p,
                           (const uint32_t*)c->tab.p);
    }
    t_end(c, VO_T_RESIZE);
    if (tiles0 > 0 && !fast0_aside) {
        t_begin(c, VO_T_FAST);
        launch_fast(s, 0, tiles0, (uint32_t)g.fast_s0, 0, 1);
        t_end(c, VO_T_FAST);
    }
    // The upper levels' launch waits for level 0's: two FAST launches side by side only halve each
    // other's share of the vector ALUs (same step time within 0.3 %), and each launch's event
    // bracket then measures that launch, not the pair.
    if (fast0_aside) VO_HIP(c, hipStreamWaitEvent(s, c->ev_fast0, 0));
    if (tiles_up > 0) {
        t_begin(c, VO_T_FAST);
        launch_fast(s, tiles0, tiles_up, (uint32_t)g.fast_s1, 1, kLevels - 1);
        t_end(c, VO_T_FAST);
    }
    // The blur needs only the pyramid, candidate selection and orientation are mostly latency
    // (LDS sorts, scattered window reads): the blur runs beside them on the auxiliary stream, which
    // is in order behind FAST on level 0; it additionally waits for the main stream (resize chain
    // and FAST on the upper levels) so that FAST keeps the vector ALUs to itself.  Serial, on the
    // main stream, while one of the overlapped groups is bracketed by timing events.
    const unsigned overlapped = (1u << VO_T_SELECT) | (1u << VO_T_BLUR) | (1u << VO_T_DESCRIBE);
    const bool blur_aside = d_desc && fast0_aside && !(c->timing & overlapped);
    auto launch_blur = [&](hipStream_t bs) {
        hipLaunchKernelGGL(blur_kernel, dim3(xcd_grid((uint32_t)g.blur_tiles * N)), dim3(256), 0, bs, g, tabs, d_imgs, pyr,
                           (const uint32_t*)c->tab.p, (uint8_t*)c->blur.p);
    };
    if (blur_aside) {
        VO_HIP(c, hipEventRecord(c->ev_fork, s));
        VO_HIP(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        launch_blur(c->aux);
        VO_HIP(c, hipEventRecord(c->ev_blur, c->aux));
    }
    t_begin(c, VO_T_SELECT);
    HarrisBlocks hb;
    hb.first[0] = 0;
    for (int l = 0; l < kLevels; ++l) {
        const int expect = std::min(2 * g.L[l].quota + 64, std::min(g.sel_cap, g.L[l].cand_cap));
        hb.first[l + 1] = hb.first[l] + std::max((expect + 255) / 256, 1);
    }
#define VO_SELECT(CAPV)                                                                                              \
    do {                                                                                                             \
        hipLaunchKernelGGL(select_a_kernel<CAPV>, dim3(N, kLevels), dim3(512), 0, s, g, (const uint32_t*)c->cand.p,  \
                           (const uint32_t*)c->cand_cnt.p, (uint32_t*)c->sel_keys.p, (uint32_t*)c->sel_cnt.p,        \
                           (uint32_t*)c->flags.p + c->flags_word, c->img_status_cur);                                       \
        hipLaunchKernelGGL(harris_kernel, dim3(xcd_grid((uint32_t)hb.first[kLevels] * N)), dim3(256), 0, s, g, CAPV, hb, d_imgs, \
                           pyr, (const uint32_t*)c->sel_keys.p, (const uint32_t*)c->sel_cnt.p, (float*)c->sel_resp.p); \
        hipLaunchKernelGGL(select_b_kernel<CAPV>, dim3(N, kLevels), dim3(512), 0, s, g,                              \
                           (const uint32_t*)c->sel_keys.p, (const uint32_t*)c->sel_cnt.p,                            \
                           (const float*)c->sel_resp.p, (uint16_t*)c->kp_x.p, (uint16_t*)c->kp_y.p,                  \
                           (float*)c->kp_resp.p, (uint32_t*)c->kp_cnt.p, (uint32_t*)c->flags.p + c->flags_word,                      \
                           c->img_status_cur);                                                              \
    } while (0)
    if (g.sel_cap == 2048) VO_SELECT(2048);
    else if (g.sel_cap == 4096) VO_SELECT(4096);
    else VO_SELECT(8192);
#undef VO_SELECT
    t_end(c, VO_T_SELECT);
    if (d_desc && !blur_aside) {
        t_begin(c, VO_T_BLUR);
        launch_blur(s);
        t_end(c, VO_T_BLUR);
    }
    // the keypoint records and counts written from here on may still be read by the previous
    // batch's ratio test on the side stream
    if (int jrc = join_tail(c)) return jrc;
    t_begin(c, VO_T_DESCRIBE);
    const int dbx = std::max((g.max_features + 31) / 32, 1);   // about four keypoint pairs per wavefront
    hipLaunchKernelGGL(orient_kernel, dim3(xcd_grid((uint32_t)dbx * N)), dim3(256), 0, s, g, dbx, d_imgs, pyr,
                       (const uint16_t*)c->kp_x.p, (const uint16_t*)c->kp_y.p, (const float*)c->kp_resp.p,
                       (const uint32_t*)c->kp_cnt.p, d_kps, cap, d_counts, (uint32_t*)c->flags.p + c->flags_word,
                       c->img_status_cur);
    if (blur_aside) VO_HIP(c, hipStreamWaitEvent(s, c->ev_blur, 0));
    // keypoints per wavefront of the descriptor kernel: sixteen amortise the per-wavefront set-up, eight keep
    // the grid of a small call large enough to fill the GPU (64 per wavefront measured 20 % slower on 1024
    // images of 2000 keypoints, 32 7 %: the kernel is bound by its LDS reads - 512 scattered bytes per
    // keypoint - and long wavefronts only lengthen the tail)
    const int kpw = (long long)N * ((g.max_features + 15) / 16) >= 4096 ? 16 : 8;
    const int ddx = std::max((g.max_features + 4 * kpw - 1) / (4 * kpw), 1);
    if (d_desc)
        hipLaunchKernelGGL(describe_kernel, dim3(xcd_grid((uint32_t)ddx * N)), dim3(256), 0, s, g, ddx, kpw,
                           (const uint8_t*)c->blur.p, (const vo_keypoint*)d_kps, (const int32_t*)d_counts, d_desc, cap);
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
    // (the last row ends after `cols` bytes: a sub-image's buffer may end there)
    const size_t valid = (size_t)(rows - 1) * step + cols;
    int rc;
    if ((rc = ensure(c, c->st_img, bytes))) return rc;
    if ((rc = ensure(c, c->st_kps, (size_t)cap * sizeof(vo_keypoint)))) return rc;
    if ((rc = ensure(c, c->st_desc, (size_t)cap * 32))) return rc;
    if ((rc = ensure(c, c->st_cnt, 64))) return rc;
    VO_HIP(c, hipMemcpyAsync(c->st_img.p, img, valid, hipMemcpyHostToDevice, c->stream));
    if ((rc = orb_prepare(c, 1, rows, cols, step, bytes, max_features))) return rc;
    c->flags_word = 16;   // this call reports from its own image's status word
    rc = orb_run(c, (const uint8_t*)c->st_img.p, (vo_keypoint*)c->st_kps.p, desc_out ? (uint8_t*)c->st_desc.p : nullptr,
                 cap, (int32_t*)c->st_cnt.p);
    c->flags_word = 0;
    if (rc) return rc;
    int32_t n = 0;
    uint32_t status = 0;
    VO_HIP(c, hipMemcpyAsync(&n, c->st_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(&status, c->img_status_cur, 4, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    n = std::min(n, cap);
    if (n > 0) {
        VO_HIP(c, hipMemcpyAsync(kps_out, c->st_kps.p, (size_t)n * sizeof(vo_keypoint), hipMemcpyDeviceToHost,
                                 c->stream));
        if (desc_out)
            VO_HIP(c, hipMemcpyAsync(desc_out, c->st_desc.p, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    // a truncated list is reported, but what was kept is still handed out
    return status ? truncation_error(c, "vo_orb_detect", status) : VO_OK;
}

extern "C" int vo_debug_level(vo_ctx* c, int image, int level, int blurred, uint8_t* out, size_t out_bytes, int* w,
                              int* h) {
    if (!c || !c->geom_valid || level < 0 || level >= kLevels || image < 0 || image >= c->g.n_images)
        return c ? fail(c, VO_ERR_INVALID, "vo_debug_level: bad argument") : VO_ERR_INVALID;
    const Geom& g = c->g;
    const LevelGeom& L = g.L[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!out || out_bytes < (size_t)L.w * L.h)
        return fail(c, VO_ERR_CAPACITY, "vo_debug_level: the level needs " + std::to_string((size_t)L.w * L.h) + " bytes");
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

extern "C" int vo_debug_fast_stats(vo_ctx* c, int enable, uint64_t out[4]) {
    if (!c) return VO_ERR_INVALID;
    VO_HIP(c, hipSetDevice(c->device));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    VO_HIP(c, hipStreamSynchronize(c->aux));
    if (enable) {
        if (int rc = ensure(c, c->fast_stats, 32)) return rc;
        VO_HIP(c, hipMemset(c->fast_stats.p, 0, 32));
        c->fast_stats_on = true;
        return VO_OK;
    }
    if (!out) return fail(c, VO_ERR_INVALID, "vo_debug_fast_stats: out is NULL");
    for (int k = 0; k < 4; ++k) out[k] = 0;
    if (c->fast_stats_on) VO_HIP(c, hipMemcpy(out, c->fast_stats.p, 32, hipMemcpyDeviceToHost));
    c->fast_stats_on = false;
    return VO_OK;
}

extern "C" int vo_debug_fast_thresholds(vo_ctx* c, int image, int32_t out[16], int* raised_out) {
    if (!c || !c->geom_valid || image < 0 || image >= c->g.n_images || !out || !raised_out)
        return c ? fail(c, VO_ERR_INVALID, "vo_debug_fast_thresholds: bad argument") : VO_ERR_INVALID;
    *raised_out = c->fast_raised ? 1 : 0;
    if (!c->fast_raised) return VO_OK;
    VO_HIP(c, hipSetDevice(c->device));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t t[kLevels];
    VO_HIP(c, hipMemcpy(t, (const uint32_t*)c->fast_thr.p + image * kLevels, kLevels * 4, hipMemcpyDeviceToHost));
    const size_t per_group = 1 + (size_t)c->g.n_images * kLevels;
    std::vector<uint32_t> lists(2 * per_group);
    VO_HIP(c, hipMemcpy(lists.data(), c->fast_redo.p, lists.size() * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < kLevels; ++l) {
        // a level sampled whole is never touched by the raised or the redo pass
        const bool whole = c->g.L[l].fast_ns >= c->g.L[l].fast_tx * c->g.L[l].fast_ty;
        out[l] = whole ? kFastThr : (int32_t)t[l];
        out[kLevels + l] = 0;
    }
    for (int group = 0; group < 2; ++group) {
        const uint32_t* list = lists.data() + group * per_group;
        for (uint32_t f = 0; f < std::min<uint32_t>(list[0], (uint32_t)per_group - 1); ++f)
            if ((int)(list[1 + f] / kLevels) == image) out[kLevels + list[1 + f] % kLevels] = 1;
    }
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
