// Brute-force Hamming 2-nearest-neighbour search and Lowe ratio test on
// gfx950.  Stands in for cv::BFMatcher(NORM_HAMMING)::knnMatch(k=2) and the
// ratio loop of ref:src/feature_matcher.cpp:26-39.
//
// Three kernels compute the same table.  The default, knn2_mfma_kernel, runs the distances on the
// matrix cores (bits as fp4 signs, see there) and leaves the vector ALUs the selection of the two
// best.  The other two are integer vector-ALU work: each lane keeps its query descriptor(s) (8
// dwords each) in registers, a train row is the same for every lane of a wavefront, distance =
// 8 x (v_xor, v_bcnt accumulate); knn2_bcast_kernel (one query per lane) broadcasts the row from
// wavefront-private LDS into vector registers, knn2_kernel<QPL> (several queries per lane, train
// sets beyond 65 536 rows with very large grids) fetches it through the scalar cache and uses it
// as the scalar operand.  All keep the running two best as packed keys (distance above train row)
// so that one minimum and one median update them and ties resolve to the lower train row exactly
// as an ascending scan with a strict '<' does.  The wavefronts of a workgroup take alternate runs
// of the train set and merge at the end.
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

#include "vo_internal.h"

namespace vo {

constexpr int MT_ROWS = 256;   // train rows per round of the four wavefronts (64 each)

// One instruction each: the compiler otherwise turns the popcount chain into
// v_bcnt(x, 0) + v_add3 trees and the median into max + min, 23 VALU instructions per
// (query, train row) instead of 19.  Integer VALU issue is what bounds this kernel.
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {   // popcount(x) + acc
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// Two smallest of {a0<=a1} U {b0<=b1} as 64-bit keys (distance << 32 | row).
__device__ __forceinline__ void merge2(unsigned long long& a0, unsigned long long& a1, unsigned long long b0,
                                       unsigned long long b1) {
    unsigned long long lo = a0 < b0 ? a0 : b0;
    unsigned long long hi = a0 < b0 ? b0 : a0;
    unsigned long long s = a0 < b0 ? a1 : b1;   // second of the list that supplied lo
    a0 = lo;
    a1 = hi < s ? hi : s;
}

template <int QPL>
__global__ __launch_bounds__(256) void knn2_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr,
                                                   int cnt_stride, int nq_max, size_t q_stride, const uint8_t* __restrict__ t,
                                                   const int32_t* __restrict__ nt_arr, int nt_max, size_t t_stride,
                                                   int32_t* __restrict__ idx2, int32_t* __restrict__ dist2) {
    __shared__ unsigned long long part[4][64 * QPL][2];

    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = nq_arr ? min(nq_arr[b * cnt_stride], nq_max) : nq_max;
    const int nt = nt_arr ? min(nt_arr[b * cnt_stride], nt_max) : nt_max;
    const int qbase = blockIdx.x * 64 * QPL;
    if (qbase >= nq) return;
    const uint8_t* qb = q + (size_t)b * q_stride;
    const uint8_t* tb = t + (size_t)b * t_stride;

    uint32_t qr[QPL][8];
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        int qi = min(qbase + k * 64 + lane, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(qb + (size_t)qi * 32);
        uint4 a = p[0], c = p[1];
        qr[k][0] = a.x; qr[k][1] = a.y; qr[k][2] = a.z; qr[k][3] = a.w;
        qr[k][4] = c.x; qr[k][5] = c.y; qr[k][6] = c.z; qr[k][7] = c.w;
    }
    unsigned long long best0[QPL], best1[QPL];
#pragma unroll
    for (int k = 0; k < QPL; ++k) best0[k] = best1[k] = ~0ull;

    for (int cbase = 0; cbase < nt; cbase += 65536) {
        const int cn = min(nt - cbase, 65536);
        uint32_t k0[QPL], k1[QPL];
#pragma unroll
        for (int k = 0; k < QPL; ++k) k0[k] = k1[k] = 0xFFFFFFFFu;
        uint32_t thr[QPL];   // distance of the current second best (0xFFFF while there is none)
#pragma unroll
        for (int k = 0; k < QPL; ++k) thr[k] = 0xFFFFu;
        for (int t0 = 0; t0 < cn; t0 += MT_ROWS) {
            // A train row is the same for all 64 lanes: its address is wave-uniform, so the row
            // comes through the scalar cache into SGPRs (s_load_dwordx8) and is an SGPR operand
            // of the xor - no LDS staging, no workgroup barrier in the loop.
            const int r0 = __builtin_amdgcn_readfirstlane(wave) * 64;
            const int rn = min(max(cn - t0 - r0, 0), 64);
            // distance of every query of this lane to one train row
            auto row_dist = [&](const uint4& ta, const uint4& tc, uint32_t* d) {
#pragma unroll
                for (int k = 0; k < QPL; ++k) {
                    uint32_t v = bcnt_acc(qr[k][0] ^ ta.x, 0u);
                    v = bcnt_acc(qr[k][1] ^ ta.y, v);
                    v = bcnt_acc(qr[k][2] ^ ta.z, v);
                    v = bcnt_acc(qr[k][3] ^ ta.w, v);
                    v = bcnt_acc(qr[k][4] ^ tc.x, v);
                    v = bcnt_acc(qr[k][5] ^ tc.y, v);
                    v = bcnt_acc(qr[k][6] ^ tc.z, v);
                    d[k] = bcnt_acc(qr[k][7] ^ tc.w, v);
                }
            };
            auto row_update = [&](const uint32_t* d, uint32_t j) {
#pragma unroll
                for (int k = 0; k < QPL; ++k) {
                    uint32_t key = (d[k] << 16) | j;
                    k1[k] = med3u(k0[k], k1[k], key);
                    k0[k] = min(k0[k], key);
                }
            };
            const uint4* tp = reinterpret_cast<const uint4*>(tb + (size_t)(cbase + t0 + r0) * 32);
            const uint32_t jb = (uint32_t)(t0 + r0);
            int r = 0;
            for (; r + 4 <= rn; r += 4) {   // four rows (two s_load_dwordx16) per step
                uint4 a0 = tp[2 * r], c0 = tp[2 * r + 1], a1 = tp[2 * r + 2], c1 = tp[2 * r + 3];
                uint4 a2 = tp[2 * r + 4], c2 = tp[2 * r + 5], a3 = tp[2 * r + 6], c3 = tp[2 * r + 7];
                uint32_t d0[QPL], d1[QPL], d2[QPL], d3[QPL];
                row_dist(a0, c0, d0);
                row_dist(a1, c1, d1);
                row_dist(a2, c2, d2);
                row_dist(a3, c3, d3);
                // Rows come in ascending index order, so a row enters a lane's top two only with a
                // distance strictly below the current second best; once the lists have settled
                // most row groups are skipped by the whole wavefront with four compares.
                bool better = false;
#pragma unroll
                for (int k = 0; k < QPL; ++k)
                    better |= (d0[k] < thr[k]) | (d1[k] < thr[k]) | (d2[k] < thr[k]) | (d3[k] < thr[k]);
                if (__any(better)) {
                    row_update(d0, jb + r);
                    row_update(d1, jb + r + 1);
                    row_update(d2, jb + r + 2);
                    row_update(d3, jb + r + 3);
#pragma unroll
                    for (int k = 0; k < QPL; ++k) thr[k] = k1[k] >> 16;
                }
            }
            for (; r < rn; ++r) {
                uint32_t d[QPL];
                row_dist(tp[2 * r], tp[2 * r + 1], d);
                row_update(d, jb + r);
            }
#pragma unroll
            for (int k = 0; k < QPL; ++k) thr[k] = k1[k] >> 16;
        }
#pragma unroll
        for (int k = 0; k < QPL; ++k) {
            unsigned long long b0 = k0[k] == 0xFFFFFFFFu
                                        ? ~0ull
                                        : ((unsigned long long)(k0[k] >> 16) << 32) | (uint32_t)(cbase + (k0[k] & 0xFFFF));
            unsigned long long b1 = k1[k] == 0xFFFFFFFFu
                                        ? ~0ull
                                        : ((unsigned long long)(k1[k] >> 16) << 32) | (uint32_t)(cbase + (k1[k] & 0xFFFF));
            merge2(best0[k], best1[k], b0, b1);
        }
    }
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
        part[wave][k * 64 + lane][0] = best0[k];
        part[wave][k * 64 + lane][1] = best1[k];
    }
    __syncthreads();
    for (int i = tid; i < 64 * QPL; i += 256) {
        int qi = qbase + i;
        if (qi >= nq) continue;
        unsigned long long a0 = part[0][i][0], a1 = part[0][i][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) merge2(a0, a1, part[w][i][0], part[w][i][1]);
        size_t o = ((size_t)b * nq_max + qi) * 2;
        idx2[o] = a0 == ~0ull ? -1 : (int32_t)(uint32_t)a0;
        idx2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(uint32_t)a1;
        dist2[o] = a0 == ~0ull ? -1 : (int32_t)(a0 >> 32);
        dist2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(a1 >> 32);
    }
}

// The same search with the train rows in vector registers.  A v_xor with an SGPR operand
// issues at the slow rate (about 4.2 cycles per wave64 instruction on gfx950, like v_bcnt),
// one between vector registers in a run of its kind at about 2.6: each wavefront copies its
// 64-row block into a private 2 KB LDS region (one coalesced row per lane) and reads the rows
// back with uniform-address ds_read_b128 (broadcast, about 4.5 cycles per instruction and CU);
// the 32 xors of four rows are issued as one run ahead of the popcount chains.  The LDS region
// is private to the wavefront and LDS operations of one wavefront execute in order, so the
// loop needs no workgroup barrier; the next block's global loads are in flight during the
// current block's arithmetic.
template <int NW>   // wavefronts per workgroup: they split the train rows and merge at the end
__global__ __launch_bounds__(64 * NW) void knn2_bcast_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr,
                                                         int cnt_stride, int nq_max, size_t q_stride,
                                                         const uint8_t* __restrict__ t, const int32_t* __restrict__ nt_arr,
                                                         int nt_max, size_t t_stride, int32_t* __restrict__ idx2,
                                                         int32_t* __restrict__ dist2, int rows_per_z,
                                                         unsigned long long* __restrict__ part_out) {
    __shared__ uint4 rows[NW][128];   // per wavefront: 64 rows of two uint4
    __shared__ unsigned long long part[NW][64][2];

    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = nq_arr ? min(nq_arr[b * cnt_stride], nq_max) : nq_max;
    const int nt = nt_arr ? min(nt_arr[b * cnt_stride], nt_max) : nt_max;
    const int qbase = blockIdx.x * 64;
    if (qbase >= nq) return;
    const uint8_t* qb = q + (size_t)b * q_stride;
    const uint8_t* tb = t + (size_t)b * t_stride;

    uint32_t qr[8];
    {
        const int qi = min(qbase + lane, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(qb + (size_t)qi * 32);
        const uint4 a = p[0], c = p[1];
        qr[0] = a.x; qr[1] = a.y; qr[2] = a.z; qr[3] = a.w;
        qr[4] = c.x; qr[5] = c.y; qr[6] = c.z; qr[7] = c.w;
    }
    unsigned long long best0 = ~0ull, best1 = ~0ull;
    uint4* my_rows = rows[wave];
    // blockIdx.z: this workgroup's chunk of the train set (one chunk, rows_per_z = INT_MAX, unless
    // the search was split; see knn2_merge_kernel)
    const int t_lo = min((int)blockIdx.z * (long long)rows_per_z, (long long)nt);
    const int t_hi = (int)min((long long)nt, (long long)t_lo + rows_per_z);

    for (int cbase = t_lo; cbase < t_hi; cbase += 65536) {
        const int cn = min(t_hi - cbase, 65536);
        uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, thr = 0xFFFFu;
        const uint4* tcb = reinterpret_cast<const uint4*>(tb + (size_t)cbase * 32);
        auto fetch = [&](int first, uint4* a, uint4* c) {   // row first + lane of the chunk (clamped)
            const int row = min(first + lane, cn - 1);
            *a = tcb[2 * row];
            *c = tcb[2 * row + 1];
        };
        auto update = [&](uint32_t d, uint32_t j) {
            const uint32_t key = (d << 16) | j;
            k1 = med3u(k0, k1, key);
            k0 = min(k0, key);
        };
        uint4 na, nc;
        int first = wave * 64;   // this wavefront takes every NW-th 64-row block
        if (first < cn) fetch(first, &na, &nc);
        for (; first < cn; first += 64 * NW) {
            my_rows[2 * lane] = na;
            my_rows[2 * lane + 1] = nc;
            if (first + 64 * NW < cn) fetch(first + 64 * NW, &na, &nc);
            const int rn = min(cn - first, 64);
            int r = 0;
            for (; r + 4 <= rn; r += 4) {
                uint32_t x[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 ra = my_rows[2 * (r + i)], rc = my_rows[2 * (r + i) + 1];
                    x[i][0] = ra.x; x[i][1] = ra.y; x[i][2] = ra.z; x[i][3] = ra.w;
                    x[i][4] = rc.x; x[i][5] = rc.y; x[i][6] = rc.z; x[i][7] = rc.w;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[i][k] ^= qr[k];
                __builtin_amdgcn_sched_barrier(0);
                uint32_t d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t v = bcnt_acc(x[i][0], 0u);
#pragma unroll
                    for (int k = 1; k < 8; ++k) v = bcnt_acc(x[i][k], v);
                    d[i] = v;
                }
                const bool better = (d[0] < thr) | (d[1] < thr) | (d[2] < thr) | (d[3] < thr);
                if (__any(better)) {
                    const uint32_t j = (uint32_t)(first + r);
                    update(d[0], j);
                    update(d[1], j + 1);
                    update(d[2], j + 2);
                    update(d[3], j + 3);
                    thr = k1 >> 16;
                }
            }
            for (; r < rn; ++r) {
                const uint4 ra = my_rows[2 * r], rc = my_rows[2 * r + 1];
                uint32_t v = bcnt_acc(qr[0] ^ ra.x, 0u);
                v = bcnt_acc(qr[1] ^ ra.y, v);
                v = bcnt_acc(qr[2] ^ ra.z, v);
                v = bcnt_acc(qr[3] ^ ra.w, v);
                v = bcnt_acc(qr[4] ^ rc.x, v);
                v = bcnt_acc(qr[5] ^ rc.y, v);
                v = bcnt_acc(qr[6] ^ rc.z, v);
                v = bcnt_acc(qr[7] ^ rc.w, v);
                update(v, (uint32_t)(first + r));
            }
            thr = k1 >> 16;
        }
        const unsigned long long b0 =
            k0 == 0xFFFFFFFFu ? ~0ull : ((unsigned long long)(k0 >> 16) << 32) | (uint32_t)(cbase + (k0 & 0xFFFF));
        const unsigned long long b1 =
            k1 == 0xFFFFFFFFu ? ~0ull : ((unsigned long long)(k1 >> 16) << 32) | (uint32_t)(cbase + (k1 & 0xFFFF));
        merge2(best0, best1, b0, b1);
    }
    part[wave][lane][0] = best0;
    part[wave][lane][1] = best1;
    __syncthreads();
    if (tid < 64) {
        const int qi = qbase + tid;
        if (qi < nq) {
            unsigned long long a0 = part[0][tid][0], a1 = part[0][tid][1];
#pragma unroll
            for (int w = 1; w < NW; ++w) merge2(a0, a1, part[w][tid][0], part[w][tid][1]);
            if (part_out) {   // split search: the chunk's two best as keys, merged by knn2_merge_kernel
                const size_t po = (((size_t)b * gridDim.z + blockIdx.z) * nq_max + qi) * 2;
                part_out[po] = a0;
                part_out[po + 1] = a1;
                return;
            }
            const size_t o = ((size_t)b * nq_max + qi) * 2;
            idx2[o] = a0 == ~0ull ? -1 : (int32_t)(uint32_t)a0;
            idx2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(uint32_t)a1;
            dist2[o] = a0 == ~0ull ? -1 : (int32_t)(a0 >> 32);
            dist2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(a1 >> 32);
        }
    }
}

// The same search on the matrix cores.  With every bit written as +-1 the Hamming distance of two
// 256-bit rows is (256 - dot) / 2, a 256-deep dot product, and gfx950 multiplies 4-bit floats
// (E2M1: sign and magnitude, +-2 and +-4 are exact codes) 32 x 32 x 64 at a time with fp32
// accumulation: a 32 x 32 tile of distances is four v_mfma_scale_f32_32x32x64_f8f6f4, about 160
// cycles on a SIMD's matrix core, against the 16 000 vector cycles of 16 xor/popcount pairs for
// each of its 1024 entries.  A bit becomes the sign of a nibble with three shifts and four and-ors
// per dword (the order of the 256 positions does not matter as long as both sides use the same
// one); train values are +-2, query values -+4 times the block scale 2^11, so a product is
// -16384 where the bits agree and +16384 where they differ and the tile comes out as
// 32768 d - 2^22, to which the accumulator's start value adds 2^22 + r: every result is the
// packed key 32768 d + r itself, an integer below 2^24 and therefore exact in fp32, r being the
// train row in a frame that moves with the tile (the running keys are aged by the rows passed).
// What is left for the vector ALUs is one v_min_f32 and one v_med3_f32 per key.
// A wavefront keeps two groups of 32 queries (the B operands, 32 registers) and streams the train
// set in 32-row tiles straight from global memory, a 16-byte load per lane and tile.  C layout of
// the instruction: lane l holds column l & 31 (the query), rows (i & 3) + 8 (i >> 2) + 4 (l >> 5)
// in register i; operand lane l supplies row / column l & 31, positions 128 (l >> 5) .. + 127.
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef float v16f32 __attribute__((ext_vector_type(16)));
constexpr int MF_FRAME = 32704;        // r of the current tile's row 0; rows behind it count down
constexpr int MF_CHUNK = 30720;        // train rows per chunk of one frame (a multiple of 32 * 16)
constexpr float MF_NONE = 3.0e7f;      // above every key (2^24 and up: no row)
constexpr int MF_LATE = 96;            // tiles of a chunk after which a wavefront tests before it updates
constexpr int MF_SCALE_ONE = 0x7F7F7F7F, MF_SCALE_2K = 0x8A8A8A8A;   // E8M0 block scales 2^0 and 2^11

// 32 bits -> 32 fp4 values of one magnitude: nibble i of dword j has bit 4 i + j as its sign
// (one v_and_or_b32 per output: the mask and the magnitude are passed in registers, made opaque
// by the caller, because a gfx9 three-operand instruction cannot carry a 32-bit literal and the
// compiler otherwise splits every output into an and and an or)
__device__ __forceinline__ v8i32 fp4_signs(uint32_t x, uint32_t sign_mask, uint32_t mag) {
    v8i32 o = {(int)(((x << 3) & sign_mask) | mag), (int)(((x << 2) & sign_mask) | mag),
               (int)(((x << 1) & sign_mask) | mag), (int)((x & sign_mask) | mag), 0, 0, 0, 0};
    return o;
}
// (compiler builtins, not inline assembly: the wait states between a matrix instruction and the
// first read of its result are put in by the compiler, which does not look inside an asm block)
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t ab = a < b ? a : b;
    return ab < c ? ab : c;
}
// (the minimum of two positive floats through their bit patterns: fminf brings a canonicalising
// v_max_f32 with it)
__device__ __forceinline__ float min2f(float a, float b) {
    return __builtin_bit_cast(float, min(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b)));
}

// NW: wavefronts per workgroup (they split the train rows and merge at the end); G: groups of 32
// queries per wavefront (the expansion of a train tile is shared by a wavefront's groups)
template <int NW, int G>
__global__ __launch_bounds__(64 * NW, G <= 2 && NW <= 8 ? 4 : 2) void knn2_mfma_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr,
                                                        int cnt_stride, int nq_max, size_t q_stride,
                                                        const uint8_t* __restrict__ t, const int32_t* __restrict__ nt_arr,
                                                        int nt_max, size_t t_stride, int32_t* __restrict__ idx2,
                                                        int32_t* __restrict__ dist2, int rows_per_z,
                                                        unsigned long long* __restrict__ part_out) {
    constexpr int QW = 32 * G;   // queries per wavefront and workgroup
    __shared__ unsigned long long part[NW][QW][2];
    __shared__ unsigned long long lane_best[64 * NW][G][2];   // [lane][query group][first, second]

    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the loop's row counters follow from it)
    const int col = lane & 31, half = lane >> 5;
    const int nq = nq_arr ? min(nq_arr[b * cnt_stride], nq_max) : nq_max;
    const int nt = nt_arr ? min(nt_arr[b * cnt_stride], nt_max) : nt_max;
    const int qbase = blockIdx.x * QW;
    if (qbase >= nq) return;
    const uint8_t* qb = q + (size_t)b * q_stride;
    const uint8_t* tb = t + (size_t)b * t_stride;

    // the queries: group g holds rows qbase + 32 g + col; -+4 (the flipped sign makes agreeing bits
    // the negative product)
    uint32_t sign_mask = 0x88888888u, mag_q = 0x66666666u, mag_t = 0x44444444u;
    asm volatile("" : "+s"(sign_mask), "+v"(mag_q), "+v"(mag_t));   // (no instruction: keeps them out of constant folding)
    v8i32 Bq[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int qi = min(qbase + 32 * g + col, nq - 1);
        const uint4 raw = *reinterpret_cast<const uint4*>(qb + (size_t)qi * 32 + 16 * half);
        Bq[g][0] = fp4_signs(~raw.x, sign_mask, mag_q);
        Bq[g][1] = fp4_signs(~raw.y, sign_mask, mag_q);
        Bq[g][2] = fp4_signs(~raw.z, sign_mask, mag_q);
        Bq[g][3] = fp4_signs(~raw.w, sign_mask, mag_q);
    }
    // accumulator start values: 2^22 + the row's r in the current frame
    v16f32 c_in;
#pragma unroll
    for (int i = 0; i < 16; ++i) c_in[i] = 4194304.0f + (float)(MF_FRAME + (i & 3) + 8 * (i >> 2) + 4 * half);

    // (the two best over the chunks done so far wait in LDS, not in eight registers of the loop)
    unsigned long long (*best)[2] = lane_best[tid];
#pragma unroll
    for (int g = 0; g < G; ++g) best[g][0] = best[g][1] = ~0ull;
    const int t_lo = min((int)blockIdx.z * (long long)rows_per_z, (long long)nt);
    const int t_hi = (int)min((long long)nt, (long long)t_lo + rows_per_z);

    for (int cbase = t_lo; cbase < t_hi; cbase += MF_CHUNK) {
        const int cn = min(t_hi - cbase, MF_CHUNK);
        float k0[G], k1[G];
#pragma unroll
        for (int g = 0; g < G; ++g) k0[g] = k1[g] = MF_NONE;
        const uint8_t* tcb = tb + (size_t)cbase * 32;
        auto fetch = [&](int first) {   // row first + col (clamped), this lane's half of it
            return *reinterpret_cast<const uint4*>(tcb + (size_t)min(first + col, cn - 1) * 32 + 16 * half);
        };
        auto tile = [&](const uint4& raw, int rows_left, bool late) {   // one 32-row tile against the query groups
            const v8i32 A0 = fp4_signs(raw.x, sign_mask, mag_t), A1 = fp4_signs(raw.y, sign_mask, mag_t);
            const v8i32 A2 = fp4_signs(raw.z, sign_mask, mag_t), A3 = fp4_signs(raw.w, sign_mask, mag_t);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v16f32 acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, Bq[g][0], c_in, 4, 4, 0, MF_SCALE_ONE, 0, MF_SCALE_2K);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, Bq[g][1], acc, 4, 4, 0, MF_SCALE_ONE, 0, MF_SCALE_2K);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, Bq[g][2], acc, 4, 4, 0, MF_SCALE_ONE, 0, MF_SCALE_2K);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A3, Bq[g][3], acc, 4, 4, 0, MF_SCALE_ONE, 0, MF_SCALE_2K);
                if (rows_left < 32) {   // wave-uniform: rows past the end of the set get keys above all others
                    const int lim = rows_left - 4 * half;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((i & 3) + 8 * (i >> 2) >= lim) acc[i] = MF_NONE;
                }
                // Late in a long scan hardly any tile holds a row that beats a query's second best:
                // the sixteen keys are first reduced to their minimum (eight v_min3_u32) and the
                // thirty-two updates skipped unless some lane's minimum gets in.  Early on nearly
                // every tile changes some lane's pair and the test would only add work.
                bool update = true;
                if (late) {
                    uint32_t u[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) u[i] = __float_as_uint(acc[i]);
                    const uint32_t m = umin3(umin3(umin3(u[0], u[1], u[2]), umin3(u[3], u[4], u[5]), umin3(u[6], u[7], u[8])),
                                             umin3(u[9], u[10], u[11]), umin3(umin3(u[12], u[13], u[14]), u[15], u[15]));
                    update = __ballot(m < __float_as_uint(k1[g])) != 0ull;
                }
                if (update) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        k1[g] = med3f(k0[g], k1[g], acc[i]);
                        k0[g] = min2f(k0[g], acc[i]);
                    }
                }
            }
        };
        auto age = [&](float rows) {   // the frame moves on by `rows`: rows seen so far fall behind
#pragma unroll
            for (int g = 0; g < G; ++g) {
                k0[g] -= rows;
                k1[g] -= rows;
            }
        };
        uint4 next;
        int first = wave * 32, frame = first;   // this wavefront takes every NW-th 32-row tile
        if (first < cn) next = fetch(first);
        for (int seen = 0; first < cn; first += 32 * NW, ++seen) {
            const uint4 raw = next;
            if (first + 32 * NW < cn) next = fetch(first + 32 * NW);
            age((float)(first - frame));   // (keys that are still MF_NONE stay above 2^24)
            frame = first;
            tile(raw, cn - first, seen >= MF_LATE);
        }
        // keys -> (distance << 32 | row); the two halves of the wavefront saw different rows
#pragma unroll
        for (int g = 0; g < G; ++g) {
            unsigned long long e[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float key = k == 0 ? k0[g] : k1[g];
                const uint32_t ki = (uint32_t)key;
                const int row = cbase + frame + (int)(ki & 32767u) - MF_FRAME;
                e[k] = key >= 16777216.0f ? ~0ull : ((unsigned long long)(ki >> 15) << 32) | (uint32_t)row;
            }
            unsigned long long b0 = best[g][0], b1 = best[g][1];
            merge2(b0, b1, e[0], e[1]);
            best[g][0] = b0;
            best[g][1] = b1;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        unsigned long long b0 = best[g][0], b1 = best[g][1];
        const unsigned long long o0 = __shfl_xor(b0, 32), o1 = __shfl_xor(b1, 32);
        merge2(b0, b1, o0, o1);
        if (half == 0) {
            part[wave][32 * g + col][0] = b0;
            part[wave][32 * g + col][1] = b1;
        }
    }
    __syncthreads();
    for (int qo = tid; qo < QW; qo += 64 * NW) {
        const int qi = qbase + qo;
        if (qi < nq) {
            unsigned long long a0 = part[0][qo][0], a1 = part[0][qo][1];
#pragma unroll
            for (int w = 1; w < NW; ++w) merge2(a0, a1, part[w][qo][0], part[w][qo][1]);
            if (part_out) {   // split search: the chunk's two best as keys, merged by knn2_merge_kernel
                const size_t po = (((size_t)b * gridDim.z + blockIdx.z) * nq_max + qi) * 2;
                part_out[po] = a0;
                part_out[po + 1] = a1;
                continue;
            }
            const size_t o = ((size_t)b * nq_max + qi) * 2;
            idx2[o] = a0 == ~0ull ? -1 : (int32_t)(uint32_t)a0;
            idx2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(uint32_t)a1;
            dist2[o] = a0 == ~0ull ? -1 : (int32_t)(a0 >> 32);
            dist2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(a1 >> 32);
        }
    }
}

// A search with few query rows (one shard of a large match: 8 192 rows are 128 workgroups on 256
// compute units) is split over the train set as well: grid.z chunks of rows_per_z train rows, each
// workgroup leaving its chunk's two best per query as keys (distance << 32 | train row).  The keys
// order by distance, then by row, which is what an ascending scan with a strict '<' keeps: the two
// smallest keys over all chunks are the result of the unsplit search, ties included.
__global__ __launch_bounds__(256) void knn2_merge_kernel(const unsigned long long* __restrict__ part,
                                                         const int32_t* __restrict__ nq_arr, int cnt_stride, int nq_max,
                                                         int split, int32_t* __restrict__ idx2,
                                                         int32_t* __restrict__ dist2) {
    const int b = blockIdx.y, qi = blockIdx.x * 256 + threadIdx.x;
    const int nq = nq_arr ? min(nq_arr[b * cnt_stride], nq_max) : nq_max;
    if (qi >= nq) return;
    unsigned long long a0 = ~0ull, a1 = ~0ull;
    for (int z = 0; z < split; ++z) {
        const size_t po = (((size_t)b * split + z) * nq_max + qi) * 2;
        merge2(a0, a1, part[po], part[po + 1]);
    }
    const size_t o = ((size_t)b * nq_max + qi) * 2;
    idx2[o] = a0 == ~0ull ? -1 : (int32_t)(uint32_t)a0;
    idx2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(uint32_t)a1;
    dist2[o] = a0 == ~0ull ? -1 : (int32_t)(a0 >> 32);
    dist2[o + 1] = a1 == ~0ull ? -1 : (int32_t)(a1 >> 32);
}

// Lowe ratio test and ordered compaction (query order), one workgroup per
// pair.  Also gathers the matched keypoint coordinates.
__global__ __launch_bounds__(256) void ratio_kernel(const int32_t* __restrict__ idx2, const int32_t* __restrict__ dist2,
                                                    const int32_t* __restrict__ nq_arr, int cnt_stride, int nq_max,
                                                    float ratio,
                                                    const vo_keypoint* __restrict__ kp1,
                                                    const vo_keypoint* __restrict__ kp2, size_t kp_stride,
                                                    int32_t* __restrict__ qidx, int32_t* __restrict__ tidx,
                                                    float* __restrict__ dist, float* __restrict__ p1,
                                                    float* __restrict__ p2, int32_t* __restrict__ nmatch) {
    __shared__ int wsum[4];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = nq_arr ? min(nq_arr[b * cnt_stride], nq_max) : nq_max;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 256) {
        int qi = q0 + tid;
        bool keep = false;
        int i0 = -1;
        float d0 = 0.f;
        if (qi < nq) {
            size_t o = ((size_t)b * nq_max + qi) * 2;
            i0 = idx2[o];
            int i1 = idx2[o + 1];
            if (i0 >= 0 && i1 >= 0) {
                d0 = (float)dist2[o];
                float d1 = (float)dist2[o + 1];
                keep = d0 < ratio * d1;
            }
        }
        unsigned long long m = __ballot(keep);
        int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            size_t o = (size_t)b * nq_max + off + before;
            qidx[o] = qi;
            tidx[o] = i0;
            dist[o] = d0;
            if (kp1 && kp2) {
                const vo_keypoint& a = kp1[(size_t)b * kp_stride + qi];
                const vo_keypoint& c = kp2[(size_t)b * kp_stride + i0];
                p1[o * 2] = a.x;
                p1[o * 2 + 1] = a.y;
                p2[o * 2] = c.x;
                p2[o * 2 + 1] = c.y;
            }
        }
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) nmatch[b] = base;
}

int knn2_launch(vo_ctx* c, const uint8_t* d_q, const int32_t* d_nq, int cnt_stride, int nq_max, size_t q_stride, const uint8_t* d_t,
                const int32_t* d_nt, int nt_max, size_t t_stride, int batch, int32_t* d_idx2, int32_t* d_dist2,
                int nq_typical, hipStream_t on) {
    if (nq_max <= 0 || batch <= 0) return VO_OK;
    // The table below may still be read by the previous batch's ratio test: the main stream waits for it.  (On a
    // tail stream the search follows that test in stream order - table and stream belong to the same tail set.)
    if (!on)
        if (int jrc = join_tail(c)) return jrc;
    // The kernel: the matrix-core one unless VO_OPT_KNN_MODE says otherwise (0: the scalar-operand
    // vector kernel, 1: the vector kernel with the train rows broadcast from LDS; tuning and test aids).
    const int mode = c->opt[VO_OPT_KNN_MODE] < 0 ? 2 : c->opt[VO_OPT_KNN_MODE];
    // Queries per lane of the vector kernels.  One query per lane measures fastest on everything up to
    // 65 536 x 65 536 (a train set of 2 MB stays in the L2 caches, whatever the batch); more
    // queries per lane only pay once the train set is larger and the grid so large that its
    // re-reads (one pass per workgroup) start to matter.
    int qpl = 1;
    if (nt_max > 65536 && mode < 2) {
        qpl = 4;
        while (qpl > 1 && (long long)((nq_max + 64 * qpl - 1) / (64 * qpl)) * batch < 16384) qpl >>= 1;
    }
    if (c->opt[VO_OPT_KNN_QUERIES_PER_LANE] > 0) qpl = c->opt[VO_OPT_KNN_QUERIES_PER_LANE];   // tuning aid
    dim3 grid((nq_max + 64 * qpl - 1) / (64 * qpl), batch);
    hipStream_t s = on ? on : c->stream;
    t_begin_on(c, VO_T_KNN2, s);
    if (qpl == 1 && mode >= 1) {
        // few workgroups (one large match): more wavefronts per workgroup, so that every SIMD
        // still has several to switch between.  Many (a batch of some hundred pairs): one
        // wavefront per workgroup - it then sees every train row of its queries, improves on its
        // two best about a third as often as four wavefronts with a quarter of the rows each
        // (2 ln n events per query against 4 x 2 ln(n/4)), and nothing is left to merge; 7 % less
        // kernel time at 512 pairs, but only with 12 or more wavefronts per SIMD to draw from.
        const int nq_live = nq_typical > 0 ? std::min(nq_typical, nq_max) : nq_max;
        // The matrix-core kernel with four groups of 32 queries per wavefront instead of two (the
        // expansion of a train tile into fp4 signs is shared by a wavefront's groups): pays on long
        // scans (65 536 x 65 536: -13 %), loses on the pipeline's per-pair searches (2 000 x 2 000 in
        // batches of 512: 0.53 against 0.33 ms - a third fewer wavefronts per SIMD, 128-query blocks
        // that the last wavefront of a pair fills badly).  Mode 3 forces it (test aid).
        const int groups = mode >= 3 || (mode == 2 && !d_nt && nt_max >= 16384 && nq_max >= 4096) ? 4 : 2;
        if (mode >= 2) grid.x = (nq_max + 32 * groups - 1) / (32 * groups);
        const long long wgs = (long long)grid.x * grid.y;
        const long long live = (long long)((nq_live + 32 * groups - 1) / (32 * groups)) * batch * (groups / 2);
        int nw = live >= 12 * 1024 ? 1 : wgs * 4 >= 8 * 1024 ? 4 : wgs * 8 >= 8 * 1024 ? 8 : 16;
        // Fewer than four workgroups per compute unit even so (a shard of a large match): the train
        // set is split as well, into chunks of whole 64-row blocks with at least 128 rows for each
        // wavefront of a workgroup, and a second kernel merges the chunks' results.
        int split = 1;
        // (only where the train count is known on the host and large: the pipeline's per-pair
        // searches pass capacities and are batched instead)
        if (wgs < 1024 && !d_nt && nt_max >= 8192) {
            nw = groups == 4 ? 4 : 8;
            split = (int)std::min<long long>({64, (1024 + wgs - 1) / wgs, nt_max / 1024});
        }
        if (c->opt[VO_OPT_KNN_WAVES] > 0) nw = c->opt[VO_OPT_KNN_WAVES];   // tuning aids
        if (groups == 4 && nw > 8) nw = 8;
        if (c->opt[VO_OPT_KNN_TRAIN_SPLIT] > 0) split = c->opt[VO_OPT_KNN_TRAIN_SPLIT];
        split = std::max(1, std::min(split, (nt_max + 63) / 64));
        int rows_per_z = 0x7FFFFFFF;
        unsigned long long* part = nullptr;
        if (split > 1) {
            rows_per_z = (((nt_max + split - 1) / split) + 63) & ~63;
            split = (nt_max + rows_per_z - 1) / rows_per_z;
        }
        if (split > 1) {
            if (int rc = ensure(c, c->k_part, (size_t)batch * split * nq_max * 16)) return rc;
            part = (unsigned long long*)c->k_part.p;
            grid.z = split;
        }
#define VO_KNN_LAUNCH(KERNEL, NW)                                                                                      \
    hipLaunchKernelGGL(KERNEL<NW>, grid, dim3(64 * NW), 0, s, d_q, d_nq, cnt_stride, nq_max, q_stride, d_t, d_nt, nt_max, \
                       t_stride, d_idx2, d_dist2, rows_per_z, part)
        if (mode >= 2) {   // the matrix-core kernel
#define VO_MFMA_LAUNCH(NW, G)                                                                                            \
    hipLaunchKernelGGL((knn2_mfma_kernel<NW, G>), grid, dim3(64 * NW), 0, s, d_q, d_nq, cnt_stride, nq_max, q_stride, d_t, \
                       d_nt, nt_max, t_stride, d_idx2, d_dist2, rows_per_z, part)
            if (groups == 4) {
                switch (nw) {
                    case 1: VO_MFMA_LAUNCH(1, 4); break;
                    case 4: VO_MFMA_LAUNCH(4, 4); break;
                    default: VO_MFMA_LAUNCH(8, 4); break;
                }
            } else {
                switch (nw) {
                    case 1: VO_MFMA_LAUNCH(1, 2); break;
                    case 4: VO_MFMA_LAUNCH(4, 2); break;
                    case 8: VO_MFMA_LAUNCH(8, 2); break;
                    default: VO_MFMA_LAUNCH(16, 2); break;
                }
            }
#undef VO_MFMA_LAUNCH
        } else {
            switch (nw) {
                case 1: VO_KNN_LAUNCH(knn2_bcast_kernel, 1); break;
                case 4: VO_KNN_LAUNCH(knn2_bcast_kernel, 4); break;
                case 8: VO_KNN_LAUNCH(knn2_bcast_kernel, 8); break;
                default: VO_KNN_LAUNCH(knn2_bcast_kernel, 16); break;
            }
        }
#undef VO_KNN_LAUNCH
        if (split > 1)
            hipLaunchKernelGGL(knn2_merge_kernel, dim3((nq_max + 255) / 256, batch), dim3(256), 0, s,
                               (const unsigned long long*)part, d_nq, cnt_stride, nq_max, split, d_idx2, d_dist2);
    } else if (qpl == 4)
        hipLaunchKernelGGL(knn2_kernel<4>, grid, dim3(256), 0, s, d_q, d_nq, cnt_stride, nq_max, q_stride, d_t, d_nt, nt_max,
                           t_stride, d_idx2, d_dist2);
    else if (qpl == 2)
        hipLaunchKernelGGL(knn2_kernel<2>, grid, dim3(256), 0, s, d_q, d_nq, cnt_stride, nq_max, q_stride, d_t, d_nt, nt_max,
                           t_stride, d_idx2, d_dist2);
    else
        hipLaunchKernelGGL(knn2_kernel<1>, grid, dim3(256), 0, s, d_q, d_nq, cnt_stride, nq_max, q_stride, d_t, d_nt, nt_max,
                           t_stride, d_idx2, d_dist2);
    t_end_on(c, VO_T_KNN2, s);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

int ratio_launch(vo_ctx* c, const int32_t* d_idx2, const int32_t* d_dist2, const int32_t* d_nq, int cnt_stride,
                 int nq_max, int batch,
                 float ratio, const vo_keypoint* d_kp1, const vo_keypoint* d_kp2, size_t kp_stride, int32_t* d_qidx,
                 int32_t* d_tidx, float* d_dist, float* d_p1, float* d_p2, int32_t* d_nmatch, hipStream_t s) {
    if (batch <= 0) return VO_OK;
    t_begin_on(c, VO_T_RATIO, s);
    hipLaunchKernelGGL(ratio_kernel, dim3(batch), dim3(256), 0, s, d_idx2, d_dist2, d_nq, cnt_stride, nq_max, ratio, d_kp1,
                       d_kp2, kp_stride, d_qidx, d_tidx, d_dist, d_p1, d_p2, d_nmatch);
    t_end_on(c, VO_T_RATIO, s);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

// ------------------------------------------------------------------ static workspace
// Workspace of vo_match_ratio_static: the kNN(2) table (nearest two train rows and their
// distances per query) that the search kernel writes and the ratio kernel reads.  It lives in
// the code object instead of in hipMalloc memory, so that the entry point allocates nothing.
// Every GPU the library runs on holds its own copy (512 KiB) until the process ends.
constexpr int kStaticMaxQueries = VO_STATIC_MAX_QUERIES;
static __device__ __align__(256) int32_t scratch[kStaticMaxQueries * 4];   // idx2 [n][2], then dist2 [n][2]

// Device address of `scratch` on each GPU.  In host code the array's name is only a handle for
// the runtime, not an address a kernel can use: the address is asked for once per device
// (vo_ctx_create, with that device current) and kernels receive the pointer kept here.
constexpr int kMaxDevices = 64;
static int32_t* g_scratch_ptr[kMaxDevices] = {nullptr};
static std::mutex g_scratch_mutex;

int static_scratch_init(vo_ctx* c) {
    if (c->device < 0 || c->device >= kMaxDevices)
        return fail(c, VO_ERR_INVALID, "static workspace: device index beyond " + std::to_string(kMaxDevices - 1));
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[c->device]) return VO_OK;
    void* p = nullptr;
    VO_HIP(c, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(c, VO_ERR_HIP, "static workspace: hipGetSymbolAddress returned a null address");
    g_scratch_ptr[c->device] = (int32_t*)p;
    return VO_OK;
}

}  // namespace vo

using namespace vo;

extern "C" int vo_match_ratio_static(vo_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, float ratio,
                                     int32_t* d_query_idx, int32_t* d_train_idx, float* d_distance,
                                     int32_t* d_n_out) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || !d_n_out || (nq > 0 && (!d_q || !d_query_idx || !d_train_idx || !d_distance)) ||
        (nt > 0 && !d_t))
        return fail(c, VO_ERR_INVALID, "vo_match_ratio_static: bad argument");
    if (((uintptr_t)d_q | (uintptr_t)d_t) & 15)
        return fail(c, VO_ERR_INVALID, "vo_match_ratio_static: descriptor arrays must be 16-byte aligned");
    if (nq > kStaticMaxQueries)
        return fail(c, VO_ERR_CAPACITY, "vo_match_ratio_static: " + std::to_string(nq) + " query rows exceed the static workspace of " +
                                            std::to_string(kStaticMaxQueries) + " (VO_STATIC_MAX_QUERIES)");
    int32_t* ws = (c->device >= 0 && c->device < kMaxDevices) ? g_scratch_ptr[c->device] : nullptr;
    if (!ws) return fail(c, VO_ERR_HIP, "vo_match_ratio_static: the static workspace of this device was not set up");
    VO_HIP(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // The table needs no clearing: the search writes both entries of every query row below nq
    // (-1 where a neighbour is missing) and the ratio kernel reads only those rows; rows left
    // from a larger earlier call are never looked at.  Empty inputs give a count of 0, written by
    // the ratio kernel itself (nq = 0: no row passes; nt = 0: every neighbour is -1).
    int32_t* idx2 = ws;
    int32_t* dist2 = ws + (size_t)kStaticMaxQueries * 2;
    if (nq > 0) {
        // the matrix-core kernel; four wavefronts per workgroup suit every size this entry point
        // admits
        dim3 grid((nq + 63) / 64, 1);
        hipLaunchKernelGGL((knn2_mfma_kernel<4, 2>), grid, dim3(256), 0, s, d_q, (const int32_t*)nullptr, 1, nq, (size_t)0, d_t,
                           (const int32_t*)nullptr, nt, (size_t)0, idx2, dist2, 0x7FFFFFFF, (unsigned long long*)nullptr);
    }
    hipLaunchKernelGGL(ratio_kernel, dim3(1), dim3(256), 0, s, (const int32_t*)idx2, (const int32_t*)dist2,
                       (const int32_t*)nullptr, 1, nq, ratio, (const vo_keypoint*)nullptr, (const vo_keypoint*)nullptr,
                       (size_t)0, d_query_idx, d_train_idx, d_distance, (float*)nullptr, (float*)nullptr, d_n_out);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

extern "C" int vo_hamming_knn2_dev(vo_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx2,
                                   int32_t* d_dist2) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_idx2 || !d_dist2)) || (nt > 0 && !d_t))
        return fail(c, VO_ERR_INVALID, "vo_hamming_knn2_dev: bad argument");
    if (((uintptr_t)d_q | (uintptr_t)d_t) & 15)
        return fail(c, VO_ERR_INVALID, "vo_hamming_knn2_dev: descriptor arrays must be 16-byte aligned");
    VO_HIP(c, hipSetDevice(c->device));
    return knn2_launch(c, d_q, nullptr, 1, nq, 0, d_t, nullptr, nt, 0, 1, d_idx2, d_dist2);
}

extern "C" int vo_hamming_knn2(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                               int32_t* dist2) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || (nq > 0 && (!q || !idx2 || !dist2)) || (nt > 0 && !t))
        return fail(c, VO_ERR_INVALID, "vo_hamming_knn2: bad argument");
    if (nq == 0) return VO_OK;
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->st_q, (size_t)nq * 32))) return rc;
    if ((rc = ensure(c, c->st_t, (size_t)std::max(nt, 1) * 32))) return rc;
    if ((rc = ensure(c, c->st_idx, (size_t)nq * 8))) return rc;
    if ((rc = ensure(c, c->st_dist, (size_t)nq * 8))) return rc;
    VO_HIP(c, hipMemcpyAsync(c->st_q.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (nt > 0) VO_HIP(c, hipMemcpyAsync(c->st_t.p, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = knn2_launch(c, (const uint8_t*)c->st_q.p, nullptr, 1, nq, 0, (const uint8_t*)c->st_t.p, nullptr, nt, 0, 1,
                          (int32_t*)c->st_idx.p, (int32_t*)c->st_dist.p)))
        return rc;
    VO_HIP(c, hipMemcpyAsync(idx2, c->st_idx.p, (size_t)nq * 8, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(dist2, c->st_dist.p, (size_t)nq * 8, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

extern "C" int vo_match_ratio(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, float ratio,
                              int32_t* query_idx, int32_t* train_idx, float* distance, int* n_out) {
    if (!c) return VO_ERR_INVALID;
    if (!n_out) return fail(c, VO_ERR_INVALID, "vo_match_ratio: n_out is NULL");
    *n_out = 0;
    if (nq <= 0 || nt <= 0 || !q || !t) return VO_OK;   // empty descriptors -> empty result
    if (!query_idx || !train_idx || !distance) return fail(c, VO_ERR_INVALID, "vo_match_ratio: NULL output");
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->st_q, (size_t)nq * 32))) return rc;
    if ((rc = ensure(c, c->st_t, (size_t)nt * 32))) return rc;
    if ((rc = ensure(c, c->st_idx, (size_t)nq * 8))) return rc;
    if ((rc = ensure(c, c->st_dist, (size_t)nq * 8))) return rc;
    if ((rc = ensure(c, c->m_qidx, (size_t)nq * 4))) return rc;
    if ((rc = ensure(c, c->m_tidx, (size_t)nq * 4))) return rc;
    if ((rc = ensure(c, c->m_dist, (size_t)nq * 4))) return rc;
    if ((rc = ensure(c, c->m_cnt, 64))) return rc;
    if ((rc = join_side(c))) return rc;   // a pipeline call's tail may still use the match buffers
    VO_HIP(c, hipMemcpyAsync(c->st_q.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->st_t.p, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = knn2_launch(c, (const uint8_t*)c->st_q.p, nullptr, 1, nq, 0, (const uint8_t*)c->st_t.p, nullptr, nt, 0, 1,
                          (int32_t*)c->st_idx.p, (int32_t*)c->st_dist.p)))
        return rc;
    if ((rc = ratio_launch(c, (const int32_t*)c->st_idx.p, (const int32_t*)c->st_dist.p, nullptr, 1, nq, 1, ratio, nullptr,
                           nullptr, 0, (int32_t*)c->m_qidx.p, (int32_t*)c->m_tidx.p, (float*)c->m_dist.p, nullptr,
                           nullptr, (int32_t*)c->m_cnt.p, c->stream)))
        return rc;
    int32_t m = 0;
    VO_HIP(c, hipMemcpyAsync(&m, c->m_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    if (m > 0) {
        VO_HIP(c, hipMemcpyAsync(query_idx, c->m_qidx.p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipMemcpyAsync(train_idx, c->m_tidx.p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipMemcpyAsync(distance, c->m_dist.p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
    }
    *n_out = m;
    return VO_OK;
}

// One large match over several GPUs of one node: the distance matrix is sharded by query rows
// (contiguous blocks, vo_block_range), the train set is replicated, and a query's two nearest
// rows are complete inside its shard, so no reduction follows - only the shards' result rows are
// brought together.  Host-buffer form: context k, on a host thread of its own, uploads its query
// block and the train set, searches, and downloads straight into rows [begin, end) of the
// caller's tables.
extern "C" int vo_hamming_knn2_multi(vo_ctx* const* ctxs, int n_ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                                     int32_t* idx2, int32_t* dist2) {
    if (!ctxs || n_ctx <= 0) return VO_ERR_INVALID;
    for (int k = 0; k < n_ctx; ++k)
        if (!ctxs[k]) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || (nq > 0 && (!q || !idx2 || !dist2)) || (nt > 0 && !t))
        return fail(ctxs[0], VO_ERR_INVALID, "vo_hamming_knn2_multi: bad argument");
    for (int a = 0; a < n_ctx; ++a)
        for (int b = a + 1; b < n_ctx; ++b)
            if (ctxs[a] == ctxs[b])
                return fail(ctxs[0], VO_ERR_INVALID, "vo_hamming_knn2_multi: a context is listed twice (contexts are not thread-safe)");
    std::vector<int> rc((size_t)n_ctx, VO_OK);
    std::vector<std::thread> workers;
    for (int k = 0; k < n_ctx; ++k) {
        int b = 0, e = 0;
        vo_block_range(nq, k, n_ctx, &b, &e);
        if (e <= b) continue;   // more contexts than query rows
        workers.emplace_back([=, &rc] {
            rc[(size_t)k] = vo_hamming_knn2(ctxs[k], q + (size_t)b * 32, e - b, t, nt, idx2 + (size_t)b * 2, dist2 + (size_t)b * 2);
        });
    }
    for (auto& w : workers) w.join();
    for (int k = 0; k < n_ctx; ++k)
        if (rc[(size_t)k] != VO_OK) {
            if (k != 0) ctxs[0]->err = "context " + std::to_string(k) + ": " + ctxs[k]->err;
            return rc[(size_t)k];
        }
    return VO_OK;
}

// Device-resident form.  d_q[k]: the query rows of block k (vo_block_range(nq, k, n_ctx)) on
// context k's GPU; d_t[k]: the whole train set on that GPU; d_idx2 / d_dist2: the full nq x 2
// tables on context 0's GPU.  Context k searches into a table of its own and sends its rows to
// their place in context 0's tables with one peer copy per table on its own stream (over xGMI
// between two GPUs; an ordinary device copy when both contexts sit on one GPU) - the path's one
// exchange step.  Everything is enqueued from the calling thread; the tables are complete once
// vo_sync() has returned for every context.
extern "C" int vo_hamming_knn2_multi_dev(vo_ctx* const* ctxs, int n_ctx, const uint8_t* const* d_q, int nq,
                                         const uint8_t* const* d_t, int nt, int32_t* d_idx2, int32_t* d_dist2) {
    if (!ctxs || n_ctx <= 0) return VO_ERR_INVALID;
    for (int k = 0; k < n_ctx; ++k)
        if (!ctxs[k]) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || !d_q || !d_t || (nq > 0 && (!d_idx2 || !d_dist2)))
        return fail(ctxs[0], VO_ERR_INVALID, "vo_hamming_knn2_multi_dev: bad argument");
    for (int a = 0; a < n_ctx; ++a)
        for (int b = a + 1; b < n_ctx; ++b)
            if (ctxs[a] == ctxs[b])
                return fail(ctxs[0], VO_ERR_INVALID, "vo_hamming_knn2_multi_dev: a context is listed twice");
    for (int k = 0; k < n_ctx; ++k) {
        vo_ctx* c = ctxs[k];
        int b = 0, e = 0;
        vo_block_range(nq, k, n_ctx, &b, &e);
        const int rows = e - b;
        if (rows <= 0) continue;
        if (!d_q[k] || (nt > 0 && !d_t[k])) return fail(ctxs[0], VO_ERR_INVALID, "vo_hamming_knn2_multi_dev: a block's pointer is NULL");
        if (((uintptr_t)d_q[k] | (uintptr_t)d_t[k]) & 15)
            return fail(ctxs[0], VO_ERR_INVALID, "vo_hamming_knn2_multi_dev: descriptor arrays must be 16-byte aligned");
        auto relay = [&](int code) {
            if (k != 0) ctxs[0]->err = "context " + std::to_string(k) + ": " + c->err;
            return code;
        };
        if (hipSetDevice(c->device) != hipSuccess) return relay(fail(c, VO_ERR_HIP, "hipSetDevice failed"));
        int32_t *li = d_idx2 + (size_t)b * 2, *ld = d_dist2 + (size_t)b * 2;
        if (k != 0) {   // context 0 writes its rows in place
            int rc;
            if ((rc = ensure(c, c->st_idx, (size_t)rows * 8)) || (rc = ensure(c, c->st_dist, (size_t)rows * 8))) return relay(rc);
            if ((rc = join_side(c))) return relay(rc);   // a pipeline call's tail may still read these tables
            li = (int32_t*)c->st_idx.p;
            ld = (int32_t*)c->st_dist.p;
        }
        if (int rc = knn2_launch(c, d_q[k], nullptr, 1, rows, 0, d_t[k], nullptr, nt, 0, 1, li, ld)) return relay(rc);
        if (k != 0) {
            const hipError_t e1 = hipMemcpyPeerAsync(d_idx2 + (size_t)b * 2, ctxs[0]->device, li, c->device, (size_t)rows * 8, c->stream);
            const hipError_t e2 = hipMemcpyPeerAsync(d_dist2 + (size_t)b * 2, ctxs[0]->device, ld, c->device, (size_t)rows * 8, c->stream);
            if (e1 != hipSuccess || e2 != hipSuccess)
                return relay(fail(c, VO_ERR_HIP, std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2)));
        }
    }
    return VO_OK;
}
