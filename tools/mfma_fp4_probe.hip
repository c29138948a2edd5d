// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands on gfx950, as used by the
// Hamming matcher: 256-bit rows expanded to +-magnitude nibbles, four K=64 steps per 32x32 tile.
// Checks the result against popcounts on the host (which fixes the operand lane map the matcher
// relies on: operand lane l <-> row / column l & 31, the two lane halves take complementary halves
// of K) and times a dependent-free stream of the instruction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 32 bits -> four dwords of eight fp4 values: nibble i of dword j carries bit 4 i + j as its sign
__device__ __forceinline__ void expand(uint32_t x, uint32_t mag, uint32_t flip, uint32_t* o) {
    o[0] = (((x << 3) & 0x88888888u) | mag) ^ flip;
    o[1] = (((x << 2) & 0x88888888u) | mag) ^ flip;
    o[2] = (((x << 1) & 0x88888888u) | mag) ^ flip;
    o[3] = ((x & 0x88888888u) | mag) ^ flip;
}

template <int SB>
__global__ void probe_kernel(const uint32_t* __restrict__ train, const uint32_t* __restrict__ query, float* __restrict__ out, float c_base) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = c_base + (float)i;
    for (int s = 0; s < 4; ++s) {
        uint32_t a[4], b[4];
        expand(train[r * 8 + 4 * h + s], 0x44444444u, 0u, a);            // +-2
        expand(query[r * 8 + 4 * h + s], 0x66666666u, 0x88888888u, b);   // -+4: products are -8 where the bits agree
        v8i A = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
        v8i B = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0x7F7F7F7F, 0, SB);
    }
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = acc[i];
}

// NM matrix instructions (four independent accumulators) and NV independent v_med3_f32 per loop pass
template <int NM, int NV>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float* __restrict__ out) {
    v16f acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    v8i A = {(int)(0x22222222u ^ threadIdx.x), 0x2A2A2A2A, 0x22222222, 0x2222AAAA, 0, 0, 0, 0};
    v8i B = {0x22222222, (int)(0xA2A2A2A2u), 0x2222AA22, 0x22222222, 0, 0, 0, 0};
    float k0[8], k1[8], x = (float)threadIdx.x;
    for (int j = 0; j < 8; ++j) { k0[j] = 1e9f + j; k1[j] = 2e9f + j; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NM; ++k)
            acc[k & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[k & 3], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#pragma unroll
        for (int k = 0; k < NV; ++k) k1[k & 7] = __builtin_amdgcn_fmed3f(k0[k & 7], k1[k & 7], x);
        x += 1.0f;
    }
    float s = 0;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    for (int j = 0; j < 8; ++j) s += k1[j];
    if (s == 12345.678f) out[0] = s;
}

template <int NM, int NV>
void rate(const char* what, float* dout) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2048, wgs = 2048;
    rate_kernel<NM, NV><<<wgs, 256>>>(16, dout);
    (void)hipEventRecord(e0);
    rate_kernel<NM, NV><<<wgs, 256>>>(iters, dout);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 2048 workgroups of 4 wavefronts on 1024 SIMDs: 8 wavefronts per SIMD
    printf("%-46s %.3f ms: %.1f cycles per loop pass and wavefront at 2.4 GHz\n", what, ms, ms * 1e-3 * 2.4e9 / (8.0 * iters));
}

int main() {
    std::vector<uint32_t> t(32 * 8), q(32 * 8);
    srand(7);
    for (auto& v : t) v = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    for (auto& v : q) v = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    for (int k = 0; k < 8; ++k) q[5 * 8 + k] = t[9 * 8 + k];   // one exact match: query 5 = train 9
    uint32_t *dt, *dq;
    float* dout;
    (void)hipMalloc(&dt, t.size() * 4);
    (void)hipMalloc(&dq, q.size() * 4);
    (void)hipMalloc(&dout, 64 * 16 * 4);
    (void)hipMemcpy(dt, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> out(64 * 16);
    int bad = 0;
    for (int variant = 0; variant < 3; ++variant) {
        // product magnitude and the accumulator's start value: 8 and 2^11, 8 * 2^11 and 2^22, 8 * 2^11 and 2^22 + 32704
        const double unit = variant == 0 ? 8.0 : 16384.0;
        const float c_base = variant == 0 ? 2048.0f : variant == 1 ? 4194304.0f : 4194304.0f + 32704.0f;
        if (variant == 0) probe_kernel<0x7F7F7F7F><<<1, 64>>>(dt, dq, dout, c_base);
        else probe_kernel<(int)0x8A8A8A8Au><<<1, 64>>>(dt, dq, dout, c_base);
        if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
        int wrong = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int reg = 0; reg < 16; ++reg) {
                const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);   // C[row = train][col = query]
                int d = 0;
                for (int k = 0; k < 8; ++k) d += __builtin_popcount(t[row * 8 + k] ^ q[col * 8 + k]);
                const double want = 2.0 * unit * d - 256.0 * unit + c_base + reg;
                if ((double)out[lane * 16 + reg] != want) {
                    if (wrong < 6) printf("variant %d lane %d reg %d: got %.1f want %.1f (d %d)\n", variant, lane, reg, out[lane * 16 + reg], want, d);
                    ++wrong;
                }
            }
        printf("fp4 mfma hamming probe, variant %d: %d of 1024 wrong\n", variant, wrong);
        bad += wrong;
    }
    rate<8, 0>("8 matrix instructions", dout);
    rate<0, 64>("64 v_med3_f32", dout);
    rate<8, 64>("8 matrix instructions + 64 v_med3_f32", dout);
    rate<8, 96>("8 matrix instructions + 96 v_med3_f32", dout);
    rate<4, 96>("4 matrix instructions + 96 v_med3_f32", dout);
    return bad != 0;
}
