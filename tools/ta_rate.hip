// Global-load issue cost on gfx950 by access pattern: cycles per wave64 load instruction and CU
// (8 workgroups of 4 waves per CU, data resident in L2 / L1: a 16 MB buffer walked repeatedly).
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate tools/ta_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

constexpr int N_IT = 512;
constexpr uint32_t BUF = 16u << 20;   // bytes

// pattern: byte address of lane's element for iteration `it`
//  0: coalesced dwords (lane i -> 4*i), rows advance with it
//  1: 8 segments of 8 lanes (32 B each), segments 640 B apart (image rows), dword aligned
//  2: same, segments start at an odd byte (unaligned dwords)
//  3: 64 lanes each in its own row (640 B apart), aligned
//  4: 16 segments of 4 lanes (16 B), 640 B apart
//  5: 2 segments of 32 lanes (128 B), 640 B apart
template <int PAT, int W>
__global__ __launch_bounds__(256) void k_ta(const uint8_t* __restrict__ buf, uint32_t* out, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    uint32_t acc = 0;
    uint32_t base = (wave * 40960u + seed * 64u) & (BUF - 1);
    for (int it = 0; it < N_IT; ++it) {
        uint32_t off;
        if (PAT == 0) off = lane * (4 * W);
        else if (PAT == 1) off = (lane >> 3) * 640u + (lane & 7) * (4 * W);
        else if (PAT == 2) off = (lane >> 3) * 640u + (lane & 7) * (4 * W) + 1;
        else if (PAT == 3) off = lane * 640u;
        else if (PAT == 4) off = (lane >> 2) * 640u + (lane & 3) * (4 * W);
        else if (PAT == 5) off = (lane >> 5) * 640u + (lane & 31) * (4 * W);
        else if (PAT == 6) off = (lane >> 1) * 640u + (lane & 1) * (4 * W);          // 32 rows x 2 lanes
        else if (PAT == 7) off = (lane >> 1) * 640u + (lane & 1) * (4 * W) + 3;      // same, unaligned
        else if (PAT == 8) off = (lane / 3) * 640u + (lane % 3) * (4 * W);           // 21 rows x 3 lanes
        else if (PAT == 9) off = (lane / 5) * 640u + (lane % 5) * (4 * W);           // 12 rows x 5 lanes
        else if (PAT == 10) off = (lane >> 2) * 640u + (lane & 3) * (4 * W) + 3;     // 16 rows x 4 lanes, unaligned
        else off = (lane >> 3) * 640u + (lane & 7) * (4 * W) + 3;                    // 8 rows x 8 lanes, unaligned by 3
        const uint8_t* p = buf + ((base + off + (uint32_t)it * 5120u) & (BUF - 1 - 63));
        if (PAT == 2) p += 1;
        if (PAT == 7 || PAT == 10 || PAT == 11) p += 3;
        if (W == 1) {
            uint32_t v;
            __builtin_memcpy(&v, p, 4);
            acc ^= v;
        } else if (W == 2) {
            uint32_t v[2];
            __builtin_memcpy(v, p, 8);
            acc ^= v[0] ^ v[1];
        } else {
            uint32_t v[4];
            __builtin_memcpy(v, p, 16);
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int PAT, int W>
static void run(const char* name, const uint8_t* buf, uint32_t* d) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_ta<PAT, W>), dim3(256 * 8), dim3(256), 0, 0, buf, d, 1u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_ta<PAT, W>), dim3(256 * 8), dim3(256), 0, 0, buf, d, 2u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts_per_cu = 32.0 * N_IT;
    const double cyc = ms * 1e-3 * 2.4e9 / insts_per_cu;
    printf("%-44s %7.3f ms  %7.1f cycles per load instruction and CU  (%5.1f B/clk)\n", name, ms, cyc, 64.0 * 4 * W / cyc);
}

int main() {
    uint8_t* buf;
    uint32_t* d;
    hipMalloc(&buf, BUF + 4096);
    hipMemset(buf, 1, BUF + 4096);
    hipMalloc(&d, 256);
    run<0, 1>("coalesced dword", buf, d);
    run<0, 4>("coalesced dwordx4", buf, d);
    run<5, 1>("2 rows x 32 lanes, dword", buf, d);
    run<1, 1>("8 rows x 8 lanes, dword aligned", buf, d);
    run<2, 1>("8 rows x 8 lanes, dword unaligned", buf, d);
    run<4, 1>("16 rows x 4 lanes, dword", buf, d);
    run<3, 1>("64 rows x 1 lane, dword", buf, d);
    run<3, 2>("64 rows x 1 lane, dwordx2", buf, d);
    run<3, 4>("64 rows x 1 lane, dwordx4", buf, d);
    run<1, 2>("8 rows x 8 lanes, dwordx2", buf, d);
    run<1, 4>("8 rows x 8 lanes, dwordx4", buf, d);
    run<6, 4>("32 rows x 2 lanes, dwordx4", buf, d);
    run<7, 4>("32 rows x 2 lanes, dwordx4 unaligned", buf, d);
    run<4, 2>("16 rows x 4 lanes, dwordx2", buf, d);
    run<10, 2>("16 rows x 4 lanes, dwordx2 unaligned", buf, d);
    run<11, 1>("8 rows x 8 lanes, dword unaligned by 3", buf, d);
    run<8, 4>("21 rows x 3 lanes, dwordx4", buf, d);
    run<9, 2>("12 rows x 5 lanes, dwordx2", buf, d);
    run<4, 4>("16 rows x 4 lanes, dwordx4", buf, d);
    hipFree(buf);
    hipFree(d);
    return 0;
}
