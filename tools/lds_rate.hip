// LDS read-rate microbenchmark on gfx950: cycles per ds_read instruction per CU for uniform
// (broadcast) and per-lane addresses, b32 / b64 / b128.  8 workgroups of 4 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int N_IT = 2048;

template <int W, bool UNIFORM>
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = i * seed;
    __syncthreads();
    uint32_t acc = 0;
    // byte offset: uniform -> the same 16-byte slot for all lanes; else consecutive slots per lane
    uint32_t base = UNIFORM ? 0u : (threadIdx.x & 63) * (W * 4);
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t addr = (base + (uint32_t)((it * 8 + j) * 16)) & 8191u & ~(uint32_t)(W * 4 - 1);
            if (W == 1) {
                uint32_t v;
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
                acc ^= v;
            } else if (W == 2) {
                uint64_t v;
                asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
                acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
            } else {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                u4 v;
                asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int W, bool U>
static void run(const char* name, uint32_t* d) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_lds<W, U>), dim3(256 * 8), dim3(256), 0, 0, d, 3u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_lds<W, U>), dim3(256 * 8), dim3(256), 0, 0, d, 5u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double insts_per_cu = 32.0 * N_IT * 8;   // 32 waves per CU
    printf("%-24s %8.3f ms  %6.2f cycles per wave instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_cu);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256);
    run<1, true>("ds_read_b32 uniform", d);
    run<1, false>("ds_read_b32 per lane", d);
    run<2, true>("ds_read_b64 uniform", d);
    run<2, false>("ds_read_b64 per lane", d);
    run<4, true>("ds_read_b128 uniform", d);
    run<4, false>("ds_read_b128 per lane", d);
    hipFree(d);
    return 0;
}
