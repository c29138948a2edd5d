// fp64 issue and latency microbenchmark for gfx950: cycles a wave64 v_fma_f64 / v_mul_f64 /
// v_add_f64 / v_rcp_f64 occupies its SIMD with CH independent dependent chains per lane and W
// wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 -o f64_rate tools/f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N_IT = 2048;

template <int CH, int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
    double a[CH];
    const double b = 1.0 + 1e-9 * threadIdx.x + seed, c = 1e-12 * threadIdx.x;
    for (int i = 0; i < CH; ++i) a[i] = b + i;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"((float)b), "v"((float)c));
            }
    }
    double s = 0;
    for (int i = 0; i < CH; ++i) s += a[i];
    if (s == 0.12345) out[0] = s;
}

template <int CH, int OP>
static void run(double* d, const char* name, int waves_per_simd) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs: a 256-thread block is one wave per SIMD of a CU
    hipLaunchKernelGGL((k<CH, OP>), dim3(blocks), dim3(256), 0, 0, d, 1.0);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<CH, OP>), dim3(blocks), dim3(256), 0, 0, d, 2.0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)N_IT * 16 * waves_per_simd);
    printf("%-10s chains %2d waves/SIMD %d: %6.2f cycles per wave instruction (per wave: %6.2f)\n", name, CH, waves_per_simd, cyc,
           cyc * waves_per_simd);
}

int main() {
    double* d;
    hipMalloc(&d, 64);
    for (int w : {1, 2, 4, 8}) {
        run<1, 0>(d, "fma_f64", w);
        run<2, 0>(d, "fma_f64", w);
        run<4, 0>(d, "fma_f64", w);
        run<16, 0>(d, "fma_f64", w);
    }
    for (int w : {1, 8}) {
        run<1, 1>(d, "mul_f64", w);
        run<16, 1>(d, "mul_f64", w);
        run<1, 2>(d, "add_f64", w);
        run<16, 2>(d, "add_f64", w);
        run<1, 3>(d, "rcp_f64", w);
        run<16, 3>(d, "rcp_f64", w);
        run<1, 4>(d, "fma_f32", w);
        run<16, 4>(d, "fma_f32", w);
    }
    return 0;
}
