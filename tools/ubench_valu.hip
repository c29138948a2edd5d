// Micro-benchmark: issue rate of the integer VALU instructions the Hamming kernel is made
// of (v_xor_b32, v_bcnt_u32_b32, v_add3_u32, v_min_u32) on gfx950.  Prints lane-ops/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] ^ (a[(i + 1) & 7] + 1u);                       // xor (+add)
                if (OP == 1) a[i] = __builtin_popcount(a[i] ^ 0x9E3779B9u) + a[(i + 1) & 7];   // xor + bcnt
                if (OP == 2) a[i] = min(a[i], a[(i + 3) & 7] ^ 0x55u);                  // xor + min
                if (OP == 3) a[i] = a[i] + a[(i + 1) & 7] + a[(i + 2) & 7];             // add3
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096 * 256 * 4);
    const char* names[4] = {"xor+add (2 ops)", "xor+bcnt_acc (2 ops)", "xor+min (2 ops)", "add3 (1 op)"};
    const int ops[4] = {2, 2, 2, 1};
    for (int op = 0; op < 4; ++op) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        int iters = 2000, blocks = 4096;
        auto launch = [&]() {
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters);
        };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double lane_ops = (double)blocks * 256 * iters * 64.0 * ops[op];
        printf("%-24s %.3f ms  %.3e lane-ops/s\n", names[op], ms, lane_ops / (ms * 1e-3));
    }
    return 0;
}
