// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on a known byte count:
// streams a 1 GiB buffer with one dword per lane (the access width of the FAST tile loads)
// and with 16 B per lane, and writes 256 MiB with one dword per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read_dword(const uint32_t* __restrict__ p, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_dwordx4(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write_dword(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = 1ull << 30;
    uint32_t *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        read_dword<<<8192, 256>>>(buf, bytes / 4, out);
        read_dwordx4<<<8192, 256>>>((const uint4*)buf, bytes / 16, out);
        write_dword<<<8192, 256>>>(buf, (bytes / 4) / 4);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("read_dword %zu B, read_dwordx4 %zu B, write_dword %zu B per launch\n", bytes, bytes, bytes / 4);
    return 0;
}
