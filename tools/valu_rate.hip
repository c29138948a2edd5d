// Issue-rate microbenchmark of a few gfx950 vector instructions: every kernel runs N_IT
// iterations of 16 independent instructions of one kind per lane on a full GPU (8 waves per
// SIMD) and the host prints the cycles one wave64 instruction occupies its SIMD
// (4 = full rate).   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int N_IT = 4096;

#define KERNEL(name, ASM)                                                                     \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) {               \
        uint32_t a[16];                                                                       \
        uint32_t b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;                   \
        for (int i = 0; i < 16; ++i) a[i] = b + i;                                            \
        for (int it = 0; it < N_IT; ++it) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                     \
        uint32_t s = 0;                                                                       \
        for (int i = 0; i < 16; ++i) s ^= a[i];                                               \
        if (s == 0x12345678u) out[0] = s;                                                     \
    }

// two instructions per slot on different registers (%0 = a[i], %1 = a[i ^ 8]; %2, %3 constant)
#define KERNEL2(name, ASM)                                                                    \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) {               \
        uint32_t a[16];                                                                       \
        uint32_t b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;                   \
        for (int i = 0; i < 16; ++i) a[i] = b + i;                                            \
        for (int it = 0; it < N_IT; ++it) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                                    \
                asm volatile(ASM : "+v"(a[i]), "+v"(a[i ^ 8]) : "v"(b), "v"(c));              \
        }                                                                                     \
        uint32_t s = 0;                                                                       \
        for (int i = 0; i < 16; ++i) s ^= a[i];                                               \
        if (s == 0x12345678u) out[0] = s;                                                     \
    }
KERNEL2(k2_add_add_dep, "v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3")
KERNEL2(k2_add_add_ind, "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3")
KERNEL2(k2_add_perm_ind, "v_add_u32 %0, %0, %2\n v_perm_b32 %1, %1, %2, %3")
KERNEL2(k2_xor_bcnt_ind, "v_xor_b32 %0, %0, %2\n v_bcnt_u32_b32 %1, %1, %3")
KERNEL2(k2_perm_perm_ind, "v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3")
KERNEL2(k2_add_add_perm, "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n v_perm_b32 %0, %0, %2, %3")
KERNEL2(k2_xor_xor_bcnt_bcnt, "v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n v_bcnt_u32_b32 %0, %0, %3\n v_bcnt_u32_b32 %1, %1, %2")
KERNEL2(k2_fma_fma_ind, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3")
KERNEL2(k2_fma_fma_dep, "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3")
// runs: RUN_F fast instructions (xor on a[0..]) followed by RUN_S slow ones (bcnt), all independent
template <int RUN_F, int RUN_S>
__global__ __launch_bounds__(256) void k_runs(uint32_t* out, uint32_t seed) {
    uint32_t a[16];
    uint32_t b = threadIdx.x * 2654435761u + seed, c = b ^ 0x5bd1e995u;
    for (int i = 0; i < 16; ++i) a[i] = b + i;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < RUN_F; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(b));
#pragma unroll
        for (int i = 0; i < RUN_S; ++i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(c));
    }
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) s ^= a[i];
    if (s == 0x12345678u) out[0] = s;
}
template <int RUN_F, int RUN_S>
static void run_runs(uint32_t* d) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_runs<RUN_F, RUN_S>), dim3(2048), dim3(256), 0, 0, d, 1u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_runs<RUN_F, RUN_S>), dim3(2048), dim3(256), 0, 0, d, 2u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double cycles = ms * 1e-3 * 2.4e9 / (8.0 * N_IT);
    printf("runs of %2d fast + %2d slow: %7.1f cycles per group (all at 4.2: %.1f, fast at 2.6: %.1f)\n", RUN_F, RUN_S, cycles,
           4.2 * (RUN_F + RUN_S), 2.6 * RUN_F + 4.2 * RUN_S);
}
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_min_i32, "v_min_i32 %0, %0, %1")
KERNEL(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
KERNEL(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
KERNEL(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
KERNEL(k_msad_u8, "v_msad_u8 %0, %1, %2, %0")
KERNEL(k_lerp_u8, "v_lerp_u8 %0, %0, %1, %2")
KERNEL(k_lerp_u8_0, "v_lerp_u8 %0, %0, %1, 0")
KERNEL(k_lerp_u8_sgpr, "v_lerp_u8 %0, %0, s20, 0")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80")
KERNEL(k_min3_u32, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_max3_u32, "v_max3_u32 %0, %0, %1, %2")
KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL(k_cvt_f32_ubyte, "v_cvt_f32_ubyte1 %0, %1")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_rndne, "v_rndne_f32 %0, %0")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(k_lshlrev_b64ish, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_pk_lshr_b16, "v_pk_lshrrev_b16 %0, 1, %0")
KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
KERNEL(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
KERNEL(k_xor_sgpr, "v_xor_b32 %0, s20, %0")
KERNEL(k_xor_bcnt, "v_xor_b32 %0, %1, %0\n v_bcnt_u32_b32 %0, %0, %2")
KERNEL(k_xor_sgpr_bcnt, "v_xor_b32 %0, s20, %0\n v_bcnt_u32_b32 %0, %0, %2")
KERNEL(k_add_lit, "v_add_u32 %0, 0x12345, %0")
KERNEL(k_and_lit, "v_and_b32 %0, 0x00FF00FF, %0")
KERNEL(k_add_inl, "v_add_u32 %0, 17, %0")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 1, %0")
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL(k_add_u16, "v_add_u16 %0, %0, %1")
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL(k_and_e64, "v_and_b32_e64 %0, %0, %1")
KERNEL(k_mix_add_perm, "v_add_u32 %0, %0, %1\n v_perm_b32 %0, %0, %1, %2")
KERNEL(k_mix_add_min, "v_add_u32 %0, %0, %1\n v_min_i32 %0, %0, %1")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
KERNEL(k_add_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

template <typename K>
static void run(const char* name, K kern, int insts_per_slot, uint32_t* d) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int wgs = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, 1u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, 2u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double waves_per_simd = 8.0, insts = (double)N_IT * 16 * insts_per_slot * waves_per_simd;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-18s %8.3f ms  %6.2f cycles per wave instruction (at %d MHz)\n", name, ms, cycles / insts, clk_khz / 1000);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256);
#define RUN(k) run(#k, k, 1, d)
    RUN(k_add); RUN(k_pk_min_i16); RUN(k_pk_max_u16); RUN(k_pk_add_i16); RUN(k_pk_sub_u16); RUN(k_pk_lshr_b16);
    RUN(k_perm); RUN(k_alignbit); RUN(k_alignbyte); RUN(k_mad_u24); RUN(k_mul_u24); RUN(k_mul_lo); RUN(k_bfe);
    RUN(k_bcnt); RUN(k_mbcnt); RUN(k_dot4); RUN(k_dot2_u16); RUN(k_sad_u8); RUN(k_lerp_u8); RUN(k_lerp_u8_0); RUN(k_lerp_u8_sgpr); RUN(k_bitop3); RUN(k_min3_u32); RUN(k_max3_u32); RUN(k_msad_u8); RUN(k_sub_sdwa);
    RUN(k_and_or); RUN(k_lshl_add); RUN(k_add3); RUN(k_bfi); RUN(k_min_i32); RUN(k_min3_i32); RUN(k_med3_i32);
    RUN(k_cvt_f32_ubyte); RUN(k_fma_f32); RUN(k_rndne); RUN(k_cvt_i32_f32); RUN(k_lshlrev_b64ish); RUN(k_add);
    RUN(k_and); RUN(k_or); RUN(k_xor); RUN(k_sub); RUN(k_lshl); RUN(k_ashr); RUN(k_max_u32); RUN(k_mov); RUN(k_add_f32);
    RUN(k_mul_f32); RUN(k_fmac_f32); RUN(k_or3); RUN(k_lshl_or); RUN(k_add_u16); RUN(k_add_co); RUN(k_cmp); RUN(k_cndmask);
    RUN(k_add_e64); RUN(k_and_e64); RUN(k_cvt_f32_i32); RUN(k_add_sdwa); RUN(k_add_dpp);
    RUN(k_xor_sgpr); RUN(k_add_lit); RUN(k_and_lit); RUN(k_add_inl);
    run("k_xor_bcnt", k_xor_bcnt, 2, d);
    run("k_xor_sgpr_bcnt", k_xor_sgpr_bcnt, 2, d);
    run("k2_add_add_dep", k2_add_add_dep, 2, d);
    run("k2_add_add_ind", k2_add_add_ind, 2, d);
    run("k2_add_perm_ind", k2_add_perm_ind, 2, d);
    run("k2_xor_bcnt_ind", k2_xor_bcnt_ind, 2, d);
    run("k2_perm_perm_ind", k2_perm_perm_ind, 2, d);
    run("k2_add_add_perm", k2_add_add_perm, 3, d);
    run("k2_xor_xor_bcnt_bcnt", k2_xor_xor_bcnt_bcnt, 4, d);
    run("k2_fma_fma_ind", k2_fma_fma_ind, 2, d);
    run("k2_fma_fma_dep", k2_fma_fma_dep, 2, d);
    run_runs<1, 1>(d); run_runs<2, 2>(d); run_runs<4, 4>(d); run_runs<8, 8>(d); run_runs<16, 16>(d); run_runs<32, 32>(d);
    run_runs<8, 0>(d); run_runs<0, 8>(d); run_runs<8, 1>(d); run_runs<16, 1>(d);
    run("k_mix_add_perm", k_mix_add_perm, 2, d);
    run("k_mix_add_min", k_mix_add_min, 2, d);
    run("k_cmp_cndmask", k_cmp_cndmask, 2, d);
    hipFree(d);
    return 0;
}
