// ubench_int.hip -- integer VALU issue-rate microbenchmark for gfx950 (which multiply forms are full rate?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o gpurun_out/ubench_int ; run on the GPU box.
// Each lane runs 8 independent dependency chains of the instruction under test; result = wave-instr/clk/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define BODY8(INS)                                                                                 \
    asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));

#define I_ADD(r) "v_add_u32 " #r ", " #r ", %8"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9"
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8"
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8"
#define I_MULHI24(r) "v_mul_hi_u32_u24 " #r ", " #r ", %8"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 3, %8"
#define I_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9"
#define I_XAD(r) "v_xad_u32 " #r ", " #r ", %8, %9"
#define I_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, 22"
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 22"
#define I_AND_OR(r) "v_and_or_b32 " #r ", " #r ", %8, %9"

#define KERNEL32(NAME, INS)                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) \
    {                                                                             \
        uint32_t t = threadIdx.x;                                                 \
        uint32_t a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7; \
        for (int i = 0; i < ITER; i++) { BODY8(INS) BODY8(INS) }                  \
        out[blockIdx.x * 256 + t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;        \
    }

KERNEL32(k_add, I_ADD)
KERNEL32(k_mad24, I_MAD24)
KERNEL32(k_mullo, I_MULLO)
KERNEL32(k_mulhi, I_MULHI)
KERNEL32(k_mul24, I_MUL24)
KERNEL32(k_mulhi24, I_MULHI24)
KERNEL32(k_lshladd, I_LSHLADD)
KERNEL32(k_add3, I_ADD3)
KERNEL32(k_xad, I_XAD)
KERNEL32(k_alignbit, I_ALIGNBIT)
KERNEL32(k_perm, I_PERM)
KERNEL32(k_bfe, I_BFE)
KERNEL32(k_andor, I_AND_OR)

// 64-bit forms
#define BODY8_64(INS)                                                                              \
    asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
#define I_MAD64(r) "v_mad_u64_u32 " #r ", vcc, %8, %9, " #r
#define I_LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 1, " #r
#define I_LSHL64(r) "v_lshlrev_b64 " #r ", 3, " #r
#define KERNEL64(NAME, INS)                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) \
    {                                                                             \
        uint64_t t = threadIdx.x;                                                 \
        uint64_t a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7; \
        for (int i = 0; i < ITER; i++) { BODY8_64(INS) BODY8_64(INS) }            \
        uint64_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                       \
        out[blockIdx.x * 256 + t] = (uint32_t)x ^ (uint32_t)(x >> 32);            \
    }
KERNEL64(k_mad64, I_MAD64)
KERNEL64(k_lshladd64, I_LSHLADD64)
KERNEL64(k_lshl64, I_LSHL64)

// 64-bit add as the compiler emits it (v_add_co + v_addc)
__global__ __launch_bounds__(256) void k_add64(uint32_t *out, uint32_t b, uint32_t c)
{
    uint64_t t = threadIdx.x, bb = ((uint64_t)b << 32) | c;
    uint64_t a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            a0 += bb; a1 += bb; a2 += bb; a3 += bb; a4 += bb; a5 += bb; a6 += bb; a7 += bb;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    uint64_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    out[blockIdx.x * 256 + t] = (uint32_t)x ^ (uint32_t)(x >> 32);
}

template <typename K>
static int run(const char *name, K kern, uint32_t *d, int cus, double insts_per_iter_per_lane)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = cus * 8; // 8 blocks x 4 waves = 32 waves per CU, 8 per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts = (double)blocks * 4 * ITER * insts_per_iter_per_lane;
    const double per_s = wave_insts / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: (cus*4 SIMDs * 2.4e9) / per_s
    printf("%-14s %8.3f ms  %8.2f G wave-instr/s  -> %5.2f clk/wave-instr/SIMD @2.4GHz  (%.1f T lane-ops/s)\n", name, ms,
           per_s / 1e9, cus * 4 * 2.4e9 / per_s, per_s * 64 / 1e12);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    uint32_t *d; CHK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
    run("v_add_u32", k_add, d, cus, 16); run("v_mad_u32_u24", k_mad24, d, cus, 16); run("v_mul_u32_u24", k_mul24, d, cus, 16);
    run("v_mul_hi_u24", k_mulhi24, d, cus, 16); run("v_mul_lo_u32", k_mullo, d, cus, 16); run("v_mul_hi_u32", k_mulhi, d, cus, 16);
    run("v_mad_u64_u32", k_mad64, d, cus, 16); run("v_lshl_add_u32", k_lshladd, d, cus, 16); run("v_add3_u32", k_add3, d, cus, 16);
    run("v_xad_u32", k_xad, d, cus, 16); run("v_alignbit_b32", k_alignbit, d, cus, 16); run("v_perm_b32", k_perm, d, cus, 16);
    run("v_bfe_u32", k_bfe, d, cus, 16); run("v_and_or_b32", k_andor, d, cus, 16);
    run("v_lshl_add_u64", k_lshladd64, d, cus, 16); run("v_lshlrev_b64", k_lshl64, d, cus, 16);
    run("add64(co+addc)", k_add64, d, cus, 16);
    return 0;
}
