// ubench_int2.hip -- which property makes an integer VALU op "slow" on gfx950: VOP3 encoding, operand count, opcode?
// Also measures the real shader clock (s_memtime vs s_memrealtime) under this load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define BODY8(INS)                                                                                 \
    asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sb) : "vcc");

#define KERNEL(NAME, INS, T)                                                      \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c, uint32_t sb, unsigned long long *clk) \
    {                                                                             \
        T t = threadIdx.x;                                                        \
        T a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < ITER; i++) { BODY8(INS) BODY8(INS) }                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        T x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                              \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)((unsigned long long)x >> 32); \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
    }

#define I_ADD_E32(r) "v_add_u32_e32 " #r ", " #r ", %8"
#define I_ADD_E64(r) "v_add_u32_e64 " #r ", " #r ", %8"
#define I_MAD24_VVV(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9"
#define I_MAD24_VCV(r) "v_mad_u32_u24 " #r ", " #r ", 15, %9"
#define I_MAD24_VSV(r) "v_mad_u32_u24 " #r ", " #r ", %10, %9"
#define I_MAD24_VCC(r) "v_mad_u32_u24 " #r ", " #r ", 15, 17"
#define I_MUL24_E32(r) "v_mul_u32_u24_e32 " #r ", 15, " #r
#define I_MULLO_VS(r) "v_mul_lo_u32 " #r ", " #r ", %10"
#define I_MAD64_VC(r) "v_mad_u64_u32 " #r ", vcc, %8, 15, " #r
#define I_MAD64_VV(r) "v_mad_u64_u32 " #r ", vcc, %8, %9, " #r
#define I_MAD64_NULL(r) "v_mad_u64_u32 " #r ", null, %8, %9, " #r
#define I_FMA_E64(r) "v_fma_f32 " #r ", " #r ", %8, %9"
#define I_FMAC_E32(r) "v_fmac_f32_e32 " #r ", %8, %9"
#define I_ADD3_VCC(r) "v_add3_u32 " #r ", " #r ", 3, 5"
#define I_CNDMASK(r) "v_cndmask_b32_e32 " #r ", " #r ", %8, vcc"
#define I_ADDCO(r) "v_add_co_u32_e32 " #r ", vcc, " #r ", %8"
#define I_ADDC(r) "v_addc_co_u32_e32 " #r ", vcc, " #r ", %8, vcc"
#define I_MOV(r) "v_mov_b32_e32 " #r ", %8"
#define I_LSHL_E32(r) "v_lshlrev_b32_e32 " #r ", 3, " #r
#define I_XOR_E32(r) "v_xor_b32_e32 " #r ", " #r ", %8"
#define I_LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 1, " #r
#define I_NOP(r) "s_nop 0"
#define I_PKADD(r) "v_pk_add_u16 " #r ", " #r ", %8"
#define I_ADD64X(r) "v_lshl_add_u64 " #r ", " #r ", 0, " #r

KERNEL(k_add_e32, I_ADD_E32, uint32_t) KERNEL(k_add_e64, I_ADD_E64, uint32_t)
KERNEL(k_mad24_vvv, I_MAD24_VVV, uint32_t) KERNEL(k_mad24_vcv, I_MAD24_VCV, uint32_t) KERNEL(k_mad24_vsv, I_MAD24_VSV, uint32_t)
KERNEL(k_mad24_vcc, I_MAD24_VCC, uint32_t) KERNEL(k_mul24_e32, I_MUL24_E32, uint32_t) KERNEL(k_mullo_vs, I_MULLO_VS, uint32_t)
KERNEL(k_mad64_vc, I_MAD64_VC, uint64_t) KERNEL(k_mad64_vv, I_MAD64_VV, uint64_t)
KERNEL(k_fma_e64, I_FMA_E64, uint32_t) KERNEL(k_fmac_e32, I_FMAC_E32, uint32_t) KERNEL(k_add3_vcc, I_ADD3_VCC, uint32_t)
KERNEL(k_cndmask, I_CNDMASK, uint32_t) KERNEL(k_addco, I_ADDCO, uint32_t) KERNEL(k_addc, I_ADDC, uint32_t) KERNEL(k_mov, I_MOV, uint32_t)
KERNEL(k_lshl_e32, I_LSHL_E32, uint32_t) KERNEL(k_xor_e32, I_XOR_E32, uint32_t) KERNEL(k_lshladd64, I_LSHLADD64, uint64_t)
KERNEL(k_nop, I_NOP, uint32_t) KERNEL(k_pkadd, I_PKADD, uint32_t)

template <typename K>
static int run(const char *name, K kern, uint32_t *d, unsigned long long *dclk, int cus, int waves_per_simd)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = cus * waves_per_simd; // each block = 4 waves = 1 per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 7u, dclk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 7u, dclk);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2]; CHK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0) ; // memrealtime ticks at 100 MHz -> ns = ticks*10
    const double wave_insts = (double)blocks * 4 * ITER * 16;
    const double per_s = wave_insts / (ms * 1e-3);
    // in-kernel cycles per wave-instr per SIMD (block 0's wave 0 view): cycles / (instrs issued by all waves on that SIMD)
    const double cyc_per_inst = (double)clk[0] / (ITER * 16.0 * waves_per_simd);
    printf("%-16s w/SIMD=%d %8.3f ms %8.1f G winst/s  clk %.2f GHz  %.2f cyc/winst/SIMD (in-kernel)\n", name, waves_per_simd, ms, per_s / 1e9, ghz, cyc_per_inst);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *d; CHK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
    unsigned long long *dclk; CHK(hipMalloc(&dclk, 16));
    for (int w : {8, 2, 1}) {
        run("add_e32", k_add_e32, d, dclk, cus, w); run("add_e64", k_add_e64, d, dclk, cus, w);
        run("xor_e32", k_xor_e32, d, dclk, cus, w); run("lshl_e32", k_lshl_e32, d, dclk, cus, w); run("mov_e32", k_mov, d, dclk, cus, w);
        run("mul24_e32", k_mul24_e32, d, dclk, cus, w);
        run("mad24 v,v,v", k_mad24_vvv, d, dclk, cus, w); run("mad24 v,15,v", k_mad24_vcv, d, dclk, cus, w);
        run("mad24 v,s,v", k_mad24_vsv, d, dclk, cus, w); run("mad24 v,15,17", k_mad24_vcc, d, dclk, cus, w);
        run("mul_lo v,s", k_mullo_vs, d, dclk, cus, w);
        run("mad64 v,15,v64", k_mad64_vc, d, dclk, cus, w); run("mad64 v,v,v64", k_mad64_vv, d, dclk, cus, w);
        run("lshl_add_u64", k_lshladd64, d, dclk, cus, w);
        run("fma_f32 e64", k_fma_e64, d, dclk, cus, w); run("fmac_f32 e32", k_fmac_e32, d, dclk, cus, w);
        run("add3 v,3,5", k_add3_vcc, d, dclk, cus, w); run("cndmask_e32", k_cndmask, d, dclk, cus, w);
        run("add_co_e32", k_addco, d, dclk, cus, w); run("addc_co_e32", k_addc, d, dclk, cus, w);
        run("pk_add_u16", k_pkadd, d, dclk, cus, w); run("s_nop 0", k_nop, d, dclk, cus, w);
        printf("\n");
    }
    return 0;
}
