// ubench_int3.hip -- issue rates of the instructions the r01 tables left open: 64-bit compares, selects with an SGPR or a
// quiescent VCC mask (r01's v_cndmask figure was 10x off every other 32-bit op: suspected measurement artefact), borrow
// chains, 64-bit moves / shifts, DPP moves and the packed / dot integer ops.  Same harness as ubench_int2.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define BODY8(INS)                                                                                 \
    asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sm), "v"(b64) : "vcc");

#define KERNEL(NAME, INS, T)                                                      \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c, unsigned long long sm, unsigned long long *clk) \
    {                                                                             \
        T t = threadIdx.x;                                                        \
        T a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7; \
        unsigned long long b64 = ((unsigned long long)b << 32) | c;               \
        asm volatile("s_mov_b64 vcc, %0" :: "s"(sm) : "vcc");                     \
        for (int i = 0; i < ITER; i++) { BODY8(INS) BODY8(INS) }                  \
        T x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                              \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)((unsigned long long)x >> 32); \
    }

#define I_ADD_E32(r) "v_add_u32_e32 " #r ", " #r ", %8"
#define I_SUB_E32(r) "v_sub_u32_e32 " #r ", " #r ", %8"
#define I_CMP_LT_U64(r) "v_cmp_lt_u64_e32 vcc, " #r ", %11"
#define I_CMP_LT_U64_S(r) "v_cmp_lt_u64_e64 s[20:21], " #r ", %11"
#define I_CMP_LT_U32(r) "v_cmp_lt_u32_e32 vcc, " #r ", %8"
#define I_CNDMASK_VCC(r) "v_cndmask_b32_e32 " #r ", " #r ", %8, vcc"
#define I_CNDMASK_S(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, %10"
#define I_CNDMASK_C(r) "v_cndmask_b32_e64 " #r ", 0, -1, %10"
#define I_SUBCO(r) "v_sub_co_u32_e32 " #r ", vcc, " #r ", %8"
#define I_SUBB(r) "v_subb_co_u32_e32 " #r ", vcc, " #r ", %8, vcc"
#define I_SUBBREV(r) "v_subbrev_co_u32_e32 " #r ", vcc, 0, " #r ", vcc"
#define I_MOV64(r) "v_mov_b64_e32 " #r ", %11"
#define I_LSHR64(r) "v_lshrrev_b64 " #r ", 7, " #r
#define I_LSHL64(r) "v_lshlrev_b64 " #r ", 7, " #r
#define I_MOVDPP(r) "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_DOT4(r) "v_dot4_u32_u8 " #r ", " #r ", %8, %9"
#define I_DOT2(r) "v_dot2_u32_u16 " #r ", " #r ", %8, %9"
#define I_PKMAD16(r) "v_pk_mad_u16 " #r ", " #r ", %8, %9"
#define I_PKMUL16(r) "v_pk_mul_lo_u16 " #r ", " #r ", %8"
#define I_MAD64_NOCARRY(r) "v_mad_u64_u32 " #r ", s[20:21], %8, %9, " #r
#define I_MADI64(r) "v_mad_i64_i32 " #r ", s[20:21], %8, %9, " #r
#define I_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, 16"
#define I_AND_E32(r) "v_and_b32_e32 " #r ", " #r ", %8"
#define I_NOT(r) "v_not_b32_e32 " #r ", " #r
#define I_LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 0, %11"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8"
// realistic pairs: the mask is written by the VALU instruction right before the select (hazard wait states spelled out)
#define I_PAIR_VCC(r) "v_cmp_lt_u32_e32 vcc, " #r ", %8\n s_nop 1\n v_cndmask_b32_e32 " #r ", " #r ", %9, vcc"
#define I_PAIR_SGPR(r) "v_cmp_lt_u32_e64 s[20:21], " #r ", %8\n s_nop 1\n v_cndmask_b32_e64 " #r ", " #r ", %9, s[20:21]"
#define I_PAIR_VCC_E64(r) "v_cmp_lt_u32_e32 vcc, " #r ", %8\n s_nop 1\n v_cndmask_b32_e64 " #r ", 0, -1, vcc"
#define I_CNDMASK_VCC_E64(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, vcc"

KERNEL(k_add, I_ADD_E32, uint32_t) KERNEL(k_sub, I_SUB_E32, uint32_t) KERNEL(k_cmp64, I_CMP_LT_U64, uint64_t) KERNEL(k_cmp64s, I_CMP_LT_U64_S, uint64_t)
KERNEL(k_cmp32, I_CMP_LT_U32, uint32_t) KERNEL(k_cnd_vcc, I_CNDMASK_VCC, uint32_t) KERNEL(k_cnd_s, I_CNDMASK_S, uint32_t) KERNEL(k_cnd_c, I_CNDMASK_C, uint32_t)
KERNEL(k_subco, I_SUBCO, uint32_t) KERNEL(k_subb, I_SUBB, uint32_t) KERNEL(k_subbrev, I_SUBBREV, uint32_t) KERNEL(k_mov64, I_MOV64, uint64_t)
KERNEL(k_lshr64, I_LSHR64, uint64_t) KERNEL(k_lshl64, I_LSHL64, uint64_t) KERNEL(k_movdpp, I_MOVDPP, uint32_t) KERNEL(k_dot4, I_DOT4, uint32_t)
KERNEL(k_dot2, I_DOT2, uint32_t) KERNEL(k_pkmad16, I_PKMAD16, uint32_t) KERNEL(k_pkmul16, I_PKMUL16, uint32_t) KERNEL(k_mad64, I_MAD64_NOCARRY, uint64_t)
KERNEL(k_madi64, I_MADI64, uint64_t) KERNEL(k_alignbit, I_ALIGNBIT, uint32_t) KERNEL(k_and, I_AND_E32, uint32_t) KERNEL(k_not, I_NOT, uint32_t)
KERNEL(k_lshladd64, I_LSHLADD64, uint64_t) KERNEL(k_mulhi, I_MULHI, uint32_t)
KERNEL(k_pair_vcc, I_PAIR_VCC, uint32_t) KERNEL(k_pair_sgpr, I_PAIR_SGPR, uint32_t) KERNEL(k_pair_vcc64, I_PAIR_VCC_E64, uint32_t) KERNEL(k_cnd_vcc64, I_CNDMASK_VCC_E64, uint32_t)

template <typename K>
static int run(const char *name, K kern, uint32_t *d, unsigned long long *dclk, int cus, int waves_per_simd)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = cus * waves_per_simd; // each block = 4 waves = 1 per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 0x5555AAAA3333CCCCull, dclk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 0x5555AAAA3333CCCCull, dclk);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts = (double)blocks * 4 * ITER * 16;
    const double per_s = wave_insts / (ms * 1e-3);
    printf("%-28s w/SIMD=%d %8.3f ms %8.1f G winst/s  -> %.2f clk/wave-instr/SIMD @2.4GHz\n", name, waves_per_simd, ms, per_s / 1e9,
           1024.0 * 2.4e9 / per_s * (cus / 256.0));
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *d; CHK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
    unsigned long long *dclk; CHK(hipMalloc(&dclk, 16));
    for (int w : {8, 4}) {
        run("v_add_u32", k_add, d, dclk, cus, w); run("v_sub_u32", k_sub, d, dclk, cus, w); run("v_and_b32", k_and, d, dclk, cus, w);
        run("v_not_b32", k_not, d, dclk, cus, w);
        run("v_cmp_lt_u64 vcc", k_cmp64, d, dclk, cus, w); run("v_cmp_lt_u64 sgpr", k_cmp64s, d, dclk, cus, w); run("v_cmp_lt_u32 vcc", k_cmp32, d, dclk, cus, w);
        run("v_cndmask vcc(quiet)", k_cnd_vcc, d, dclk, cus, w); run("v_cndmask sgpr", k_cnd_s, d, dclk, cus, w);
        run("v_cndmask 0,-1,sgpr", k_cnd_c, d, dclk, cus, w); run("v_cndmask_e64 vcc(quiet)", k_cnd_vcc64, d, dclk, cus, w);
        run("PAIR cmp->vcc,cndmask e32", k_pair_vcc, d, dclk, cus, w); run("PAIR cmp->sgpr,cndmask e64", k_pair_sgpr, d, dclk, cus, w);
        run("PAIR cmp->vcc,cndmask e64", k_pair_vcc64, d, dclk, cus, w);
        run("v_sub_co_u32", k_subco, d, dclk, cus, w); run("v_subb_co_u32", k_subb, d, dclk, cus, w); run("v_subbrev_co_u32", k_subbrev, d, dclk, cus, w);
        run("v_mov_b64", k_mov64, d, dclk, cus, w); run("v_lshrrev_b64", k_lshr64, d, dclk, cus, w); run("v_lshlrev_b64", k_lshl64, d, dclk, cus, w);
        run("v_lshl_add_u64", k_lshladd64, d, dclk, cus, w);
        run("v_mov_b32_dpp", k_movdpp, d, dclk, cus, w); run("v_alignbit_b32", k_alignbit, d, dclk, cus, w); run("v_mul_hi_u32", k_mulhi, d, dclk, cus, w);
        run("v_mad_u64_u32 sgpr-carry", k_mad64, d, dclk, cus, w); run("v_mad_i64_i32", k_madi64, d, dclk, cus, w);
        run("v_dot4_u32_u8", k_dot4, d, dclk, cus, w); run("v_dot2_u32_u16", k_dot2, d, dclk, cus, w);
        run("v_pk_mad_u16", k_pkmad16, d, dclk, cus, w); run("v_pk_mul_lo_u16", k_pkmul16, d, dclk, cus, w);
        printf("\n");
    }
    return 0;
}
