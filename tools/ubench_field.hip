// ubench_field.hip -- register-only timings of the field arithmetic the two hot kernels are made of, as compiled from the
// product headers (gl_math.h / ntt_math.h / poseidon_math.h): an in-register radix-16 DFT, the general multiply, one full
// Poseidon round, the whole permutation -- and, for the int8-MFMA question, the VALU cost of recombining the
// fifteen i32 byte-position sums an MFMA formulation of a constant-matrix product would hand back per output.
// Build twice to A/B an arithmetic variant:  hipcc -O3 --offload-arch=gfx950 [-DMI_REDUCE_SUBB_ASM=1] -o ubench_field ubench_field.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../merlin-zkevm-prover_amd/csrc/ntt_math.h"
#include "../merlin-zkevm-prover_amd/csrc/poseidon_math.h"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__constant__ u64 c_rc[360];
__constant__ pos::SparseTables c_sparse;

__device__ __forceinline__ u64 seed(uint32_t i) { return (u64)(threadIdx.x + 64 * blockIdx.x + 1) * 0x9E3779B97F4A7C15ULL + (u64)i * 0xBF58476D1CE4E5B9ULL; }

template <bool INV>
__global__ __launch_bounds__(256) void k_dft16(u64 *out, int iters)
{
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = seed(i);
#pragma unroll 1
    for (int it = 0; it < iters; it++) nttm::dft_reg<4, INV>(x);
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) a ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_mulw(u64 *out, int iters)
{
    u64 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = seed(i);
    const u64 w = seed(99);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = gl::mul_w(x[i], w);
    }
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_addsub(u64 *out, int iters)
{
    u64 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = seed(i);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) { // one butterfly without twiddle
            const u64 u = x[i], v = gl::canon(x[i + 1]);
            x[i] = gl::add_wc(u, v);
            x[i + 1] = gl::sub_wc(u, v);
        }
    }
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_full_round(u64 *out, int iters)
{
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed(i);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = pos::sbox(s[i]);
        pos::mds_half32(s, c_rc + 12 * (it & 15));
    }
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) a ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <int MDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_permute_loop(u64 *out, int iters)
{
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed(i);
#pragma unroll 1
    for (int it = 0; it < iters; it++) pos::permute<MDS, 0>(s, c_rc, &c_sparse);
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) a ^= s[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

// What an int8-MFMA formulation of an 11 x 11 constant-matrix product leaves to the vector ALU per output and state: fifteen
// signed 32-bit sums T[t] (byte position t of the product), to be recombined as sum_t T[t] * 2^(8 t) mod p.  Cheapest
// exact form found: even and odd positions into two signed 128-bit accumulators by shifts of 16 bits ... here as 4 64-bit
// lanes of Horner steps (shift by 8 = one v_lshl_add_u64 with a sign-extended addend), then one 128-bit reduction.
__device__ __forceinline__ u64 recombine15(const int (&T)[15])
{
    // value = sum T[t] 2^(8t), |T| < 2^21.  Split positions 0..7 -> L (< 2^77 magnitude), 8..14 -> H, each by Horner
    // on signed 128-bit (hi:lo) pairs; then L + H * 2^64 reduced.
    long long lo = 0, hi = 0; // positions 8..14
#pragma unroll
    for (int t = 14; t >= 8; t--) hi = (hi << 8) + T[t];      // < 2^(21 + 48 + 1): fits i64
#pragma unroll
    for (int t = 7; t >= 0; t--) lo = (lo << 8) + T[t];       // < 2^(21 + 56 + 1) -- overflows i64 by design of the estimate: a real
                                                              // implementation needs one more limb here (cost not counted)
    // value = lo + hi * 2^64 = lo + hi * eps (mod p); signs folded in by adding p multiples (not counted either)
    const u64 ulo = (u64)lo, uhi = (u64)hi;
    u64 l, h;
    gl::mul64x64(uhi, GL_EPS, l, h);
    return gl::add_wc(gl::reduce128_w(l, h), gl::canon(ulo));
}

__global__ __launch_bounds__(256) void k_recombine(u64 *out, int iters)
{
    int T[15];
#pragma unroll
    for (int t = 0; t < 15; t++) T[t] = (int)(seed(t) >> 43) - (1 << 20);
    u64 acc = 0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 11; i++) { // eleven outputs of one 11 x 11 product
            acc ^= recombine15(T);
#pragma unroll
            for (int t = 0; t < 15; t++) T[t] += (int)(acc >> (t + 3)) & 1023;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// the 121 dot-product terms (6 v_mad_u64_u32 each) + 11 closings that recombine15 x 11 would replace
__global__ __launch_bounds__(256) void k_dot11x11(u64 *out, int iters)
{
    u64 z[11];
#pragma unroll
    for (int i = 0; i < 11; i++) z[i] = seed(i);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        u64 o[11];
#pragma unroll
        for (int i = 0; i < 11; i++) {
            pos::DotAcc d = {};
#pragma unroll
            for (int j = 0; j < 11; j++) pos::dot_acc(d, z[j], c_sparse.g[0].pre[i][j]);
            o[i] = pos::dot_close(d);
        }
#pragma unroll
        for (int i = 0; i < 11; i++) z[i] = o[i];
    }
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) a ^= z[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <typename K>
static int run(const char *name, K kern, u64 *d, int cus, int waves_per_simd, int iters, double units_per_iter, const char *unit)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = cus * waves_per_simd; // 256 threads = 4 waves = one per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // SIMD-clocks per unit per wave at 2.4 GHz: time * clock / (iters * units * waves per SIMD)
    const double clk = ms * 1e-3 * 2.4e9 / ((double)iters * units_per_iter * waves_per_simd);
    printf("%-34s w/SIMD=%d %8.3f ms  %8.1f SIMD-clk per %s per wave (2.4 GHz)   %.3f G %s/s chip-wide (lane level)\n", name,
           waves_per_simd, ms, clk, unit, (double)blocks * 256 * iters * units_per_iter / (ms * 1e-3) / 1e9, unit);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u64 *d; CHK(hipMalloc(&d, (size_t)cus * 8 * 256 * 8));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_rc), MI_POS_RC, sizeof(MI_POS_RC)));
    static pos::SparseTables t;
    pos::fill_sparse_tables(t);
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_sparse), &t, sizeof(t)));
#ifdef MI_REDUCE_SUBB_ASM
    printf("variant: MI_REDUCE_SUBB_ASM=%d\n", MI_REDUCE_SUBB_ASM);
#endif
    for (int w : {4, 3, 8}) {
        run("butterfly add_wc+sub_wc+canon", k_addsub, d, cus, w, 4096, 4, "butterfly");
        run("dft16 fwd (32 butterflies)", k_dft16<false>, d, cus, w, 512, 1, "dft16");
        run("dft16 inv", k_dft16<true>, d, cus, w, 512, 1, "dft16");
        run("mul_w", k_mulw, d, cus, w, 2048, 8, "mul");
        run("full round (12 sbox + mds)", k_full_round, d, cus, w, 512, 1, "round");
        run("11x11 dot products (121 terms)", k_dot11x11, d, cus, w, 256, 1, "matvec");
        run("mfma recombination x11 (lower bound)", k_recombine, d, cus, w, 256, 1, "matvec");
        if (w == 3) {
            run("permute variant 2", k_permute_loop<pos::MDS_SPARSE>, d, cus, w, 64, 1, "perm");
            run("permute variant 0", k_permute_loop<pos::MDS_HALF32>, d, cus, w, 64, 1, "perm");
        }
        printf("\n");
    }
    return 0;
}
