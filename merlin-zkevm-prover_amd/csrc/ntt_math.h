// ntt_math.h -- in-register small DFTs over Goldilocks used by the Stockham NTT passes and the FRI fold.
//
// In Goldilocks 2 has multiplicative order 192 (2^96 = -1), and the reference's root table
// (w(4) = 4096 = 2^12, w(5) = 64, w(6) = 8) makes every twiddle of a 64-point DFT a
// power of two: w_64^j = 2^(3 j).  A butterfly inside a radix-16/64 step therefore costs shifts and adds
// only -- no 64x64 multiply, no table.
#pragma once
#include <type_traits>
#include <utility>
#include "gl_math.h"

#ifndef MI_DFT_CANON
#define MI_DFT_CANON(x) gl::canon(x)
#endif
// Diagnosis builds (tools/pmc_ntt_classes.sh, `make ab-ntt-classes`): ONE class of the butterflies' arithmetic compiled out -- results are
// wrong, the data flow stays -- so that the difference in SQ_INSTS_VALU against the shipped build is that class's instruction count:
//   MI_NTT_AB_NOCANON   no canonical form (the butterflies' v operand, the stores)        MI_NTT_AB_NOADDSUB  u + v, u - v become u ^ v
//   MI_NTT_AB_NOPOW2    the shift twiddles 2^e become 1                                   MI_NTT_AB_NOMULW    twiddle multiplies become xor
#ifdef MI_NTT_AB_NOCANON
#undef MI_DFT_CANON
#define MI_DFT_CANON(x) (x)
#endif
#ifdef MI_NTT_AB_NOADDSUB
#define MI_DFT_ADD(u, v) ((u) ^ (v))
#define MI_DFT_SUB(u, v) ((u) ^ (v) ^ 1)
#else
#define MI_DFT_ADD(u, v) gl::add_wc((u), (v))
#define MI_DFT_SUB(u, v) gl::sub_wc((u), (v))
#endif

namespace nttm {

template <int I, int N, typename F>
MI_HD void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// x * 2^E mod p for a compile-time 0 <= E < 192; any u64 in, weakly reduced out.
template <int E>
MI_HD u64 mul_pow2(u64 x)
{
    static_assert(E >= 0 && E < 192, "exponent out of range");
#ifdef MI_NTT_AB_NOPOW2
    return x;
#endif
    if constexpr (E == 0) {
        return x;
    } else if constexpr (E >= 96) {
        return mul_pow2<E - 96>(gl::neg_w(x)); // 2^96 = -1
    } else if constexpr (E >= 64) {
        return mul_pow2<E - 48>(mul_pow2<48>(x));
    } else if constexpr (E < 32) {
        // x * 2^E = lo + hi * 2^64 with hi < 2^31, and 2^64 = 2^32 - 1
        return gl::add_mul_eps(x << E, (u32)(x >> (64 - E)));
    } else {
        return gl::reduce128_w(x << E, x >> (64 - E));
    }
}

template <int Q>
MI_HD constexpr int bitrev(int k)
{
    int r = 0;
    for (int i = 0; i < Q; i++) r |= ((k >> i) & 1) << (Q - 1 - i);
    return r;
}

// 2^Q-point DFT (Q <= 6) in registers, natural order in and out.  Inputs: any u64 encodings; outputs weakly
// reduced (callers canonicalise at the store).  DIF radix-2 layers with shift-only twiddles; the closing
// bit-reversal is a compile-time register renaming.
template <int Q, bool INV>
MI_HD void dft_reg(u64 (&x)[1 << Q])
{
    constexpr int N = 1 << Q;
#ifdef MI_NTT_NO_ARITH // diagnosis build (profiles/r03_pmc_ntt.txt): every load, LDS round trip and store, no butterflies
    return;
#endif
    static_for<0, Q>([&](auto L) {
        constexpr int len = N >> decltype(L)::value;
        constexpr int half = len / 2;
        static_for<0, N / len>([&](auto B) {
            constexpr int s = decltype(B)::value * len;
            static_for<0, half>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int e0 = 3 * j * (64 / len);              // w_len^j = w_64^(j * 64/len) = 2^e0
                constexpr int e = INV ? (192 - e0) % 192 : e0;
                const u64 u = x[s + j], v = MI_DFT_CANON(x[s + j + half]);
                x[s + j] = MI_DFT_ADD(u, v);
                if constexpr (e >= 96) {
                    // (u - v) * 2^e with 2^96 = -1: (v - u) * 2^(e - 96).  Every nontrivial twiddle of the INVERSE transform is of this kind
                    // (e = 192 - e0); taking the difference the other way round costs the canonical form of u (one compare on the common
                    // path) instead of a negation of the difference (canonical form + a 64-bit subtraction): profiles/r05_ntt_valu_breakdown.txt
                    x[s + j + half] = mul_pow2<e - 96>(MI_DFT_SUB(v, MI_DFT_CANON(u)));
                } else {
                    x[s + j + half] = mul_pow2<e>(MI_DFT_SUB(u, v));
                }
            });
        });
    });
    u64 y[N];
#pragma unroll
    for (int k = 0; k < N; k++) y[k] = x[bitrev<Q>(k)];
#pragma unroll
    for (int k = 0; k < N; k++) x[k] = y[k];
}

} // namespace nttm
