// chelpers_acc.h -- lazy accumulation of 64x64-bit products for the compiled constraint evaluators (chelpers_native.hip).
// Compiled into the library (host debug executor) and, as text, into every generated kernel.
//
// A Horner chain  y <- y * C + v  over a challenge C is evaluated as  y_0 * C^m + sum_j v_j * C^(m-j): every term is a product
// with a constant of the running proof, the products are summed unreduced and reduced ONCE per chain piece (2^64 = 2^32 - 1,
// 2^96 = -1, 2^128 = -2^32 mod p), instead of one extension multiplication (six base multiplications, thirteen additions, all
// reduced) per step.
#pragma once
#ifndef __HIPCC_RTC__
#include "gl_math.h"
#endif

namespace chpa {

// A sum of 64x64-bit products kept as three 64-bit limb accumulators and their carry counts:
//   value = a0 + 2^32 a1 + 2^64 a2 + 2^64 c0 + 2^96 c1 + 2^128 c2
// A multiply-accumulate is then four v_mad_u64_u32 (x0 w0 -> a0, x0 w1 and x1 w0 -> a1, x1 w1 -> a2, each adding into its limb
// accumulator) and four v_addc that count the carries out of the limbs: 8 VALU instructions, no cross-limb carry chain, nothing
// to move between register pairs.  The compiler cannot express this (it will not use the carry-out of v_mad_u64_u32), hence asm.
struct Acc { u64 a0, a1, a2; u32 c0, c1, c2; };

MI_HD void acc_set(Acc &A, u64 v) { A.a0 = v; A.a1 = 0; A.a2 = 0; A.c0 = 0; A.c1 = 0; A.c2 = 0; }
MI_HD void acc_add(Acc &A, u64 v)
{
    A.a0 += v;
    A.c0 += A.a0 < v;
}

#if defined(__HIP_DEVICE_COMPILE__)
// VALU writes a carry mask to an SGPR pair -> a VALU instruction reads it as carry-in: two wait states on gfx950; in the order
// below three instructions lie between every v_mad and the v_addc that consumes its carry, so no s_nop is needed
#define CHPA_MAC_ASM(WC)                                                                                                        \
    u64 s0, s1, s2, s3;                                                                                                         \
    asm("v_mad_u64_u32 %[a1], %[s1], %[x0], %[w1], %[a1]\n\t"                                                                   \
        "v_mad_u64_u32 %[a0], %[s0], %[x0], %[w0], %[a0]\n\t"                                                                   \
        "v_mad_u64_u32 %[a1], %[s3], %[x1], %[w0], %[a1]\n\t"                                                                   \
        "v_mad_u64_u32 %[a2], %[s2], %[x1], %[w1], %[a2]\n\t"                                                                   \
        "v_addc_co_u32_e64 %[c1], %[s1], %[c1], 0, %[s1]\n\t"                                                                   \
        "v_addc_co_u32_e64 %[c0], %[s0], %[c0], 0, %[s0]\n\t"                                                                   \
        "v_addc_co_u32_e64 %[c1], %[s3], %[c1], 0, %[s3]\n\t"                                                                   \
        "v_addc_co_u32_e64 %[c2], %[s2], %[c2], 0, %[s2]"                                                                       \
        : [a0] "+v"(A.a0), [a1] "+v"(A.a1), [a2] "+v"(A.a2), [c0] "+v"(A.c0), [c1] "+v"(A.c1), [c2] "+v"(A.c2), [s0] "=&s"(s0),  \
          [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)                                                                        \
        : [x0] "v"((u32)x), [x1] "v"((u32)(x >> 32)), [w0] WC((u32)w), [w1] WC((u32)(w >> 32)))
// w a constant of the running proof (an SGPR pair: one scalar operand per instruction fits the constant bus)
MI_HD void acc_mac_s(Acc &A, u64 x, u64 w) { CHPA_MAC_ASM("s"); }
MI_HD void acc_mac(Acc &A, u64 x, u64 w) { CHPA_MAC_ASM("v"); }
#else
MI_HD void acc_mac(Acc &A, u64 x, u64 w)
{
    const u64 x0 = (u32)x, x1 = x >> 32, w0 = (u32)w, w1 = w >> 32;
    u64 p;
    p = x0 * w0; A.a0 += p; A.c0 += A.a0 < p;
    p = x0 * w1; A.a1 += p; A.c1 += A.a1 < p;
    p = x1 * w0; A.a1 += p; A.c1 += A.a1 < p;
    p = x1 * w1; A.a2 += p; A.c2 += A.a2 < p;
}
MI_HD void acc_mac_s(Acc &A, u64 x, u64 w) { acc_mac(A, x, w); }
#endif

// any value of the accumulator -> weakly reduced.  2^128 = -2^32 mod p; the carry counts are far below 2^31
MI_HD u64 acc_reduce(const Acc &A)
{
    typedef unsigned __int128 u128;
    const u128 L = (u128)A.a0 + ((u128)A.a1 << 32);
    const u128 H = (u128)A.a2 + A.c0 + ((u128)A.c1 << 32) + (u64)(L >> 64);
    const u64 top = (u64)(H >> 64) + A.c2;
    return gl::sub_wc(gl::reduce128_w((u64)L, (u64)H), top << 32);
}

// r += (y0 + y1 x + y2 x^2) * (w0 + w1 x + w2 x^2) in F_p[x] / (x^3 - x - 1)   (x^3 = x + 1, x^4 = x^2 + x)
#define CHPA_MUL33(MAC)                                                                        \
    MAC(r0, y0, w0); MAC(r0, y1, w2); MAC(r0, y2, w1);                                         \
    MAC(r1, y0, w1); MAC(r1, y1, w0); MAC(r1, y1, w2); MAC(r1, y2, w1); MAC(r1, y2, w2);       \
    MAC(r2, y0, w2); MAC(r2, y1, w1); MAC(r2, y2, w0); MAC(r2, y2, w2)
MI_HD void acc_mul33_s(Acc &r0, Acc &r1, Acc &r2, u64 y0, u64 y1, u64 y2, u64 w0, u64 w1, u64 w2) { CHPA_MUL33(acc_mac_s); }
MI_HD void acc_mul33(Acc &r0, Acc &r1, Acc &r2, u64 y0, u64 y1, u64 y2, u64 w0, u64 w1, u64 w2) { CHPA_MUL33(acc_mac); }
MI_HD void acc_mul13_s(Acc &r0, Acc &r1, Acc &r2, u64 v, u64 w0, u64 w1, u64 w2)
{
    acc_mac_s(r0, v, w0);
    acc_mac_s(r1, v, w1);
    acc_mac_s(r2, v, w2);
}

} // namespace chpa
