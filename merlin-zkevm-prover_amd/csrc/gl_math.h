// gl_math.h -- Goldilocks field / cubic-extension / Poseidon arithmetic for gfx950 kernels.
//
// p = 2^64 - 2^32 + 1.  Elements travel through HBM as canonical u64 in [0,p) (the reference's
// Goldilocks::Element is a bare uint64_t fe: SURVEY 8(b) "Element representation"); inside a kernel
// values are kept "weakly reduced" in [0,2^64) and canonicalised once, at the store.
//
// CDNA4 has no 64-bit integer multiplier: a 64x64 product is four v_mad_u64_u32, and the special
// form of p turns the 128-bit reduction into adds (2^64 = 2^32-1, 2^96 = -1 mod p) -- no Montgomery
// or Barrett constants.  Everything here is __host__ __device__ so the very same inline code can be
// exercised on a machine without a GPU (mi_dbg_host_* in capi.hip; test use only).
#pragma once
#ifndef __HIPCC_RTC__ // hiprtc (the constraint-program compiler in chelpers.hip) brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

typedef unsigned long long u64;
typedef unsigned int u32;

#define MI_HD __host__ __device__ __forceinline__

static constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
static constexpr u64 GL_EPS = 0xFFFFFFFFULL; // 2^64 mod p = 2^32 - 1

// How a carry flag becomes a 32-bit correction word in the hand-written sequences (A/B switch, see tools/ubench_field.hip):
// 0: carry in VCC + v_cndmask_b32 (VOP2); 1: carry in an SGPR pair + v_cndmask_b32_e64; 2: -carry by v_subb_co_u32
#ifndef MI_MASK_FORM
#define MI_MASK_FORM 0
#endif

namespace gl {

// Where a row sits inside its tile of 64 in a TILE-MAJOR section ([tile][column][64 rows], include/mi_stark.h): at the bit-reversal of
// its index.  The words of a column are one 512-byte run of a tile either way; bit-reversed, the rows that are multiples of 2^e -- the
// rows evmap reads of an extension with blow-up 2^e (starks.cpp:555-668) -- are the FIRST 64 >> e words of the run, whatever e is, and
// not every 2^e-th word of it (half of every 64-byte sector fetched for nothing at blow-up 2, 7/8 at blow-up 8).
MI_HD u32 tile_pos(u32 row_in_tile) { return __builtin_bitreverse32(row_in_tile) >> 26; }

// true in at least one lane of the wave?  A correction that is needed with probability ~2^-32 per value (x >= p,
// lo < hh after a multiply) is put behind a wave-uniform branch: the common path pays one compare instead of a
// compare, two selects and a 64-bit add.  Host build: plain condition.
// MI_NO_RARE_BRANCH (the generated constraint-evaluator kernels): always take the select form -- thousands of wave-uniform
// branches in one straight-line kernel split it into as many basic blocks and wreck register allocation (measured: 512 VGPRs
// and 4 500 spills with the branches, 146 VGPRs without).
MI_HD bool rare(bool c)
{
#if defined(MI_NO_RARE_BRANCH)
    (void)c;
    return true;
#elif defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(c) != 0;
#else
    return c;
#endif
}

// keeps the compiler from turning a rare() block back into selects
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MI_NO_RARE_BRANCH)
#define MI_KEEP_BRANCH() asm volatile("" ::: "memory")
#else
#define MI_KEEP_BRANCH() do { } while (0)
#endif

// a 32-bit value the compiler cannot see through (stays in an SGPR): keeps "x * k + acc" one v_mad_u64_u32 where it
// would otherwise strength-reduce the multiplication into shifts, zero-extending moves and 64-bit adds
MI_HD u32 opaque_u32(u32 k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(k));
#endif
    return k;
}

MI_HD u64 canon_sel(u64 a) { return a >= GL_P ? a - GL_P : a; } // select form (no branch)

MI_HD u64 canon(u64 a)
{
    if (rare(a >= GL_P)) {
        MI_KEEP_BRANCH();
        a = a >= GL_P ? a - GL_P : a;
    }
    return a;
}

// a any u64, b canonical (< p)  ->  weakly reduced a + b
MI_HD u64 add_wc(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__) && MI_MASK_FORM == 1
    u64 s, m;
    u32 e;
    asm("v_lshl_add_u64 %0, %3, 0, %4\n\tv_cmp_lt_u64_e64 %2, %0, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, %2"
        : "=&v"(s), "=v"(e), "=&s"(m) : "v"(a), "v"(b));
    return s + e;
#else
    u64 s = a + b;
    return s + (s < a ? GL_EPS : 0); // wrapped: true sum < 2^64 + p, so s < p and s + eps cannot wrap again
#endif
}

// both canonical -> canonical
MI_HD u64 add(u64 a, u64 b)
{
    u64 s = a + b;
    u64 t = s + GL_EPS; // s - p (mod 2^64)
    return (s < a || s >= GL_P) ? t : s;
}

// both canonical -> canonical
MI_HD u64 sub(u64 a, u64 b)
{
    u64 d = a - b;
    return a < b ? d - GL_EPS : d; // d + p (mod 2^64)
}

MI_HD u64 neg(u64 a) { return a ? GL_P - a : 0; }

// a any u64, b canonical (< p)  ->  weakly reduced a - b
// As compiled: v_sub_co, v_subb_co, v_cmp_lt_u64, 2 x v_cndmask, v_lshl_add_u64 (20 issue clocks).  The compare
// re-derives the chain's own borrow, but every way of using that borrow directly was tried in r02 and is no cheaper:
// __builtin_usubll_overflow is canonicalised back into sub + compare; an inline-asm chain must either finish with a
// second 32-bit carry chain (2 x 4 clk instead of one v_lshl_add_u64) or hand two selected words back to the compiler,
// which then forms the 64-bit addend with extra moves (measured: +8 % instructions in k_ntt_pass).
MI_HD u64 sub_wc(u64 a, u64 b)
{
    u64 d = a - b;
    return d - (a < b ? GL_EPS : 0); // borrowed: d >= 2^64 - p + 1 > eps, so d - eps cannot borrow again
}
// any u64 -> weakly reduced -a
MI_HD u64 neg_w(u64 a) { return GL_P - canon(a); }

// lo + h * (2^32 - 1)  ->  weakly reduced; any lo, any 32-bit h.
// ONE v_mad_u64_u32 whose carry-out drives the wrap correction; the compiler will not use that carry (it splits the
// multiply-add to re-derive it with a 64-bit compare), hence the two-instruction asm.
MI_HD u64 add_mul_eps(u64 lo, u32 h)
{
    u64 r1;
    u32 e; // wrapped ? 2^32 - 1 : 0
#if defined(__HIP_DEVICE_COMPILE__)
    // VALU write of a carry mask -> VALU read of it as a select mask / carry-in: two wait states on gfx950
#if MI_MASK_FORM == 1
    u64 m;
    asm("v_mad_u64_u32 %0, %2, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, %2" : "=v"(r1), "=v"(e), "=&s"(m) : "v"(h), "v"(lo));
#elif MI_MASK_FORM == 2
    u64 m;
    asm("v_mad_u64_u32 %0, vcc, %3, -1, %4\n\ts_nop 1\n\tv_subb_co_u32_e64 %1, %2, 0, 0, vcc" : "=v"(r1), "=v"(e), "=&s"(m) : "v"(h), "v"(lo) : "vcc");
#else
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\ts_nop 1\n\tv_cndmask_b32 %1, 0, -1, vcc" : "=v"(r1), "=v"(e) : "v"(h), "v"(lo) : "vcc");
#endif
#else
    r1 = (u64)h * 0xFFFFFFFFu + lo;
    e = r1 < lo ? 0xFFFFFFFFu : 0;
#endif
    return r1 + e; // the wrapped sum is < h * (2^32-1) <= (2^32-1)^2: adding eps cannot wrap again
}

// 128-bit (hi:lo) -> weakly reduced u64.   x = lo + hl*2^64 + hh*2^96 = lo + hl*(2^32-1) - hh
// The subtraction of hh is spelled as the 32-bit borrow chain so that the rare-borrow branch tests the flag itself.
#ifndef MI_REDUCE_SUBB_ASM
#define MI_REDUCE_SUBB_ASM 0
#endif
MI_HD u64 reduce128_w(u64 lo, u64 hi)
{
    const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    const u64 r2 = add_mul_eps(lo, hl);
#if defined(__HIP_DEVICE_COMPILE__) && MI_REDUCE_SUBB_ASM
    // r2 - hh as the two-instruction borrow chain (the compiler materialises the first borrow with a select and then
    // subtracts it: v_sub_co, v_cndmask, v_sub_co); the final borrow mask goes to an SGPR pair for the rare-case branch
    u32 d0, d1;
    u64 m;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\ts_nop 1\n\tv_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\ts_nop 1\n\ts_mov_b64 %2, vcc"
        : "=&v"(d0), "=&v"(d1), "=s"(m) : "v"((u32)r2), "v"((u32)(r2 >> 32)), "v"(hh) : "vcc");
    u64 t = ((u64)d1 << 32) | d0;
    if (m != 0) { // wave-uniform: some lane borrowed 2^64 = p + eps (needs r2 < 2^32: ~never)
        MI_KEEP_BRANCH();
        t = ((m >> __lane_id()) & 1) ? t - GL_EPS : t;
    }
    return t;
#else
    u32 b1, b2;
    const u32 d0 = __builtin_subc((u32)r2, hh, 0u, &b1);
    const u32 d1 = __builtin_subc((u32)(r2 >> 32), 0u, b1, &b2);
    u64 t = ((u64)d1 << 32) | d0;
    if (rare(b2 != 0)) { // borrowed 2^64 = p + eps (needs r2 < 2^32: ~never)
        MI_KEEP_BRANCH();
        t = b2 ? t - GL_EPS : t; // t >= 2^64 - 2^32 + 1 > eps: no second borrow
    }
    return t;
#endif
}

MI_HD void mul64x64(u64 a, u64 b, u64 &lo, u64 &hi)
{
    // four 32x32->64 multiply-adds (v_mad_u64_u32)
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p10 = (u64)a1 * b0 + (p00 >> 32);          // < 2^64
    u64 p01 = (u64)a0 * b1 + (u32)p10;             // < 2^64
    u64 p11 = (u64)a1 * b1 + (p10 >> 32) + (p01 >> 32);
    lo = (p01 << 32) | (u32)p00;
    hi = p11;
}

MI_HD void sqr64(u64 a, u64 &lo, u64 &hi)
{
    // a^2 = p00 + 2*p01*2^32 + p11*2^64 : three multiplies instead of four
    u32 a0 = (u32)a, a1 = (u32)(a >> 32);
    u64 p00 = (u64)a0 * a0;
    u64 p01 = (u64)a0 * a1;
    u64 p11 = (u64)a1 * a1;
    u64 mid = (p00 >> 32) + ((p01 << 1) & 0xFFFFFFFFULL); // word 1, < 2^33
    lo = (mid << 32) | (u32)p00;
    hi = p11 + (p01 >> 31) + (mid >> 32);
}

// any u64 inputs -> weakly reduced
MI_HD u64 mul_w(u64 a, u64 b)
{
    u64 lo, hi;
    mul64x64(a, b, lo, hi);
    return reduce128_w(lo, hi);
}
MI_HD u64 sqr_w(u64 a)
{
    u64 lo, hi;
    sqr64(a, lo, hi);
    return reduce128_w(lo, hi);
}
// canonical result
MI_HD u64 mul(u64 a, u64 b) { return canon(mul_w(a, b)); }

MI_HD u64 pow(u64 a, u64 e)
{
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul_w(r, a);
        a = sqr_w(a);
        e >>= 1;
    }
    return canon(r);
}
MI_HD u64 inv(u64 a) { return pow(a, GL_P - 2); } // inv(0) = 0

// ---- cubic extension F_p[x]/(x^3 - x - 1); reference form: polinomial.hpp:195-205
struct E3 { u64 v[3]; };

MI_HD E3 e3_add(const E3 &a, const E3 &b) { return {{add(a.v[0], b.v[0]), add(a.v[1], b.v[1]), add(a.v[2], b.v[2])}}; }
MI_HD E3 e3_sub(const E3 &a, const E3 &b) { return {{sub(a.v[0], b.v[0]), sub(a.v[1], b.v[1]), sub(a.v[2], b.v[2])}}; }
MI_HD E3 e3_mul1(const E3 &a, u64 b) { return {{mul(a.v[0], b), mul(a.v[1], b), mul(a.v[2], b)}}; }
MI_HD E3 e3_mul(const E3 &a, const E3 &b)
{
    u64 A = mul(add(a.v[0], a.v[1]), add(b.v[0], b.v[1]));
    u64 B = mul(add(a.v[0], a.v[2]), add(b.v[0], b.v[2]));
    u64 C = mul(add(a.v[1], a.v[2]), add(b.v[1], b.v[2]));
    u64 D = mul(a.v[0], b.v[0]);
    u64 E = mul(a.v[1], b.v[1]);
    u64 F = mul(a.v[2], b.v[2]);
    u64 G = sub(D, E);
    return {{sub(add(C, G), F), sub(sub(sub(add(A, C), E), E), D), sub(B, G)}};
}
// exact inverse via the adjugate of the multiplication-by-a matrix (inv(0) = 0)
MI_HD E3 e3_inv(const E3 &a)
{
    // columns a*1, a*x, a*x^2  (x^3 = x + 1)
    u64 a0 = a.v[0], a1 = a.v[1], a2 = a.v[2];
    u64 m00 = a0, m10 = a1, m20 = a2;                       // a*1
    u64 m01 = a2, m11 = add(a0, a2), m21 = a1;              // a*x   = a2 + (a0+a2) x + a1 x^2
    u64 m02 = a1, m12 = add(a1, a2), m22 = add(a0, a2);     // a*x^2 = a1 + (a1+a2) x + (a0+a2) x^2
    u64 c00 = sub(mul(m11, m22), mul(m12, m21));
    u64 c01 = sub(mul(m10, m22), mul(m12, m20));
    u64 c02 = sub(mul(m10, m21), mul(m11, m20));
    u64 det = add(sub(mul(m00, c00), mul(m01, c01)), mul(m02, c02));
    u64 di = inv(det);
    return {{mul(c00, di), mul(neg(c01), di), mul(c02, di)}};
}

} // namespace gl
