// poseidon_math.h -- Poseidon-over-Goldilocks permutation (t=12, R_F=8, R_P=22, x^7), one state
// per lane, all 12 words in VGPRs.
//
// Spec followed: src/sm/poseidon_g/poseidon_g_executor.cpp:174-205 (naive round form: add RC, S-box on
// all lanes in rounds 0-3/26-29 else lane 0, state = M*state) with M = circ(MCIRC) + diag(MDIAG)
// (poseidon_g_executor.hpp:37-50).  The upstream library uses the algebraically equal "optimised partial
// round" form; on CDNA4 the naive form is the cheaper one: its MDS constants are < 2^6, so the
// mat-vec needs no 64x64 multiplies at all (see mds_limb22), whereas the optimised form trades it for
// 23 full 64-bit modular multiplies per partial round.
//
// Why one state per lane and not a wave-cooperative round with the state spread over 12 lanes: 22 of
// the 30 rounds apply the S-box to lane 0 only, which would idle 11 of 12 cooperating lanes for ~2/3 of
// the multiplies; with a whole state per lane every VALU slot does useful work and no LDS / cross-lane
// traffic is needed.  (MFMA: not applicable -- exact 64-bit modular accumulation, 12x12 only.)
#pragma once
#include "gl_math.h"
#include "poseidon_constants.h"
#include "poseidon_sparse_constants.h"

namespace pos {

static constexpr int MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20}; // == MI_POS_MCIRC
static constexpr int MD0 = 8;                                                   // == MI_POS_MDIAG[0]

enum { MDS_HALF32 = 0, MDS_LIMB22 = 1, MDS_SPARSE = 2 };

MI_HD u64 sbox(u64 x)
{
    u64 x2 = gl::sqr_w(x);
    u64 x4 = gl::sqr_w(x2);
    u64 x3 = gl::mul_w(x, x2);
    return gl::mul_w(x3, x4);
}

// MDS on 32-bit halves: two 64-bit accumulators per output (v_mad_u64_u32).  Row sum of M is 264, so each
// accumulator stays below 2^41 + 2^32.  The NEXT round's constants are folded in for free: the accumulators start
// at the two halves of rc_next[x] instead of zero (exact integer identity), which removes the 12 modular additions
// of the following "add round constants" layer.  rc_next == nullptr: plain M * s.
MI_HD void mds_half32(u64 (&s)[12], const u64 *__restrict__ rc_next = nullptr)
{
    u32 lo[12], hi[12];
    u64 rcn[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); rcn[i] = rc_next ? rc_next[i] : 0; }
#pragma unroll
    for (int x = 0; x < 12; x++) {
        u64 al = (u64)(u32)rcn[x], ah = rcn[x] >> 32;
#pragma unroll
        for (int y = 0; y < 12; y++) {
            const u32 m = (u32)(MC[(y - x + 12) % 12] + ((x == 0 && y == 0) ? MD0 : 0));
            al += (u64)lo[y] * m;
            ah += (u64)hi[y] * m;
        }
        // value = al + ah*2^32 = al + (ah_lo << 32) + ah_hi * 2^64 ;  2^64 = 2^32 - 1
        u64 ahh = ah >> 32;                          // < 2^10
        u64 r = al + ((ahh << 32) - ahh);            // < 2^43, no wrap
        u64 b = ah << 32;
        u64 t = r + b;
        s[x] = t < b ? t + GL_EPS : t; // (a wave-uniform branch here splits the block and un-hoists the constant loads)
    }
}

// MDS on 22/22/20-bit limbs: three 32-bit accumulators per output, full-rate v_mad_u32_u24.
// acc_k <= 264 * (2^22 - 1) < 2^31.
MI_HD u32 mad24(u32 a, u32 b, u32 c)
{
    // both factors are provably < 2^24 (a is masked, b is a literal < 2^6): the AMDGPU backend selects
    // the full-rate v_mad_u32_u24 for this expression (checked in the ISA)
    return a * b + c;
}

MI_HD void mds_limb22(u64 (&s)[12])
{
    u32 l0[12], l1[12], l2[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        l0[i] = (u32)s[i] & 0x3FFFFFu;
        l1[i] = (u32)(s[i] >> 22) & 0x3FFFFFu;
        l2[i] = (u32)(s[i] >> 44);
    }
#pragma unroll
    for (int x = 0; x < 12; x++) {
        u32 a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int y = 0; y < 12; y++) {
            const u32 m = (u32)(MC[(y - x + 12) % 12] + ((x == 0 && y == 0) ? MD0 : 0));
            a0 = mad24(l0[y], m, a0);
            a1 = mad24(l1[y], m, a1);
            a2 = mad24(l2[y], m, a2);
        }
        // value = a0 + a1*2^22 + a2*2^44  (< 2^74)
        u64 v = (u64)a0 + ((u64)a1 << 22);           // < 2^54
        u64 b = (u64)a2 << 44;                       // low 20 bits of a2 land in bits 44..63
        u64 lo = v + b;
        u64 hi = (u64)(a2 >> 20) + (lo < b ? 1 : 0); // < 2^10, multiples of 2^64
        u64 t1 = (hi << 32) - hi;
        u64 r = lo + t1;
        s[x] = r < t1 ? r + GL_EPS : r;
    }
}

// ---- optimised partial rounds (variant MDS_SPARSE): tables derived and verified by tools/gen_poseidon_sparse.py.
// A partial round becomes: S-box on s0, one 12-term dot product with 64-bit constants for the new s0 and eleven
// "s_i += w_i * s0".  The dot product is accumulated WITHOUT carries: each constant is pre-split into 22/22/20-bit
// limbs, so the six 64-bit accumulators (2 halves of s_j x 3 limbs) stay below 2^58 over 12 terms and a single
// reduction closes the row.
struct SparseTables {
    u64 first_rc[12];
    u64 k[22];            // k[21] unused (0)
    u64 w[22][11];
    u32 vhat_l[22][11][3];
    u32 pre_l[11][11][3];
};

inline void fill_sparse_tables(SparseTables &t)
{
    for (int i = 0; i < 12; i++) t.first_rc[i] = MI_POS_FIRST_RC[i];
    for (int r = 0; r < 22; r++) {
        t.k[r] = r < 21 ? MI_POS_K[r] : 0;
        for (int j = 0; j < 11; j++) {
            t.w[r][j] = MI_POS_W[r * 11 + j];
            const u64 v = MI_POS_VHAT[r * 11 + j];
            t.vhat_l[r][j][0] = (u32)(v & 0x3FFFFF); t.vhat_l[r][j][1] = (u32)((v >> 22) & 0x3FFFFF); t.vhat_l[r][j][2] = (u32)(v >> 44);
        }
    }
    for (int i = 0; i < 11; i++)
        for (int j = 0; j < 11; j++) {
            const u64 v = MI_POS_PRE[i * 11 + j];
            t.pre_l[i][j][0] = (u32)(v & 0x3FFFFF); t.pre_l[i][j][1] = (u32)((v >> 22) & 0x3FFFFF); t.pre_l[i][j][2] = (u32)(v >> 44);
        }
}

struct DotAcc { u64 a[2][3]; };

MI_HD void dot_acc(DotAcc &d, u64 x, const u32 (&c)[3])
{
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
#pragma unroll
    for (int l = 0; l < 3; l++) {
        d.a[0][l] += (u64)lo * c[l];
        d.a[1][l] += (u64)hi * c[l];
    }
}

// value = sum_{h,l} a[h][l] * 2^(32h + 22l), every a < 2^60  ->  weakly reduced
MI_HD u64 dot_close(const DotAcc &d)
{
    typedef unsigned __int128 u128;
    const u128 X = (u128)d.a[0][0] + ((u128)d.a[0][1] << 22) + ((u128)d.a[0][2] << 44); // < 2^105
    const u128 Y = (u128)d.a[1][0] + ((u128)d.a[1][1] << 22) + ((u128)d.a[1][2] << 44);
    const u64 ylo = (u64)Y, yhi = (u64)(Y >> 64);                                        // yhi < 2^41
    // Y * 2^32 = (ylo mod 2^32) * 2^32 + (ylo >> 32) * 2^64 + yhi * 2^96  ==  ... + (ylo >> 32) * eps - yhi
    const u128 T = (u128)(u64)X + (u128)(u64)(X >> 64) * GL_EPS + ((u128)(ylo & GL_EPS) << 32) + (u128)(ylo >> 32) * GL_EPS;
    const u64 r = gl::reduce128_w((u64)T, (u64)(T >> 64));
    return gl::sub_wc(r, yhi);
}

// s_i + w * s0 (all 64-bit, any encodings) -> weakly reduced
MI_HD u64 axpy_w(u64 si, u64 w, u64 s0)
{
    u64 lo, hi;
    gl::mul64x64(w, s0, lo, hi);
    const u64 l2 = lo + si;
    hi += l2 < lo ? 1 : 0; // (2^64-1)^2 + 2^64 - 1 < 2^128: no overflow of hi
    return gl::reduce128_w(l2, hi);
}

MI_HD void partial_rounds_sparse(u64 (&s)[12], const SparseTables &t)
{
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_wc(s[i], t.first_rc[i]);
    { // s[1:] = PRE * s[1:]
        u64 o[11];
#pragma unroll 1
        for (int i = 0; i < 11; i++) {
            DotAcc d = {};
#pragma unroll
            for (int j = 0; j < 11; j++) dot_acc(d, s[1 + j], t.pre_l[i][j]);
            o[i] = dot_close(d);
        }
#pragma unroll
        for (int i = 0; i < 11; i++) s[1 + i] = o[i];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        u64 s0 = gl::add_wc(sbox(s[0]), t.k[r]);
        DotAcc d = {};
        d.a[0][0] = (u64)(u32)s0 * MI_POS_M00;
        d.a[1][0] = (u64)(u32)(s0 >> 32) * MI_POS_M00;
#pragma unroll
        for (int j = 0; j < 11; j++) dot_acc(d, s[1 + j], t.vhat_l[r][j]);
#pragma unroll
        for (int i = 0; i < 11; i++) s[1 + i] = axpy_w(s[1 + i], t.w[r][i], s0);
        s[0] = dot_close(d);
    }
}

// s <- M * s + rc_next (rc_next may be null).  Only the half32 path folds the constants; the others add them.
template <int MDS>
MI_HD void mds(u64 (&s)[12], const u64 *__restrict__ rc_next)
{
    if (MDS == MDS_LIMB22) {
        mds_limb22(s);
        if (rc_next) {
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = gl::add_wc(s[i], rc_next[i]);
        }
    } else {
        mds_half32(s, rc_next);
    }
}

// rc: 360 round constants (canonical); sp: tables of the optimised partial rounds (only read by MDS_SPARSE).
// State in: any u64 encodings; out: the first NCANON words canonical, the rest weakly reduced (a chained sponge
// permutation needs none canonical, a tree node only its 4 digest words).
template <int MDS, int NCANON = 12>
MI_HD void permute(u64 (&s)[12], const u64 *__restrict__ rc, const SparseTables *__restrict__ sp = nullptr)
{
    constexpr int FULL_MDS = (MDS == MDS_SPARSE) ? MDS_HALF32 : MDS;
    // round r: s <- M * S(s + rc_r).  The "+ rc_{r+1}" of the next round rides in the MDS of round r.
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_wc(s[i], rc[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        mds<FULL_MDS>(s, (MDS == MDS_SPARSE && r == 3) ? nullptr : rc + (r + 1) * 12);
    }
    if (MDS == MDS_SPARSE) {
        partial_rounds_sparse(s, *sp);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl::add_wc(s[i], rc[26 * 12 + i]);
    } else {
#pragma unroll 1
        for (int r = 4; r < 26; r++) {
            s[0] = sbox(s[0]);
            mds<FULL_MDS>(s, rc + (r + 1) * 12);
        }
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        mds<FULL_MDS>(s, r < 29 ? rc + (r + 1) * 12 : nullptr);
    }
#pragma unroll
    for (int i = 0; i < NCANON; i++) s[i] = gl::canon(s[i]);
}

} // namespace pos
