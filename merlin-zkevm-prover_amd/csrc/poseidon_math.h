// poseidon_math.h -- Poseidon-over-Goldilocks permutation (t=12, R_F=8, R_P=22, x^7), one state
// per lane, all 12 words in VGPRs.
//
// Spec followed: src/sm/poseidon_g/poseidon_g_executor.cpp:174-205 (naive round form: add RC, S-box on
// all lanes in rounds 0-3/26-29 else lane 0, state = M*state) with M = circ(MCIRC) + diag(MDIAG)
// (poseidon_g_executor.hpp:37-50).  Full rounds are computed in that form: the MDS constants are < 2^6, so the
// mat-vec needs no 64x64 multiplies (mds_half32).  The 22 partial rounds are computed in an algebraically equal
// "grouped" form derived from the textbook optimised partial rounds (partial_rounds_sparse, tables generated and
// verified against the naive form by tools/gen_poseidon_sparse.py); the naive partial rounds stay available as
// variants 0 / 1 and all variants are bit-identical.
//
// Why one state per lane and not a wave-cooperative round with the state spread over 12 lanes: 22 of
// the 30 rounds apply the S-box to lane 0 only, which would idle 11 of 12 cooperating lanes for ~2/3 of
// the multiplies; with a whole state per lane every VALU slot does useful work and no LDS / cross-lane
// traffic is needed.  (MFMA: not applicable -- exact 64-bit modular accumulation, 12x12 only.)
#pragma once
#include <type_traits>
#include "gl_math.h"
#include "poseidon_constants.h"
#include "poseidon_sparse_constants.h"

namespace pos {

static constexpr int MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20}; // == MI_POS_MCIRC
static constexpr int MD0 = 8;                                                   // == MI_POS_MDIAG[0]

enum { MDS_HALF32 = 0, MDS_LIMB22 = 1, MDS_SPARSE = 2 };

template <int I, int N, typename F>
MI_HD void nttm_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        nttm_static_for<I + 1, N>(f);
    }
}

MI_HD u64 sbox(u64 x)
{
    // squarings go through the general multiply: the dedicated 3-product square needs a doubling and two more
    // zero-extended adds, which on this ISA cost more than the fourth v_mad_u64_u32
    u64 x2 = gl::mul_w(x, x);
    u64 x4 = gl::mul_w(x2, x2);
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
    bool ovf[12];
    // the entries 2 and 16 would otherwise become "zero-extend to a register pair (2 moves), shift-add": keep them
    // multiply-adds like the other ten
    const u32 m2 = gl::opaque_u32(2), m16 = gl::opaque_u32(16);
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); rcn[i] = rc_next ? rc_next[i] : 0; }
#pragma unroll
    for (int x = 0; x < 12; x++) {
        u64 al = (u64)(u32)rcn[x], ah = rcn[x] >> 32;
#pragma unroll
        for (int y = 0; y < 12; y++) {
            const u32 mc = (u32)(MC[(y - x + 12) % 12] + ((x == 0 && y == 0) ? MD0 : 0));
            const u32 m = mc == 2 ? m2 : mc == 16 ? m16 : mc;
            al += (u64)lo[y] * m;
            ah += (u64)hi[y] * m;
        }
        // value = al + ah*2^32 = al + (ah_lo << 32) + ah_hi * 2^64 ;  2^64 = 2^32 - 1
        const u32 ahl = (u32)ah, ahh = (u32)(ah >> 32);   // ahh < 2^10
        const u64 r = (u64)ahh * 0xFFFFFFFFu + al;        // < 2^43: one multiply-add, no wrap
        const u32 thi = (u32)(r >> 32) + ahl;             // adding ah_lo * 2^32 touches the high word only
        ovf[x] = thi < ahl;                               // wraps 2^64 with probability ~2^-21 per value
        s[x] = ((u64)thi << 32) | (u32)r;
    }
    // one wave-uniform check for all twelve rows, after every multiply-add has been issued (a branch per row splits
    // the block and un-hoists the constant loads)
    bool any = false;
#pragma unroll
    for (int x = 0; x < 12; x++) any |= ovf[x];
    if (gl::rare(any)) {
        MI_KEEP_BRANCH();
#pragma unroll
        for (int x = 0; x < 12; x++) s[x] = ovf[x] ? s[x] + GL_EPS : s[x]; // wrapped value < 2^43: no second wrap
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
// The textbook optimised form makes a partial round "S-box on s0, one 12-term dot product with 64-bit constants for
// the new s0, eleven s_i += w_i * s0".  On CDNA4 the eleven multiply-adds are the expensive part (a full 128-bit
// reduction each, ~25 instructions), while a term of a dot product costs 6 v_mad_u64_u32 and nothing else: each
// constant is pre-split into 22/22/20-bit limbs, so the six 64-bit accumulators (2 halves of the variable x 3 limbs)
// stay below 2^59 over 23 terms and a single reduction (dot_close) closes the row.  The grouped form below therefore
// rewrites the s_i updates of 11 consecutive rounds into dot products as well.
// Tables of the grouped optimised partial rounds (tools/gen_poseidon_sparse.py, section 3), every 64-bit constant
// pre-split into 22/22/20-bit limbs.  Two groups of 11 rounds share one code path.
struct GroupTables {
    u64 k[12];             // scalar added to element 0 after the S-box of round r (k of round 21 is 0; [11] pads)
    u32 d[11][11][3];      // D[r][j]: coefficient of z_j (the group's starting s[1+j]) in round r's new element 0
    u32 c[55][3];          // C[r (r-1)/2 + t]: coefficient of y_t (post-S-box element 0 of round t < r)
    u32 w[11][11][3];      // W[i][t]: coefficient of y_t in the group's closing z_i
    u32 pre[11][11][3];    // PRE[i][j]: closing z_i = PRE[i] . z (group 0) -- group 1 closes with z_i itself
};
struct SparseTables {
    u64 first_rc[12];
    GroupTables g[2];
};

inline void split22(u32 (&l)[3], u64 v)
{
    l[0] = (u32)(v & 0x3FFFFF); l[1] = (u32)((v >> 22) & 0x3FFFFF); l[2] = (u32)(v >> 44);
}

inline void fill_sparse_tables(SparseTables &t)
{
    t = SparseTables{};
    for (int i = 0; i < 12; i++) t.first_rc[i] = MI_POS_FIRST_RC[i];
    for (int g = 0; g < 2; g++) {
        GroupTables &T = t.g[g];
        for (int r = 0; r < 11; r++) {
            T.k[r] = (11 * g + r < 21) ? MI_POS_K[11 * g + r] : 0;
            for (int j = 0; j < 11; j++) {
                split22(T.d[r][j], MI_POS_GD[(g * 11 + r) * 11 + j]);
                split22(T.w[j][r], MI_POS_W[(11 * g + r) * 11 + j]); // W[i = j][t = r] = w[11 g + t][i]
                split22(T.pre[r][j], MI_POS_PRE[r * 11 + j]);
            }
        }
        for (int e = 0; e < 55; e++) split22(T.c[e], MI_POS_GC[g * 55 + e]);
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

// value = sum_{h,l} a[h][l] * 2^(32h + 22l), every a < 2^58  ->  weakly reduced.
// The twelve 32-bit words of the six accumulators are gathered by multiply-adds into the coefficients of
// 2^0, 2^32, 2^64, 2^96 (U0..U3, no overflow: each stays below 2^60) and folded with 2^64 = 2^32 - 1, 2^96 = -1.
MI_HD u64 dot_close(const DotAcc &d)
{
    const u32 k22 = gl::opaque_u32(1u << 22), k12 = gl::opaque_u32(1u << 12);
    const u64 A00 = d.a[0][0], A01 = d.a[0][1], A02 = d.a[0][2], A10 = d.a[1][0], A11 = d.a[1][1], A12 = d.a[1][2];
    // bit positions: A00 0, A01 22, A02 44, A10 32, A11 54, A12 76; a word w of A at position q sits at q (+32 for the high word)
    const u64 U0 = (u64)(u32)A01 * k22 + A00;                       // 2^0 : A00 (whole) + lo(A01) 2^22
    u64 U1 = (u64)(u32)(A01 >> 32) * k22 + A10;                     // 2^32: A10 (whole) + hi(A01) 2^22 + lo(A02) 2^12 + lo(A11) 2^22
    U1 = (u64)(u32)A02 * k12 + U1;
    U1 = (u64)(u32)A11 * k22 + U1;
    u64 U2 = (u64)(u32)(A02 >> 32) * k12;                           // 2^64: hi(A02) 2^12 + hi(A11) 2^22 + lo(A12) 2^12
    U2 = (u64)(u32)(A11 >> 32) * k22 + U2;
    U2 = (u64)(u32)A12 * k12 + U2;
    const u64 U3 = (u64)(u32)(A12 >> 32) * k12;                     // 2^96: hi(A12) 2^12   (< 2^38)
    const u64 lo = U0 + (U1 << 32);
    const u64 W2 = U2 + (U1 >> 32) + (lo < U0 ? 1 : 0);             // coefficient of 2^64, < 2^50
    const u32 hl = (u32)W2;
    const u64 HH = (W2 >> 32) + U3;                                 // coefficient of 2^96, < 2^39
    u64 t0 = lo - HH;
    if (gl::rare(lo < HH)) { // needs lo < 2^39
        MI_KEEP_BRANCH();
        t0 = lo < HH ? t0 - GL_EPS : t0;
    }
    const u64 t1 = (u64)hl * 0xFFFFFFFFu;
    const u64 r = t0 + t1;
    return r + (r < t1 ? GL_EPS : 0);
}

// The 22 partial rounds, grouped (see the generator): per round one S-box and ONE dot product (11 terms over the
// group's starting z = s[1:], r terms over the earlier y_t); per group eleven closing dot products.  No
// "s_i += w * s0" with its own 128-bit reduction is ever materialised.
MI_HD void partial_rounds_sparse(u64 (&s)[12], const SparseTables &t)
{
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_wc(s[i], t.first_rc[i]);
#pragma unroll 1
    for (int g = 0; g < 2; g++) {
        const GroupTables &T = t.g[g];
        u64 y[11];
        u64 s0 = s[0];
        nttm_static_for<0, 11>([&](auto R) {
            constexpr int r = decltype(R)::value;
            y[r] = gl::add_wc(sbox(s0), T.k[r]);
            DotAcc d = {};
            d.a[0][0] = (u64)(u32)y[r] * MI_POS_M00;
            d.a[1][0] = (u64)(u32)(y[r] >> 32) * MI_POS_M00;
#pragma unroll
            for (int j = 0; j < 11; j++) dot_acc(d, s[1 + j], T.d[r][j]);
#pragma unroll
            for (int tt = 0; tt < r; tt++) dot_acc(d, y[tt], T.c[r * (r - 1) / 2 + tt]);
            s0 = dot_close(d);
        });
        u64 z[11];
#pragma unroll
        for (int i = 0; i < 11; i++) {
            DotAcc d = {};
            if (g == 0) {
#pragma unroll
                for (int j = 0; j < 11; j++) dot_acc(d, s[1 + j], T.pre[i][j]);
            } else {
                d.a[0][0] = (u32)s[1 + i];
                d.a[1][0] = s[1 + i] >> 32;
            }
#pragma unroll
            for (int tt = 0; tt < 11; tt++) dot_acc(d, y[tt], T.w[i][tt]);
            z[i] = dot_close(d);
        }
#pragma unroll
        for (int i = 0; i < 11; i++) s[1 + i] = z[i];
        s[0] = s0;
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
// before_last_rounds: called once, after the partial rounds and before the closing four full rounds -- the point of
// lowest register pressure with still ~a quarter of the permutation to run.  The leaf sponge issues its next line
// fetch there, so the registers the data lands in are not live across the (register-hungry) partial rounds.
struct NoHook { MI_HD void operator()() const {} };

template <int MDS, int NCANON = 12, typename Hook = NoHook>
MI_HD void permute(u64 (&s)[12], const u64 *__restrict__ rc, const SparseTables *__restrict__ sp = nullptr,
                   Hook before_last_rounds = Hook())
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
    before_last_rounds();
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
