/*
 * gl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See gl_oracle.h.
 *
 * Plain C11 + optional OpenMP.  No code here is reachable from the product library; it is loaded
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#include "gl_oracle.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
#define P GLO_P

void glo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int glo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ base field */
uint64_t glo_canon(uint64_t a) { return a >= P ? a - P : a; }

uint64_t glo_add(uint64_t a, uint64_t b)
{
    a = glo_canon(a); b = glo_canon(b);
    uint64_t s = a + b;
    if (s < a || s >= P) s -= P; /* on wrap, s + 2^64 - P == s - P (mod 2^64) */
    return s;
}

uint64_t glo_sub(uint64_t a, uint64_t b)
{
    a = glo_canon(a); b = glo_canon(b);
    return a >= b ? a - b : a + (P - b);
}

static inline uint64_t reduce128(u128 x)
{
    /* 2^64 = 2^32 - 1, 2^96 = -1 (mod p):  x = lo + 2^64*(hl + 2^32*hh) = lo + hl*(2^32-1) - hh */
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFULL;      /* wrapped by 2^64 = p + (2^32-1) */
    uint64_t t1 = hl * 0xFFFFFFFFULL;
    uint64_t r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFULL;        /* carry out: 2^64 = 2^32-1 */
    return r >= P ? r - P : r;
}

uint64_t glo_mul(uint64_t a, uint64_t b) { return reduce128((u128)a * b); }

uint64_t glo_pow(uint64_t a, uint64_t e)
{
    uint64_t r = 1;
    a = glo_canon(a);
    while (e) {
        if (e & 1) r = glo_mul(r, a);
        a = glo_mul(a, a);
        e >>= 1;
    }
    return r;
}

uint64_t glo_inv(uint64_t a) { return glo_pow(a, P - 2); }

/* SURVEY App. B [PROBE]: w(n) = 7277203076849721926^(2^(32-n)); regenerates w(1..8) =
 * p-1, 2^48, 2^24, 4096, 64, 8, ...  and is confirmed for n in {3,4,9,12,16,20} by the golden
 * proofs' FRI relations (tests/test_oracle_golden.py). */
uint64_t glo_w(unsigned nbits)
{
    uint64_t r = 7277203076849721926ULL;
    for (unsigned i = nbits; i < 32; i++) r = glo_mul(r, r);
    return r;
}

uint64_t glo_shift(void) { return 49; }

/* ------------------------------------------------------------------ cubic extension
 * polinomial.hpp:195-205 (Karatsuba form); irreducible x^3 - x - 1. */
void glo3_mul(uint64_t out[3], const uint64_t a[3], const uint64_t b[3])
{
    uint64_t A = glo_mul(glo_add(a[0], a[1]), glo_add(b[0], b[1]));
    uint64_t B = glo_mul(glo_add(a[0], a[2]), glo_add(b[0], b[2]));
    uint64_t C = glo_mul(glo_add(a[1], a[2]), glo_add(b[1], b[2]));
    uint64_t D = glo_mul(a[0], b[0]);
    uint64_t E = glo_mul(a[1], b[1]);
    uint64_t F = glo_mul(a[2], b[2]);
    uint64_t G = glo_sub(D, E);
    uint64_t o0 = glo_sub(glo_add(C, G), F);
    uint64_t o1 = glo_sub(glo_sub(glo_sub(glo_add(A, C), E), E), D);
    uint64_t o2 = glo_sub(B, G);
    out[0] = o0; out[1] = o1; out[2] = o2;
}

void glo3_mul1(uint64_t out[3], const uint64_t a[3], uint64_t b)
{
    out[0] = glo_mul(a[0], b); out[1] = glo_mul(a[1], b); out[2] = glo_mul(a[2], b);
}
void glo3_add(uint64_t out[3], const uint64_t a[3], const uint64_t b[3])
{
    for (int i = 0; i < 3; i++) out[i] = glo_add(a[i], b[i]);
}
void glo3_sub(uint64_t out[3], const uint64_t a[3], const uint64_t b[3])
{
    for (int i = 0; i < 3; i++) out[i] = glo_sub(a[i], b[i]);
}

/* exact inverse in F_p^3 by Fermat over the norm: a^-1 = a^(p^2+p) / N(a); implemented with the
 * adjugate of the multiplication matrix (any exact inverse is bit-identical once canonical). */
void glo3_inv(uint64_t out[3], const uint64_t a[3])
{
    /* multiplication-by-a matrix columns: a*1, a*x, a*x^2 with x^3 = x + 1 */
    uint64_t one[3] = {1, 0, 0}, x1[3] = {0, 1, 0}, x2[3] = {0, 0, 1};
    uint64_t c0[3], c1[3], c2[3];
    glo3_mul(c0, a, one); glo3_mul(c1, a, x1); glo3_mul(c2, a, x2);
    /* M = [c0 c1 c2] (columns); solve M * y = e0 by Cramer */
    uint64_t m[3][3] = {{c0[0], c1[0], c2[0]}, {c0[1], c1[1], c2[1]}, {c0[2], c1[2], c2[2]}};
#define M2(a_, b_, c_, d_) glo_sub(glo_mul(a_, d_), glo_mul(b_, c_))
    uint64_t cof00 = M2(m[1][1], m[1][2], m[2][1], m[2][2]);
    uint64_t cof01 = M2(m[1][0], m[1][2], m[2][0], m[2][2]);
    uint64_t cof02 = M2(m[1][0], m[1][1], m[2][0], m[2][1]);
    uint64_t det = glo_add(glo_sub(glo_mul(m[0][0], cof00), glo_mul(m[0][1], cof01)), glo_mul(m[0][2], cof02));
    uint64_t di = glo_inv(det);
    /* y = first column of M^-1 = (cof00, -cof01, cof02) / det */
    out[0] = glo_mul(cof00, di);
    out[1] = glo_mul(glo_sub(0, cof01), di);
    out[2] = glo_mul(cof02, di);
#undef M2
}

/* ------------------------------------------------------------------ Poseidon
 * naive round form, exactly poseidon_g_executor.cpp:174-205: per round add RC, x^7 on all lanes in
 * rounds 0-3 and 26-29 else lane 0 only, then state = M*state with
 * M[i][j] = MCIRC[(j-i) mod 12] + (i==j ? MDIAG[i] : 0)  (poseidon_g_executor.hpp:40-50). */
static inline uint64_t pow7(uint64_t a)
{
    uint64_t a2 = glo_mul(a, a), a4 = glo_mul(a2, a2), a3 = glo_mul(a, a2);
    return glo_mul(a3, a4);
}

void glo_poseidon_perm(uint64_t st[12])
{
    for (int i = 0; i < 12; i++) st[i] = glo_canon(st[i]);
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) st[i] = glo_add(st[i], GLO_POS_RC[r * 12 + i]);
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; i++) st[i] = pow7(st[i]);
        } else {
            st[0] = pow7(st[0]);
        }
        uint64_t acc[12];
        for (int x = 0; x < 12; x++) {
            u128 s = 0; /* 12 * 2^64 * 49 < 2^74 */
            for (int y = 0; y < 12; y++) {
                uint64_t m = GLO_POS_MCIRC[(y - x + 12) % 12] + (x == y ? GLO_POS_MDIAG[x] : 0);
                s += (u128)st[y] * m;
            }
            acc[x] = reduce128(s);
        }
        memcpy(st, acc, sizeof acc);
    }
}

void glo_hash_full_result(uint64_t out[12], const uint64_t in[12])
{
    uint64_t st[12];
    memcpy(st, in, sizeof st);
    glo_poseidon_perm(st);
    memcpy(out, st, sizeof st);
}

void glo_hash(uint64_t out[4], const uint64_t in[12])
{
    uint64_t st[12];
    glo_hash_full_result(st, in);
    memcpy(out, st, 4 * sizeof(uint64_t));
}

/* size <= 4 -> copy + zero pad (no hash); else sponge, rate 8, capacity carried in
 * state[8..12) from the previous block's out[0..4). */
void glo_linear_hash(uint64_t out[4], const uint64_t *in, uint64_t size)
{
    if (size <= 4) {
        for (uint64_t i = 0; i < 4; i++) out[i] = i < size ? glo_canon(in[i]) : 0;
        return;
    }
    uint64_t st[12];
    uint64_t remaining = size;
    while (remaining) {
        if (remaining == size) memset(st + 8, 0, 4 * sizeof(uint64_t));
        else memcpy(st + 8, st, 4 * sizeof(uint64_t));
        uint64_t n = remaining < 8 ? remaining : 8;
        memset(st, 0, 8 * sizeof(uint64_t));
        memcpy(st, in + (size - remaining), n * sizeof(uint64_t));
        glo_poseidon_perm(st);
        remaining -= n;
    }
    memcpy(out, st, 4 * sizeof(uint64_t));
}

/* merkleTreeGL.cpp:37-44 -> PoseidonGoldilocks::merkletree_avx(nodes, source, width, height);
 * layout merkleTreeGL.hpp:61-68: leaves (nrows*4), then each upper level appended; root = last 4.
 * parent = hash(left || right || 0^4)[0..4). nrows must be a power of two (always is in starkpil). */
void glo_merkletree(uint64_t *nodes, const uint64_t *src, uint64_t ncols, uint64_t nrows)
{
    if (nrows == 0) return;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < nrows; i++) glo_linear_hash(nodes + i * 4, src + i * ncols, ncols);
    uint64_t pending = nrows, next_index = 0;
    while (pending > 1) {
        uint64_t next_n = pending / 2;
        uint64_t *lvl = nodes + next_index;
#pragma omp parallel for schedule(static) if (next_n > 256)
        for (uint64_t i = 0; i < next_n; i++) {
            uint64_t in[12];
            memcpy(in, lvl + i * 8, 8 * sizeof(uint64_t));
            memset(in + 8, 0, 4 * sizeof(uint64_t));
            glo_hash(lvl + (pending + i) * 4, in);
        }
        next_index += pending * 4;
        pending = next_n;
    }
}

void glo_merkle_group_proof(uint64_t *proof, const uint64_t *nodes, const uint64_t *src,
                            uint64_t height, uint64_t width, uint64_t idx)
{
    memcpy(proof, src + idx * width, width * sizeof(uint64_t));
    uint64_t *p = proof + width;
    uint64_t offset = 0, n = height;
    while (n > 1) {
        memcpy(p, nodes + offset + (idx ^ 1) * 4, 4 * sizeof(uint64_t));
        p += 4;
        offset += n * 4;
        n >>= 1;
        idx >>= 1;
    }
}

int glo_merkle_verify(const uint64_t root[4], const uint64_t *vals, uint64_t width,
                      const uint64_t *siblings, uint64_t nsib, uint64_t idx)
{
    uint64_t cur[4];
    glo_linear_hash(cur, vals, width);
    for (uint64_t l = 0; l < nsib; l++) {
        uint64_t in[12] = {0};
        const uint64_t *sib = siblings + l * 4;
        if (idx & 1) { memcpy(in, sib, 32); memcpy(in + 4, cur, 32); }
        else         { memcpy(in, cur, 32); memcpy(in + 4, sib, 32); }
        glo_hash(cur, in);
        idx >>= 1;
    }
    for (int i = 0; i < 4; i++)
        if (cur[i] != glo_canon(root[i])) return 0;
    return 1;
}

/* ------------------------------------------------------------------ NTT / LDE */
static unsigned ilog2(uint64_t n)
{
    unsigned b = 0;
    while ((1ULL << b) < n) b++;
    return b;
}

static uint64_t bitrev(uint64_t x, unsigned bits)
{
    uint64_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1ULL) << (bits - 1 - i);
    return r;
}

void glo_dft_naive(uint64_t *dst, const uint64_t *src, uint64_t n, int inverse)
{
    unsigned bits = ilog2(n);
    uint64_t w = glo_w(bits);
    if (inverse) w = glo_inv(w);
    uint64_t ninv = inverse ? glo_inv(n % P) : 1;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t wk = glo_pow(w, k), acc = 0, cur = 1;
        for (uint64_t i = 0; i < n; i++) {
            acc = glo_add(acc, glo_mul(src[i], cur));
            cur = glo_mul(cur, wk);
        }
        dst[k] = glo_mul(acc, ninv);
    }
}

/* Iterative radix-2 DIT on rows: dst[i] = src[bitrev(i)] (for the inverse the source index is
 * (n - bitrev(i)) % n, build_const_tree.cpp:115-124), then log2 n butterfly layers; inverse scales
 * by 1/n.  Natural order in and out; columns independent.  dst == src is allowed (starks.cpp:325). */
static void ntt_core(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t ncols, int inverse,
                     const uint64_t *row_scale /* optional per-row multiplier applied at the end */)
{
    unsigned bits = ilog2(n);
    uint64_t *tmp = NULL;
    const uint64_t *s = src;
    if (dst == src) {
        tmp = (uint64_t *)malloc(n * ncols * sizeof(uint64_t));
        memcpy(tmp, src, n * ncols * sizeof(uint64_t));
        s = tmp;
    }
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) {
        uint64_t ri = bitrev(i, bits);
        if (inverse) ri = (n - ri) % n;
        for (uint64_t c = 0; c < ncols; c++) dst[i * ncols + c] = glo_canon(s[ri * ncols + c]);
    }
    free(tmp);
    /* twiddle table w^j, j < n/2 */
    uint64_t half = n / 2 ? n / 2 : 1;
    uint64_t *tw = (uint64_t *)malloc(half * sizeof(uint64_t));
    uint64_t w = glo_w(bits);
    tw[0] = 1;
    for (uint64_t j = 1; j < half; j++) tw[j] = glo_mul(tw[j - 1], w);
    for (unsigned s_ = 1; s_ <= bits; s_++) {
        uint64_t m = 1ULL << s_, md2 = m >> 1, tstride = n / m;
#pragma omp parallel for schedule(static)
        for (uint64_t b = 0; b < n / 2; b++) {
            uint64_t grp = b / md2, j = b % md2;
            uint64_t i0 = grp * m + j, i1 = i0 + md2;
            uint64_t wj = tw[j * tstride];
            uint64_t *r0 = dst + i0 * ncols, *r1 = dst + i1 * ncols;
            for (uint64_t c = 0; c < ncols; c++) {
                uint64_t t = glo_mul(wj, r1[c]);
                uint64_t u = r0[c];
                r0[c] = glo_add(u, t);
                r1[c] = glo_sub(u, t);
            }
        }
    }
    free(tw);
    if (inverse || row_scale) {
        uint64_t ninv = inverse ? glo_inv(n % P) : 1;
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < n; i++) {
            uint64_t f = row_scale ? glo_mul(ninv, row_scale[i]) : ninv;
            for (uint64_t c = 0; c < ncols; c++) dst[i * ncols + c] = glo_mul(dst[i * ncols + c], f);
        }
    }
}

void glo_ntt(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t ncols, int inverse)
{
    if (n == 0 || ncols == 0) return;
    ntt_core(dst, src, n, ncols, inverse, NULL);
}

/* extendPol (starks.cpp:52,133,214): out[i] = P(shift * w_ext^i), natural order
 * = NTT_next( zero-pad( INTT_n(in)[k] * shift^k ) ), build_const_tree.cpp:160-196,198-331. */
void glo_extend_pol(uint64_t *out, const uint64_t *in, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    if (n == 0 || ncols == 0) return;
    uint64_t *sc = (uint64_t *)malloc(n * sizeof(uint64_t));
    sc[0] = 1;
    for (uint64_t i = 1; i < n; i++) sc[i] = glo_mul(sc[i - 1], glo_shift());
    uint64_t *coef = (uint64_t *)calloc(n_ext * ncols, sizeof(uint64_t));
    ntt_core(coef, in, n, ncols, 1, sc);
    free(sc);
    ntt_core(out, coef, n_ext, ncols, 0, NULL);
    free(coef);
}

/* ------------------------------------------------------------------ Transcript (transcript.cpp) */
void glo_transcript_init(glo_transcript *t) { memset(t, 0, sizeof *t); }

static void transcript_update(glo_transcript *t)
{
    uint64_t in[12];
    memcpy(in, t->pending, 8 * sizeof(uint64_t));
    memcpy(in + 8, t->state, 4 * sizeof(uint64_t));
    glo_hash_full_result(t->out, in);
    t->out_cursor = 12;
    memset(t->pending, 0, sizeof t->pending);
    t->pending_cursor = 0;
    memcpy(t->state, t->out, 4 * sizeof(uint64_t));
}

void glo_transcript_put(glo_transcript *t, const uint64_t *in, uint64_t size)
{
    for (uint64_t i = 0; i < size; i++) { /* transcript.cpp:12-29 _add1 */
        t->pending[t->pending_cursor++] = in[i];
        t->out_cursor = 0;
        if (t->pending_cursor == 8) transcript_update(t);
    }
}

uint64_t glo_transcript_get_fields1(glo_transcript *t)
{
    if (t->out_cursor == 0) transcript_update(t); /* transcript.cpp:41-57 */
    uint64_t r = t->out[(12 - t->out_cursor) % 12];
    t->out_cursor--;
    return r;
}

void glo_transcript_get_field(glo_transcript *t, uint64_t out[3])
{
    for (int i = 0; i < 3; i++) out[i] = glo_transcript_get_fields1(t);
}

void glo_transcript_get_permutations(glo_transcript *t, uint64_t *res, uint64_t n, uint64_t nbits)
{
    uint64_t total = n * nbits; /* transcript.cpp:59-87 */
    uint64_t nfields = (total - 1) / 63 + 1;
    uint64_t *fields = (uint64_t *)malloc(nfields * sizeof(uint64_t));
    for (uint64_t i = 0; i < nfields; i++) fields[i] = glo_transcript_get_fields1(t);
    uint64_t cur_field = 0, cur_bit = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t a = 0;
        for (uint64_t j = 0; j < nbits; j++) {
            uint64_t bit = (glo_canon(fields[cur_field]) >> cur_bit) & 1;
            if (bit) a += (1ULL << j);
            if (++cur_bit == 63) { cur_bit = 0; cur_field++; }
        }
        res[i] = a;
    }
    free(fields);
}

/* ------------------------------------------------------------------ FRI (friProve.cpp) */
static void intt_small3(uint64_t *c, const uint64_t *v, unsigned nx_bits)
{
    /* INTT over nX ext elements = 3 interleaved base columns (friProve.cpp:100-102) */
    uint64_t nx = 1ULL << nx_bits;
    if (nx == 1) { memcpy(c, v, 3 * sizeof(uint64_t)); return; }
    uint64_t winv = glo_inv(glo_w(nx_bits)), ninv = glo_inv(nx);
    for (uint64_t k = 0; k < nx; k++) {
        uint64_t wk = glo_pow(winv, k), cur = 1, acc[3] = {0, 0, 0};
        for (uint64_t i = 0; i < nx; i++) {
            for (int d = 0; d < 3; d++) acc[d] = glo_add(acc[d], glo_mul(v[i * 3 + d], cur));
            cur = glo_mul(cur, wk);
        }
        for (int d = 0; d < 3; d++) c[k * 3 + d] = glo_mul(acc[d], ninv);
    }
}

static void fold_coeffs(uint64_t out[3], uint64_t *c, uint64_t nx, uint64_t sinv, const uint64_t x[3])
{
    uint64_t r = 1; /* polMulAxi friProve.cpp:192-200 */
    for (uint64_t k = 0; k < nx; k++) {
        glo3_mul1(c + 3 * k, c + 3 * k, r);
        r = glo_mul(r, sinv);
    }
    uint64_t acc[3]; /* evalPol Horner friProve.cpp:201-217 */
    memcpy(acc, c + 3 * (nx - 1), sizeof acc);
    for (int64_t k = (int64_t)nx - 2; k >= 0; k--) {
        uint64_t t[3];
        glo3_mul(t, acc, x);
        glo3_add(acc, t, c + 3 * k);
    }
    memcpy(out, acc, sizeof acc);
}

void glo_fri_fold_group(uint64_t out[3], const uint64_t *vals, unsigned nx_bits, unsigned prev_bits,
                        unsigned nbits_ext, uint64_t g, const uint64_t special_x[3])
{
    uint64_t nx = 1ULL << nx_bits;
    uint64_t *c = (uint64_t *)malloc(nx * 3 * sizeof(uint64_t));
    intt_small3(c, vals, nx_bits);
    /* polShiftInv after the previous steps = shift^-(2^(nbits_ext - prev_bits)) (friProve.cpp:143-147) */
    uint64_t sinv = glo_inv(glo_shift());
    for (unsigned j = 0; j < nbits_ext - prev_bits; j++) sinv = glo_mul(sinv, sinv);
    uint64_t wi = glo_inv(glo_w(prev_bits));
    sinv = glo_mul(sinv, glo_pow(wi, g));
    fold_coeffs(out, c, nx, sinv, special_x);
    free(c);
}

void glo_fri_fold(uint64_t *out, const uint64_t *pol, unsigned prev_bits, unsigned cur_bits,
                  unsigned nbits_ext, const uint64_t special_x[3])
{
    uint64_t pol2n = 1ULL << cur_bits;
    unsigned nx_bits = prev_bits - cur_bits;
    uint64_t nx = 1ULL << nx_bits;
    if (nx == 1) { memcpy(out, pol, pol2n * 3 * sizeof(uint64_t)); return; }
#pragma omp parallel for schedule(static)
    for (uint64_t g = 0; g < pol2n; g++) {
        uint64_t ppar[64 * 3];
        uint64_t *pp = nx <= 64 ? ppar : (uint64_t *)malloc(nx * 3 * sizeof(uint64_t));
        for (uint64_t i = 0; i < nx; i++) memcpy(pp + 3 * i, pol + 3 * (i * pol2n + g), 3 * sizeof(uint64_t));
        glo_fri_fold_group(out + 3 * g, pp, nx_bits, prev_bits, nbits_ext, g, special_x);
        if (pp != ppar) free(pp);
    }
}

void glo_fri_transpose(uint64_t *aux, const uint64_t *pol, uint64_t degree, unsigned transpose_bits)
{
    uint64_t w = 1ULL << transpose_bits, h = degree / w; /* friProve.cpp:252-271 */
    for (uint64_t i = 0; i < w; i++)
        for (uint64_t j = 0; j < h; j++)
            memcpy(aux + 3 * (i * h + j), pol + 3 * (j * w + i), 3 * sizeof(uint64_t));
}

/* ------------------------------------------------------------------ step-4 split (starks.cpp:265-280) */
void glo_q_split(uint64_t *qq2, const uint64_t *qq1, uint64_t n, unsigned qdeg)
{
    uint64_t shift_in = glo_pow(glo_inv(glo_shift()), n);
    uint64_t cur = 1;
    for (unsigned p = 0; p < qdeg; p++) {
        for (uint64_t k = 0; k < n; k++)
            glo3_mul1(qq2 + 3 * (k * qdeg + p), qq1 + 3 * (p * n + k), cur);
        cur = glo_mul(cur, shift_in);
    }
}

/* ------------------------------------------------------------------ evmap (starks.cpp:555-668) */
void glo_evmap(uint64_t *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits,
               const uint64_t *const *pol_ptr, const uint32_t *pol_dim, const uint64_t *pol_stride,
               const uint8_t *prime, const uint64_t *lev, const uint64_t *lpev)
{
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n_evals; i++) {
        uint64_t acc[3] = {0, 0, 0};
        const uint64_t *L = prime[i] ? lpev : lev;
        for (uint64_t k = 0; k < n; k++) {
            const uint64_t *b = pol_ptr[i] + (k << ext_bits) * pol_stride[i];
            uint64_t t[3];
            if (pol_dim[i] == 1) glo3_mul1(t, L + 3 * k, b[0]); /* polinomial.hpp:722-743 */
            else glo3_mul(t, L + 3 * k, b);
            glo3_add(acc, acc, t);
        }
        memcpy(evals + 3 * i, acc, sizeof acc);
    }
}

void glo_batch_inverse3(uint64_t *res, const uint64_t *src, uint64_t n)
{
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; i++) {
        uint64_t t[3];
        glo3_inv(t, src + 3 * i);
        memcpy(res + 3 * i, t, sizeof t);
    }
}

/* ---- plookup h1 / h2 (polinomial.hpp:303-347): the reference's std::map from the value of a row of t to its (last) index is a
 * sorted array here */
typedef struct { uint64_t k[3]; uint64_t idx; } h1h2_key;
static int h1h2_cmp(const void *a, const void *b)
{
    const h1h2_key *x = a, *y = b;
    for (int j = 0; j < 3; j++)
        if (x->k[j] != y->k[j]) return x->k[j] < y->k[j] ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
int64_t glo_calculate_h1h2(uint64_t *h1, uint64_t h1_stride, uint64_t *h2, uint64_t h2_stride, const uint64_t *f, uint64_t f_stride,
                           const uint64_t *t, uint64_t t_stride, unsigned dim, uint64_t n)
{
    if (!n) return 0;
    h1h2_key *keys = malloc(n * sizeof *keys);
    uint64_t *counter = malloc(n * sizeof *counter);
    for (uint64_t i = 0; i < n; i++) {
        for (unsigned j = 0; j < 3; j++) keys[i].k[j] = j < dim ? t[i * t_stride + j] : 0;
        keys[i].idx = i;
        counter[i] = 1; /* :309 vector<int> counter(tPol.degree(), 1) */
    }
    qsort(keys, n, sizeof *keys, h1h2_cmp);
    int64_t bad = 0;
    for (uint64_t i = 0; i < n && !bad; i++) {
        h1h2_key q = {{0, 0, 0}, UINT64_MAX};
        for (unsigned j = 0; j < dim; j++) q.k[j] = f[i * f_stride + j];
        uint64_t lo = 0, hi = n; /* the last entry that sorts below (value, infinity): idx_t[key] = i + 1 keeps the last row (:314) */
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2;
            if (h1h2_cmp(&keys[mid], &q) < 0) lo = mid + 1; else hi = mid;
        }
        if (lo == 0 || memcmp(keys[lo - 1].k, q.k, sizeof q.k)) { bad = (int64_t)i + 1; break; } /* :321-325 */
        counter[keys[lo - 1].idx]++;
    }
    if (!bad) {
        uint64_t id = 0; /* :330-346 */
        for (uint64_t i = 0; i < n; i++) {
            if (counter[id] == 0) ++id;
            counter[id] -= 1;
            memcpy(h1 + i * h1_stride, t + id * t_stride, dim * 8);
            if (counter[id] == 0) ++id;
            counter[id] -= 1;
            memcpy(h2 + i * h2_stride, t + id * t_stride, dim * 8);
        }
    }
    free(keys); free(counter);
    return bad;
}

int glo_calculate_z(uint64_t *z, uint64_t z_stride, const uint64_t *num, uint64_t num_stride, const uint64_t *den, uint64_t den_stride,
                    uint64_t n)
{
    if (!n) return 1;
    uint64_t cur[3] = {1, 0, 0}, di[3], tmp[3]; /* polinomial.hpp:592-593 */
    for (uint64_t i = 0; i < n; i++) {
        memcpy(z + i * z_stride, cur, sizeof cur);
        glo3_inv(di, den + i * den_stride);          /* :595 batchInverse(denI, den): the same values one at a time */
        glo3_mul(tmp, num + i * num_stride, di);     /* :599 */
        glo3_mul(cur, cur, tmp);                     /* :600 */
    }
    return cur[0] == 1 && cur[1] == 0 && cur[2] == 0; /* :603-606 */
}

void glo_geom_seq(uint64_t *out, uint64_t n, uint64_t start, uint64_t ratio)
{
    uint64_t x = glo_canon(start); /* starks.hpp:149-160 */
    for (uint64_t i = 0; i < n; i++) { out[i] = x; x = glo_mul(x, ratio); }
}

void glo_geom_seq3(uint64_t *out, uint64_t n, const uint64_t ratio[3])
{
    uint64_t cur[3] = {1, 0, 0}; /* starks.cpp:311-323 */
    for (uint64_t k = 0; k < n; k++) {
        memcpy(out + 3 * k, cur, sizeof cur);
        glo3_mul(cur, cur, ratio);
    }
}

void glo_zhinv(uint64_t *out, unsigned nbits, unsigned nbits_ext)
{
    uint64_t w = 1, sn = glo_shift(); /* zhInv.cpp:7-31 */
    unsigned ext = nbits_ext - nbits;
    for (unsigned i = 0; i < nbits; i++) sn = glo_mul(sn, sn);
    for (uint64_t i = 0; i < (1ULL << ext); i++) {
        out[i] = glo_inv(glo_sub(glo_mul(sn, w), 1));
        w = glo_mul(w, glo_w(ext));
    }
}
