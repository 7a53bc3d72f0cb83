/* cpu_baseline_avx2.c -- the CPU BASELINE leg of bench.py (TEST INFRASTRUCTURE, never linked or called by the product, and not the
 * checker either: tests/ compare the GPU with the plain restatement in gl_oracle.c; this file is itself checked against that
 * restatement in tests/test_cpu_baseline.py).
 *
 * Why it exists: SURVEY 8(d) asks for the reference's CPU path (src/goldilocks: AVX2 + OpenMP) timed beside the GPU on the node's
 * own cores.  That library is an absent git submodule, and the checker's naive scalar code (32 us per permutation, a layer-by-layer
 * NTT over the whole matrix) is a straw man next to it.  This is a hand-vectorised restatement of the same algorithms in the style of
 * the upstream library -- labelled "restatement, not upstream" wherever its number is printed:
 *   Poseidon  4 sponges per __m256i (one row of the trace per 64-bit lane), state in 12 vectors; 64 x 64 -> 128 products from four
 *             _mm256_mul_epu32, the Goldilocks reduction in adds / shifts (2^64 = 2^32 - 1, 2^96 = -1); the MDS with its constants
 *             (< 2^6) as 32 x 32 multiply-adds on the two halves of every word, no reduction until a row closes; round structure of
 *             poseidon_g_executor.cpp:174-205 (the 22 partial rounds apply the S-box to element 0 only); OpenMP over groups of rows.
 *   LDE       per block of 4 adjacent columns: gathered into a private [rows][4] buffer (one vector per row), decimation-in-frequency
 *             INTT_N (natural in, bit-reversed out), scale by shift^k / N, zero-padded decimation-in-time NTT_2N (bit-reversed in,
 *             natural out) -- no bit-reversal pass -- butterflies on vectors, scattered back; OpenMP over column blocks.
 */
#include <immintrin.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "gl_oracle.h"
#include "poseidon_constants.h"

typedef __m256i V;
#define P_ 0xFFFFFFFF00000001ULL
#define EPS_ 0xFFFFFFFFULL

static inline V vset(uint64_t x) { return _mm256_set1_epi64x((long long)x); }
static inline V lt_u(V a, V b) /* a < b, unsigned, as a lane mask */
{
    const V s = vset(0x8000000000000000ULL);
    return _mm256_cmpgt_epi64(_mm256_xor_si256(b, s), _mm256_xor_si256(a, s));
}
/* any u64 + any u64 whose true sum is < 2^64 + p  ->  weakly reduced */
static inline V add_w(V a, V b)
{
    V s = _mm256_add_epi64(a, b);
    return _mm256_add_epi64(s, _mm256_and_si256(lt_u(s, a), vset(EPS_)));
}
/* a any u64, b canonical -> weakly reduced a - b */
static inline V sub_w(V a, V b)
{
    V d = _mm256_sub_epi64(a, b);
    return _mm256_sub_epi64(d, _mm256_and_si256(lt_u(a, b), vset(EPS_)));
}
static inline V canon(V a)
{
    V ge = _mm256_xor_si256(lt_u(a, vset(P_)), _mm256_set1_epi64x(-1));
    return _mm256_sub_epi64(a, _mm256_and_si256(ge, vset(P_)));
}
/* (hi:lo) -> weakly reduced: lo - hh + hl * (2^32 - 1) */
static inline V reduce128(V lo, V hi)
{
    const V eps = vset(EPS_);
    V hh = _mm256_srli_epi64(hi, 32), hl = _mm256_and_si256(hi, eps);
    V t0 = _mm256_sub_epi64(lo, hh);
    t0 = _mm256_sub_epi64(t0, _mm256_and_si256(lt_u(lo, hh), eps));
    V t1 = _mm256_sub_epi64(_mm256_slli_epi64(hl, 32), hl);
    V r = _mm256_add_epi64(t0, t1);
    return _mm256_add_epi64(r, _mm256_and_si256(lt_u(r, t1), eps));
}
static inline V mul_w(V a, V b)
{
    const V m32 = vset(EPS_);
    V ah = _mm256_srli_epi64(a, 32), bh = _mm256_srli_epi64(b, 32);
    V ll = _mm256_mul_epu32(a, b), lh = _mm256_mul_epu32(a, bh), hl = _mm256_mul_epu32(ah, b), hh = _mm256_mul_epu32(ah, bh);
    V mid = _mm256_add_epi64(lh, _mm256_srli_epi64(ll, 32));           /* < 2^64 */
    V mid2 = _mm256_add_epi64(hl, _mm256_and_si256(mid, m32));         /* < 2^64 */
    V hi = _mm256_add_epi64(_mm256_add_epi64(hh, _mm256_srli_epi64(mid, 32)), _mm256_srli_epi64(mid2, 32));
    V lo = _mm256_or_si256(_mm256_and_si256(ll, m32), _mm256_slli_epi64(mid2, 32));
    return reduce128(lo, hi);
}
static inline V sbox(V x)
{
    V x2 = mul_w(x, x), x4 = mul_w(x2, x2), x3 = mul_w(x, x2);
    return mul_w(x3, x4);
}

static const int MC_[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

/* s <- M s, M = circ(MC_) + diag(8, 0, ...), on any u64 encodings -> weakly reduced */
static inline void mds(V s[12])
{
    const V m32 = vset(EPS_);
    V hi[12], out[12];
    for (int j = 0; j < 12; j++) hi[j] = _mm256_srli_epi64(s[j], 32);
    for (int i = 0; i < 12; i++) {
        V al = _mm256_setzero_si256(), ah = _mm256_setzero_si256();
        for (int j = 0; j < 12; j++) {
            const V c = vset((uint64_t)(MC_[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8 : 0)));
            al = _mm256_add_epi64(al, _mm256_mul_epu32(s[j], c)); /* low halves: mul_epu32 reads bits 0..31 of each lane */
            ah = _mm256_add_epi64(ah, _mm256_mul_epu32(hi[j], c));
        }
        /* value = al + ah_lo 2^32 + ah_hi 2^64, al < 2^41, ah_hi < 2^9 */
        V ahh = _mm256_srli_epi64(ah, 32), ahl = _mm256_slli_epi64(_mm256_and_si256(ah, m32), 32);
        V r = _mm256_add_epi64(al, _mm256_sub_epi64(_mm256_slli_epi64(ahh, 32), ahh)); /* + ah_hi (2^32 - 1): < 2^42 */
        out[i] = add_w(ahl, r);
    }
    for (int i = 0; i < 12; i++) s[i] = out[i];
}

/* four permutations at once: s[i] lane k = element i of state k; any encodings in, canonical out */
static inline void perm4(V s[12])
{
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = add_w(s[i], vset(GLO_POS_RC[r * 12 + i]));
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        } else {
            s[0] = sbox(s[0]);
        }
        mds(s);
    }
    for (int i = 0; i < 12; i++) s[i] = canon(s[i]);
}

/* test hook: count states of 12 words, count a multiple of 4 */
void glb_poseidon_perm_batch(uint64_t *states, uint64_t count)
{
#pragma omp parallel for schedule(static)
    for (uint64_t g = 0; g < count / 4; g++) {
        uint64_t *p = states + g * 48;
        V s[12];
        for (int i = 0; i < 12; i++) s[i] = _mm256_set_epi64x((long long)p[36 + i], (long long)p[24 + i], (long long)p[12 + i], (long long)p[i]);
        perm4(s);
        for (int i = 0; i < 12; i++) {
            uint64_t t[4];
            _mm256_storeu_si256((V *)t, s[i]);
            for (int k = 0; k < 4; k++) p[k * 12 + i] = t[k];
        }
    }
}

/* linear_hash of four rows at once: widths <= 4 are copied, else the sponge over blocks of 8 */
static void linear_hash4(uint64_t *out /* 4 digests */, const uint64_t *r0, const uint64_t *r1, const uint64_t *r2, const uint64_t *r3, uint64_t ncols)
{
    if (ncols <= 4) {
        const uint64_t *rows[4] = {r0, r1, r2, r3};
        for (int k = 0; k < 4; k++)
            for (uint64_t c = 0; c < 4; c++) out[k * 4 + c] = c < ncols ? glo_canon(rows[k][c]) : 0;
        return;
    }
    V s[12];
    V cap[4] = {_mm256_setzero_si256(), _mm256_setzero_si256(), _mm256_setzero_si256(), _mm256_setzero_si256()};
    for (uint64_t c0 = 0; c0 < ncols; c0 += 8) {
        for (uint64_t j = 0; j < 8; j++) {
            const uint64_t c = c0 + j;
            s[j] = c < ncols ? _mm256_set_epi64x((long long)r3[c], (long long)r2[c], (long long)r1[c], (long long)r0[c]) : _mm256_setzero_si256();
        }
        for (int j = 0; j < 4; j++) s[8 + j] = cap[j];
        perm4(s);
        for (int j = 0; j < 4; j++) cap[j] = s[j];
    }
    for (int j = 0; j < 4; j++) {
        uint64_t t[4];
        _mm256_storeu_si256((V *)t, cap[j]);
        for (int k = 0; k < 4; k++) out[k * 4 + j] = t[k];
    }
}

/* merkletree (merkleTreeGL.cpp:37-44 -> PoseidonGoldilocks::merkletree_avx): nodes = leaf digests, then every level */
void glb_merkletree(uint64_t *nodes, const uint64_t *src, uint64_t ncols, uint64_t nrows)
{
    if (nrows == 0) return;
    if (nrows < 4) { glo_merkletree(nodes, src, ncols, nrows); return; }
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < nrows; r += 4)
        linear_hash4(nodes + r * 4, src + r * ncols, src + (r + 1) * ncols, src + (r + 2) * ncols, src + (r + 3) * ncols, ncols);
    uint64_t off = 0;
    for (uint64_t n = nrows; n > 1; n >>= 1) {
        const uint64_t *in = nodes + off;
        uint64_t *outp = nodes + off + n * 4;
        const uint64_t pairs = n / 2;
        if (pairs >= 4) {
#pragma omp parallel for schedule(static) if (pairs > 1024)
            for (uint64_t i = 0; i < pairs; i += 4) {
                V s[12];
                for (int j = 0; j < 8; j++) s[j] = _mm256_set_epi64x((long long)in[(i + 3) * 8 + j], (long long)in[(i + 2) * 8 + j], (long long)in[(i + 1) * 8 + j], (long long)in[i * 8 + j]);
                for (int j = 8; j < 12; j++) s[j] = _mm256_setzero_si256();
                perm4(s);
                for (int j = 0; j < 4; j++) {
                    uint64_t t[4];
                    _mm256_storeu_si256((V *)t, s[j]);
                    for (int k = 0; k < 4; k++) outp[(i + k) * 4 + j] = t[k];
                }
            }
        } else {
            for (uint64_t i = 0; i < pairs; i++) {
                uint64_t st[12] = {0}, o[12];
                memcpy(st, in + i * 8, 64);
                glo_hash_full_result(o, st);
                memcpy(outp + i * 4, o, 32);
            }
        }
        off += n * 4;
    }
}

/* ------------------------------------------------------------------ LDE */
static unsigned ilog2_(uint64_t n) { unsigned b = 0; while ((1ULL << b) < n) b++; return b; }
static uint64_t bitrev_(uint64_t x, unsigned bits) { uint64_t r = 0; for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

/* out[i] = P(shift w_ext^i) per column: build_const_tree.cpp:160-196,198-331 semantics (= glo_extend_pol) */
void glb_extend_pol(uint64_t *out, const uint64_t *in, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    if (n == 0 || ncols == 0) return;
    const unsigned lb = ilog2_(n), le = ilog2_(n_ext);
    /* twiddles: winv[j] = w_n^-j (j < n/2), wext[j] = w_ext^j (j < n_ext/2); scale in bit-reversed order: sc[i] = shift^br(i) / n */
    uint64_t *winv = (uint64_t *)malloc(((n / 2) ? n / 2 : 1) * 8), *wext = (uint64_t *)malloc(((n_ext / 2) ? n_ext / 2 : 1) * 8), *sc = (uint64_t *)malloc(n * 8);
    const uint64_t wi = glo_inv(glo_w(lb)), we = glo_w(le), ninv = glo_inv(n % P_);
    winv[0] = 1;
    for (uint64_t j = 1; j < n / 2; j++) winv[j] = glo_mul(winv[j - 1], wi);
    wext[0] = 1;
    for (uint64_t j = 1; j < n_ext / 2; j++) wext[j] = glo_mul(wext[j - 1], we);
    {
        uint64_t cur = ninv;
        for (uint64_t k = 0; k < n; k++) { sc[bitrev_(k, lb)] = cur; cur = glo_mul(cur, glo_shift()); }
    }
    const uint64_t step = n_ext / n; /* coefficient k sits at position br_n(k) after the DIF; br_ext(k) = br_n(k) * step for k < n */
#pragma omp parallel
    {
        V *a = (V *)aligned_alloc(64, n_ext * sizeof(V));
#pragma omp for schedule(dynamic, 1)
        for (uint64_t c0 = 0; c0 < ncols; c0 += 4) {
            const uint64_t cw = ncols - c0 < 4 ? ncols - c0 : 4;
            for (uint64_t r = 0; r < n; r++) {
                uint64_t t[4] = {0, 0, 0, 0};
                memcpy(t, in + r * ncols + c0, cw * 8);
                a[r] = _mm256_loadu_si256((const V *)t);
            }
            /* INTT_n, decimation in frequency, inverse twiddles: natural in -> bit-reversed out */
            for (uint64_t len = n; len >= 2; len >>= 1) {
                const uint64_t half = len / 2, ts = n / len;
                for (uint64_t i = 0; i < n; i += len)
                    for (uint64_t j = 0; j < half; j++) {
                        V u = a[i + j], v = canon(a[i + j + half]);
                        a[i + j] = add_w(u, v);
                        V d = sub_w(u, v);
                        a[i + j + half] = j ? mul_w(d, vset(winv[j * ts])) : d;
                    }
            }
            /* scale and spread: position i (bit-reversed index of coefficient k) -> i * step of the extended bit-reversed input */
            for (uint64_t i = n; i-- > 0;) {
                V v = mul_w(a[i], vset(sc[i]));
                for (uint64_t z = 1; z < step; z++) a[i * step + z] = _mm256_setzero_si256();
                a[i * step] = v;
            }
            /* NTT_ext, decimation in time: bit-reversed in -> natural out */
            for (uint64_t len = 2; len <= n_ext; len <<= 1) {
                const uint64_t half = len / 2, ts = n_ext / len;
                for (uint64_t i = 0; i < n_ext; i += len)
                    for (uint64_t j = 0; j < half; j++) {
                        V t = canon(j ? mul_w(a[i + j + half], vset(wext[j * ts])) : a[i + j + half]);
                        V u = a[i + j];
                        a[i + j] = add_w(u, t);
                        a[i + j + half] = sub_w(u, t);
                    }
            }
            for (uint64_t r = 0; r < n_ext; r++) {
                uint64_t t[4];
                _mm256_storeu_si256((V *)t, canon(a[r]));
                memcpy(out + r * ncols + c0, t, cw * 8);
            }
        }
        free(a);
    }
    free(winv); free(wext); free(sc);
}
