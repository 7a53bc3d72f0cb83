// fri.hip -- FRI fold / transpose and the remaining device-friendly loops of Starks::genProof.
//
// Replaces: FRIProve::prove fold loop (friProve.cpp:44-108, polMulAxi :192-200, evalPol :201-217),
// FRIProve::getTransposed (:252-271), the step-4 split (starks.cpp:265-280), Starks::evmap
// (starks.cpp:555-668), Polinomial::batchInverse[Parallel] (polinomial.hpp:612-720), the x / x_n / x_2ns /
// LEv tables (starks.hpp:149-183, starks.cpp:305-323), xDivXSubXi (starks.cpp:350-365), ZhInv (zhInv.cpp).
// All of it is exact field arithmetic, so any evaluation order is bit-identical to the reference's.
#include "common.h"
#include "ntt_math.h"
#include "chelpers_acc.h"
#include <algorithm>

using gl::E3;

// One folded element per lane.  Group g gathers pol[i * 2^cur + g], i < nX (consecutive lanes read
// consecutive 24-byte elements: coalesced), INTT_nX in registers one extension component at a time,
// coefficients parked in LDS ([k][d][lane], conflict free), then Horner in y = sinv_g * special_x, which
// equals sum_k (c_k * sinv_g^k) * x^k of polMulAxi + evalPol.
template <int LOG_NX>
__global__ __launch_bounds__(64) void k_fri_fold(u64 *__restrict__ out, const u64 *__restrict__ pol, uint32_t cur_bits,
                                                 u64 sinv0, u64 wi, E3 x, u64 nx_inv, uint64_t g0, uint64_t g_end)
{
    constexpr int NX = 1 << LOG_NX;
    extern __shared__ __attribute__((aligned(16))) u64 smem[]; // [NX][3][64]
    const uint32_t lane = threadIdx.x;
    const uint64_t g = g0 + (uint64_t)blockIdx.x * 64 + lane; // outputs [g0, g_end): a rank's share when the fold is sharded
    const uint64_t pol2n = 1ull << cur_bits;
    if (g >= g_end) return;
#pragma unroll 1
    for (int d = 0; d < 3; d++) {
        u64 v[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) v[i] = gl::canon(pol[((uint64_t)i * pol2n + g) * 3 + d]);
        nttm::dft_reg<LOG_NX, true>(v);
#pragma unroll
        for (int k = 0; k < NX; k++) smem[(k * 3 + d) * 64 + lane] = gl::mul(v[k], nx_inv); // dft_reg output is weak; mul canonicalises
    }
    const u64 sinv = gl::mul(sinv0, gl::pow(wi, g));
    const E3 y = gl::e3_mul1(x, sinv);
    E3 acc = {{smem[((NX - 1) * 3 + 0) * 64 + lane], smem[((NX - 1) * 3 + 1) * 64 + lane], smem[((NX - 1) * 3 + 2) * 64 + lane]}};
#pragma unroll 1
    for (int k = NX - 2; k >= 0; k--) {
        const E3 c = {{smem[(k * 3 + 0) * 64 + lane], smem[(k * 3 + 1) * 64 + lane], smem[(k * 3 + 2) * 64 + lane]}};
        acc = gl::e3_add(gl::e3_mul(acc, y), c);
    }
    out[g * 3 + 0] = acc.v[0];
    out[g * 3 + 1] = acc.v[1];
    out[g * 3 + 2] = acc.v[2];
}

__global__ __launch_bounds__(256) void k_copy_canon_flat(u64 *dst, const u64 *src, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[i] = gl::canon(src[i]);
}

int launch_fri_fold(mi_ctx *ctx, u64 *out, const u64 *pol, unsigned prev_bits, unsigned cur_bits, unsigned nbits_ext,
                    const u64 x[3], uint64_t g0, uint64_t g_count)
{
    MI_REQUIRE(prev_bits >= cur_bits && nbits_ext >= prev_bits && prev_bits <= 40, "bad FRI step sizes");
    const unsigned lnx = prev_bits - cur_bits;
    MI_REQUIRE(lnx <= 6, "FRI reduction of more than 6 bits per step is not supported");
    const uint64_t pol2n = 1ull << cur_bits;
    MI_REQUIRE(g0 <= pol2n && g_count <= pol2n - g0, "output range beyond the folded polynomial");
    if (g_count == 0) return MI_OK;
    if (lnx == 0) { // friProve.cpp:82-85 (step 0 is a copy)
        MI_REQUIRE_1D_GRID(g_count * 3);
        hipLaunchKernelGGL(k_copy_canon_flat, dim3((unsigned)((g_count * 3 + 255) / 256)), dim3(256), 0, ctx->stream, out + g0 * 3,
                           pol + g0 * 3, g_count * 3);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
    u64 sinv0 = gl::inv(49); // Goldilocks::shift()^-1, squared once per bit already folded (friProve.cpp:143-147)
    for (unsigned j = 0; j < nbits_ext - prev_bits; j++) sinv0 = gl::mul(sinv0, sinv0);
    u64 w = 7277203076849721926ULL;
    for (unsigned i = prev_bits; i < 32; i++) w = gl::mul(w, w);
    const u64 wi = gl::inv(w);
    const u64 nx_inv = gl::inv(1ull << lnx);
    const E3 xe = {{gl::canon(x[0]), gl::canon(x[1]), gl::canon(x[2])}};
    const unsigned grid = (unsigned)((g_count + 63) / 64);
    const size_t lds = (size_t)(1u << lnx) * 3 * 64 * 8;
#define FOLD(Q)                                                                                                         \
    case Q:                                                                                                             \
        if (lds > 48 * 1024)                                                                                            \
            MI_HIP_CHECK(hipFuncSetAttribute((const void *)k_fri_fold<Q>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                             (int)lds));                                                                \
        hipLaunchKernelGGL(k_fri_fold<Q>, dim3(grid), dim3(64), lds, ctx->stream, out, pol, (uint32_t)cur_bits, sinv0,  \
                           wi, xe, nx_inv, g0, g0 + g_count);                                                           \
        break;
    switch (lnx) {
        FOLD(1) FOLD(2) FOLD(3) FOLD(4) FOLD(5) FOLD(6)
    }
#undef FOLD
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// aux[i*h + j] = pol[j*w + i]: 32x32 LDS tile of 24-byte elements, coalesced on both sides
__global__ __launch_bounds__(256) void k_fri_transpose(u64 *__restrict__ aux, const u64 *__restrict__ pol, uint64_t w,
                                                       uint64_t h)
{
    __shared__ u64 t[32][33][3];
    const uint64_t i0 = (uint64_t)blockIdx.x * 32, j0 = (uint64_t)blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint64_t j = j0 + r, i = i0 + tx;
        if (j < h && i < w) {
            const u64 *p = pol + (j * w + i) * 3;
            t[r][tx][0] = p[0]; t[r][tx][1] = p[1]; t[r][tx][2] = p[2];
        }
    }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint64_t i = i0 + r, j = j0 + tx;
        if (j < h && i < w) {
            u64 *q = aux + (i * h + j) * 3;
            q[0] = gl::canon(t[tx][r][0]); q[1] = gl::canon(t[tx][r][1]); q[2] = gl::canon(t[tx][r][2]);
        }
    }
}

int launch_fri_transpose(mi_ctx *ctx, u64 *aux, const u64 *pol, uint64_t degree, unsigned tbits)
{
    const uint64_t w = 1ull << tbits;
    MI_REQUIRE(w && degree % w == 0, "degree must be a multiple of 2^transpose_bits");
    const uint64_t h = degree / w;
    if (!degree) return MI_OK;
    dim3 grid((unsigned)((w + 31) / 32), (unsigned)((h + 31) / 32));
    MI_REQUIRE(grid.y < 65536u * 1024u, "transpose grid too large");
    hipLaunchKernelGGL(k_fri_transpose, grid, dim3(256), 0, ctx->stream, aux, pol, w, h);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ---- step-4 split: qq2 viewed as n_ext rows of qdeg*3; rows >= n are zero (calloc at starks.cpp:232)
__global__ __launch_bounds__(256) void k_q_split(u64 *__restrict__ qq2, const u64 *__restrict__ qq1, uint64_t n,
                                                 uint64_t n_ext, uint32_t qdeg, u64 shift_in)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; // ext element index in qq2
    if (e >= n_ext * qdeg) return;
    const uint64_t k = e / qdeg;
    const uint32_t p = (uint32_t)(e % qdeg);
    u64 o0 = 0, o1 = 0, o2 = 0;
    if (k < n) {
        const u64 s = gl::pow(shift_in, p);
        const u64 *src = qq1 + (p * n + k) * 3;
        o0 = gl::mul(src[0], s); o1 = gl::mul(src[1], s); o2 = gl::mul(src[2], s);
    }
    qq2[e * 3 + 0] = o0; qq2[e * 3 + 1] = o1; qq2[e * 3 + 2] = o2;
}

int launch_q_split(mi_ctx *ctx, u64 *qq2, const u64 *qq1, uint64_t n, uint64_t n_ext, unsigned qdeg)
{
    MI_REQUIRE(qdeg >= 1 && (uint64_t)qdeg * n <= n_ext, "qdeg * n must not exceed n_ext");
    const u64 shift_in = gl::pow(gl::inv(49), n);
    const uint64_t tot = n_ext * qdeg;
    MI_REQUIRE_1D_GRID(tot);
    hipLaunchKernelGGL(k_q_split, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, qq2, qq1, n, n_ext,
                       (uint32_t)qdeg, shift_in);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ---- evmap: a tall-skinny mat-vec.  Lane <-> evaluated polynomial (sorted by address on the host so
// neighbouring lanes read neighbouring columns of one LDE row), block <-> slice of rows; partial sums
// [slice][eval] are reduced by a second kernel.  Field addition is exact, so the order is free.
struct EvDesc {
    const u64 *ptr;
    uint64_t stride;
    uint32_t dim;
    uint32_t prime;
    uint32_t out_index;
    uint32_t pad;
};

__global__ __launch_bounds__(256) void k_evmap_partial(u64 *__restrict__ partial, const EvDesc *__restrict__ desc, uint32_t n_act,
                                                       uint32_t n_evals, uint64_t n, uint32_t ext_bits, uint64_t rows_per_slice,
                                                       const u64 *__restrict__ lev, const u64 *__restrict__ lpev, uint64_t row0)
{
    // LEv / LpEv of 128 rows at a time through LDS: the same three words for every lane, which as vector loads cost an address-path slot
    // each (12 a four-row step against 4 for the polynomial values: the kernel was bound there, not by HBM or issue)
    constexpr uint32_t CH = 128, LW = CH * 3 + 2; // (+2: the two tables' rows fall into different banks)
    __shared__ u64 sL[2][LW];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t k0 = row0 + (uint64_t)blockIdx.y * rows_per_slice; // rows [row0, n) of the base domain (row0 > 0: a row shard's partial sums)
    uint64_t k1 = k0 + rows_per_slice;
    if (k1 > n) k1 = n;
    const bool active = i < n_act; // (the first n_act descriptors: the row-major polynomials; `partial` has n_evals per slice)
    EvDesc d = {nullptr, 0, 1, 0, 0, 0};
    if (active) d = desc[i];
    const u64 *sl = sL[d.prime ? 1 : 0];
    // the slice's sum of products unreduced in limb accumulators (chelpers_acc.h: a multiply-accumulate is 8 instructions, no reduction),
    // reduced once at the end: exact integer sums, so the evaluations are the same field elements
    chpa::Acc r0, r1, r2;
    chpa::acc_set(r0, 0); chpa::acc_set(r1, 0); chpa::acc_set(r2, 0);
    const uint64_t step = d.stride << ext_bits;
    for (uint64_t kc = k0; kc < k1; kc += CH) { // trip count is uniform over the workgroup
        const uint32_t cnt = (uint32_t)(k1 - kc < CH ? k1 - kc : CH);
        __syncthreads(); // the previous rows' readers are done
        for (uint32_t e = threadIdx.x; e < 2 * CH * 3; e += 256) {
            const uint32_t t = e >= CH * 3, j = e - t * CH * 3;
            sL[t][j] = j < cnt * 3 ? (t ? lpev : lev)[kc * 3 + j] : 0;
        }
        __syncthreads();
        if (!active) continue;
        const u64 *p = d.ptr + kc * step;
        if (d.dim == 1) {
            // four rows' loads issued together: a wave has one 512-byte row segment per load in flight, and latency, not bandwidth, is
            // what a one-load-at-a-time loop runs into
            uint32_t r = 0;
            for (; r + 4 <= cnt; r += 4, p += 4 * step) {
                const u64 v0 = p[0], v1 = p[step], v2 = p[2 * step], v3 = p[3 * step];
                const u64 *w = sl + r * 3;
                chpa::acc_mac(r0, v0, w[0]); chpa::acc_mac(r1, v0, w[1]); chpa::acc_mac(r2, v0, w[2]);
                chpa::acc_mac(r0, v1, w[3]); chpa::acc_mac(r1, v1, w[4]); chpa::acc_mac(r2, v1, w[5]);
                chpa::acc_mac(r0, v2, w[6]); chpa::acc_mac(r1, v2, w[7]); chpa::acc_mac(r2, v2, w[8]);
                chpa::acc_mac(r0, v3, w[9]); chpa::acc_mac(r1, v3, w[10]); chpa::acc_mac(r2, v3, w[11]);
            }
            for (; r < cnt; r++, p += step) {
                const u64 v = p[0];
                chpa::acc_mac(r0, v, sl[r * 3]);
                chpa::acc_mac(r1, v, sl[r * 3 + 1]);
                chpa::acc_mac(r2, v, sl[r * 3 + 2]);
            }
        } else {
            for (uint32_t r = 0; r < cnt; r++, p += step)
                chpa::acc_mul33(r0, r1, r2, p[0], p[1], p[2], sl[r * 3], sl[r * 3 + 1], sl[r * 3 + 2]);
        }
    }
    if (!active) return;
    u64 *o = partial + ((uint64_t)blockIdx.y * n_evals + i) * 3;
    o[0] = gl::canon(chpa::acc_reduce(r0)); o[1] = gl::canon(chpa::acc_reduce(r1)); o[2] = gl::canon(chpa::acc_reduce(r2));
}

// The same sums for polynomials that live in TILE-MAJOR sections ([tile of 64 rows][column][gl::tile_pos(row in tile)], Starks::genProof's
// extended sections): there a column's rows are contiguous and the rows evmap wants -- the multiples of 2^ext_bits -- are the first
// 64 >> ext_bits words of a tile's run, so a lane takes ROWS (64 consecutive rows of the base domain = the heads of 2^ext_bits tiles'
// runs, whole sectors) and a wave takes CG evaluations, each lane summing its rows' products; the lanes' sums meet at the end.
// LEv / LpEv of the slice are read by every group of evaluations: the groups of one slice are neighbours in the grid (x runs fastest),
// so the slice's 48 bytes per row stay in the L2s while they are wanted.  Slices start at multiples of 64 rows (launch_evmap).
// D3: the group's polynomials are extension-valued (three adjacent columns); base-field ones otherwise -- launched separately so that
// the many base-field ones do not carry three words per value in registers.
template <int CG, int UR, bool D3>
__global__ __launch_bounds__(256) void k_evmap_partial_tiled(u64 *__restrict__ partial, const EvDesc *__restrict__ desc, uint32_t i0, uint32_t n_t,
                                                            uint32_t n_evals, uint64_t n, uint32_t ext_bits, uint64_t rows_per_slice,
                                                            const u64 *__restrict__ lev, const u64 *__restrict__ lpev, uint64_t row0)
{
    // a step of the loop takes UR x 64 base rows: their LEv / LpEv words go through LDS once for the workgroup's four waves (4 x CG
    // evaluations), and every lane has UR x CG polynomial loads in flight (the kernel is bound by latency, not by bytes)
    constexpr uint32_t ROWS = UR * 64, VW = D3 ? 3 : 1;
    __shared__ u64 sL[2][ROWS * 3];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g0 = i0 + (blockIdx.x * 4 + wave) * CG;
    const uint64_t k0 = row0 + (uint64_t)blockIdx.y * rows_per_slice;
    uint64_t k1 = k0 + rows_per_slice;
    if (k1 > n) k1 = n;
    EvDesc d[CG];
    bool on[CG];
#pragma unroll
    for (int j = 0; j < CG; j++) {
        on[j] = g0 + j < i0 + n_t;
        d[j] = desc[on[j] ? g0 + j : i0];
    }
    chpa::Acc a[CG][3];
#pragma unroll
    for (int j = 0; j < CG; j++)
#pragma unroll
        for (int e = 0; e < 3; e++) chpa::acc_set(a[j][e], 0);
    // lane -> (tile tj of the 2^ext_bits tiles that 64 base rows spread over, word q of the tile's run): that word is row tile_pos(q)
    // of the tile (an involution), base row tile_pos(q) >> ext_bits of the tile's 64 >> ext_bits
    const uint32_t per_log = 6 - ext_bits, q = lane & ((1u << per_log) - 1), tj = lane >> per_log;
    const uint32_t kin = (tj << per_log) + (gl::tile_pos(q) >> ext_bits);
    const uint32_t q_idle = gl::tile_pos((uint32_t)((k0 << ext_bits) & 63));
    for (uint64_t kc = k0; kc < k1; kc += ROWS) { // trip count is uniform over the workgroup
        // the polynomial values first: their latency runs beside the LDS round of the weights
        u64 v[UR][CG][VW];
#pragma unroll
        for (uint32_t u = 0; u < UR; u++) {
            const uint64_t kb = kc + u * 64;
            const bool act = kb + kin < k1;
            const uint64_t toff = act ? (kb >> per_log) + tj : (k0 >> per_log); // the tile of the extension
            const uint32_t qq = act ? q : q_idle;
#pragma unroll
            for (int j = 0; j < CG; j++) {
                const u64 *p = d[j].ptr + toff * d[j].stride + qq;
#pragma unroll
                for (uint32_t e = 0; e < VW; e++) v[u][j][e] = p[64 * e];
            }
        }
        __syncthreads(); // the previous rows' readers are done
        for (uint32_t e = threadIdx.x; e < 2 * ROWS * 3; e += 256) {
            const uint32_t t = e >= ROWS * 3, j = e - t * ROWS * 3;
            sL[t][j] = kc * 3 + j < k1 * 3 ? (t ? lpev : lev)[kc * 3 + j] : 0; // rows past the slice's end weigh zero
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < UR; u++) {
            const u64 *l = &sL[0][(u * 64 + kin) * 3], *lp = &sL[1][(u * 64 + kin) * 3];
#pragma unroll
            for (int j = 0; j < CG; j++) {
                if (!on[j]) continue; // (wave-uniform)
                const u64 *w = d[j].prime ? lp : l;
                if (!D3) {
                    chpa::acc_mac(a[j][0], v[u][j][0], w[0]); chpa::acc_mac(a[j][1], v[u][j][0], w[1]); chpa::acc_mac(a[j][2], v[u][j][0], w[2]);
                } else chpa::acc_mul33(a[j][0], a[j][1], a[j][2], v[u][j][0], v[u][j][VW - 1 ? 1 : 0], v[u][j][VW - 1], w[0], w[1], w[2]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CG; j++) {
        if (!on[j]) continue;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            u64 x = gl::canon(chpa::acc_reduce(a[j][e]));
#pragma unroll
            for (int o = 32; o; o >>= 1) x = gl::add(x, (u64)__shfl_xor((unsigned long long)x, o));
            if (lane == 0) partial[((uint64_t)blockIdx.y * n_evals + g0 + j) * 3 + e] = x;
        }
    }
}

// evaluations [0, n_rm): the row-major kernel's s_rm slices; the others: the tile-major kernels' s_t slices.  A wave per evaluation, its
// lanes over the slices (up to 8 192 of them: one thread adding them up one after the other took 2 ms).
__global__ __launch_bounds__(64) void k_evmap_reduce(u64 *__restrict__ evals, const u64 *__restrict__ p_rm, uint32_t s_rm, uint32_t n_rm,
                                                     const u64 *__restrict__ p_t, uint32_t s_t, uint32_t n_t, const EvDesc *__restrict__ desc, uint32_t n_evals)
{
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_evals) return;
    const bool rm = i < n_rm;
    const u64 *partial = rm ? p_rm + (uint64_t)i * 3 : p_t + (uint64_t)(i - n_rm) * 3;
    const uint32_t n_slices = rm ? s_rm : s_t, pitch = rm ? n_rm : n_t;
    E3 acc = {{0, 0, 0}};
    for (uint32_t s = lane; s < n_slices; s += 64) {
        const u64 *p = partial + (uint64_t)s * pitch * 3;
        acc = gl::e3_add(acc, E3{{p[0], p[1], p[2]}});
    }
#pragma unroll
    for (int e = 0; e < 3; e++)
#pragma unroll
        for (int o = 32; o; o >>= 1) acc.v[e] = gl::add(acc.v[e], (u64)__shfl_xor((unsigned long long)acc.v[e], o));
    if (lane == 0) {
        u64 *o = evals + (uint64_t)desc[i].out_index * 3;
        o[0] = acc.v[0]; o[1] = acc.v[1]; o[2] = acc.v[2];
    }
}

int launch_evmap(mi_ctx *ctx, u64 *evals, uint64_t n_evals, uint64_t n_total, unsigned ext_bits, const u64 *const *pol_ptr,
                 const uint32_t *pol_dim, const u64 *pol_stride, const uint8_t *prime, const u64 *lev, const u64 *lpev, uint64_t row0, uint64_t nrows,
                 const uint64_t *tile_cols)
{
    if (n_evals == 0) return MI_OK;
    MI_REQUIRE(row0 + nrows <= n_total && nrows > 0, "rows outside the base domain");
    const uint64_t n = nrows; // the slices below cover [row0, row0 + nrows)
    MI_REQUIRE(n_evals < (1u << 24), "too many evaluations");
    // row-major polynomials first, then the ones in tile-major sections (tile_cols[i] = the section's width; pol_ptr[i] = its element of row 0)
    std::vector<EvDesc> d, dt;
    for (uint64_t i = 0; i < n_evals; i++) {
        MI_REQUIRE(pol_dim[i] == 1 || pol_dim[i] == 3, "polynomial dim must be 1 or 3");
        if (tile_cols && tile_cols[i]) {
            MI_REQUIRE(((n_total << ext_bits) & 63) == 0 && tile_cols[i] < (1ull << 32), "a tile-major section has a multiple of 64 rows");
            dt.push_back(EvDesc{pol_ptr[i], tile_cols[i] * 64, pol_dim[i], prime[i] ? 1u : 0u, (uint32_t)i, 0});
        } else d.push_back(EvDesc{pol_ptr[i], pol_stride[i], pol_dim[i], prime[i] ? 1u : 0u, (uint32_t)i, 0});
    }
    // order by address like the reference does (starks.cpp:560-607) -- here it buys coalescing
    auto by_ptr = [](const EvDesc &a, const EvDesc &b) { return a.ptr < b.ptr; };
    std::sort(d.begin(), d.end(), by_ptr);
    std::stable_sort(dt.begin(), dt.end(), [](const EvDesc &a, const EvDesc &b) { return a.dim != b.dim ? a.dim < b.dim : a.ptr < b.ptr; }); // base-field ones first
    const uint32_t n_rm = (uint32_t)d.size(), n_t = (uint32_t)dt.size();
    uint32_t n_t1 = 0;
    for (const EvDesc &e : dt) n_t1 += e.dim == 1;
    d.insert(d.end(), dt.begin(), dt.end());
    // slices of rows: the row-major kernel's workgroups take 256 evaluations each, so a STARK whose committed polynomials are tile-major
    // leaves it a few hundred (the constants, the quotient's chunks) and one workgroup per slice -- more, shorter slices then, or its
    // waves walk their rows one latency after the other (13 ms for the zkEVM's 230 at 1024 slices); the tile-major kernels' steps take
    // 128 base rows from a multiple of 64.  The two families keep their partial sums apart.
    uint32_t s_rm = (uint32_t)std::min<uint64_t>(n, 1024), s_t = 0;
    uint64_t rps_rm = (n + s_rm - 1) / s_rm, rps_t = 0;
    if (n_t) {
        MI_REQUIRE(row0 == 0 && ext_bits <= 6, "tile-major polynomials are summed over the whole domain");
        rps_t = ((n + 1023) / 1024 + 127) & ~127ull;
        s_t = (uint32_t)((n + rps_t - 1) / rps_t);
        if (n_rm && n_rm <= 1024 && n >= (1ull << 16)) {
            rps_rm = std::max<uint64_t>(256, n / (8192 / ((n_rm + 255) / 256)));
            s_rm = (uint32_t)((n + rps_rm - 1) / rps_rm);
        }
    }
    const uint64_t desc_bytes = (n_evals * sizeof(EvDesc) + 63) & ~63ull, part_rm = (uint64_t)s_rm * n_rm * 3 * 8, part_t = (uint64_t)s_t * n_t * 3 * 8; // [slice][evaluation of the family]
    char *scratch = nullptr;
    MI_TRY(mi_scratch(ctx, desc_bytes + part_rm + part_t + 64, (void **)&scratch));
    EvDesc *ddesc = (EvDesc *)scratch;
    u64 *p_rm = (u64 *)(scratch + desc_bytes), *p_t = (u64 *)(scratch + desc_bytes + part_rm);
    u64 *p_t0 = p_t - 3 * (uint64_t)n_rm; // the tile-major kernels index by descriptor: descriptor n_rm is the family's first
    MI_HIP_CHECK(hipMemcpyAsync(ddesc, d.data(), n_evals * sizeof(EvDesc), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // d is a stack-lifetime host buffer
    if (n_rm)
        hipLaunchKernelGGL(k_evmap_partial, dim3((n_rm + 255) / 256, s_rm), dim3(256), 0, ctx->stream, p_rm, ddesc, n_rm, n_rm, row0 + nrows,
                           (uint32_t)ext_bits, rps_rm, lev, lpev, row0);
    if (n_t1) { // (CG x UR = 4 x 2: 164 VGPRs, three waves per SIMD, 22.4 ms at zkEVM size; 4 x 4: 180 VGPRs, two waves, 27.2; 3 x 2: 128 VGPRs, 21.6; 2 x 4: 100 VGPRs, 23.6)
        constexpr int CG = 4, UR = 2;
        hipLaunchKernelGGL((k_evmap_partial_tiled<CG, UR, false>), dim3((n_t1 + 4 * CG - 1) / (4 * CG), s_t), dim3(256), 0, ctx->stream, p_t0, ddesc, n_rm, n_t1,
                           n_t, row0 + nrows, (uint32_t)ext_bits, rps_t, lev, lpev, row0);
    }
    if (n_t > n_t1) {
        constexpr int CG = 2, UR = 2;
        hipLaunchKernelGGL((k_evmap_partial_tiled<CG, UR, true>), dim3((n_t - n_t1 + 4 * CG - 1) / (4 * CG), s_t), dim3(256), 0, ctx->stream, p_t0, ddesc,
                           n_rm + n_t1, n_t - n_t1, n_t, row0 + nrows, (uint32_t)ext_bits, rps_t, lev, lpev, row0);
    }
    hipLaunchKernelGGL(k_evmap_reduce, dim3((unsigned)n_evals), dim3(64), 0, ctx->stream, evals, p_rm, s_rm, n_rm, p_t, s_t, n_t, ddesc, (uint32_t)n_evals);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ---- element-wise inverse in F_p^3 (res == src allowed): Polinomial::batchInverse / batchInverseParallel (polinomial.hpp:612-720).  As
// there, a run of elements shares one inversion (Montgomery's trick; here four elements of a thread, a grid stride apart); inv(0) = 0.
__global__ __launch_bounds__(256) void k_batch_inverse3(u64 *res, const u64 *src, uint64_t n)
{
    constexpr int XB = 4;
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    E3 d[XB], pre[XB], acc = {{1, 0, 0}};
    bool dead[XB];
#pragma unroll
    for (int j = 0; j < XB; j++) {
        const uint64_t i = t + (uint64_t)j * stride;
        d[j] = E3{{1, 0, 0}};
        if (i < n) d[j] = E3{{gl::canon(src[i * 3]), gl::canon(src[i * 3 + 1]), gl::canon(src[i * 3 + 2])}};
        dead[j] = (d[j].v[0] | d[j].v[1] | d[j].v[2]) == 0;
        if (dead[j]) d[j] = E3{{1, 0, 0}};
        pre[j] = acc;
        acc = gl::e3_mul(acc, d[j]);
    }
    E3 inv = gl::e3_inv(acc);
#pragma unroll
    for (int j = XB - 1; j >= 0; j--) {
        const uint64_t i = t + (uint64_t)j * stride;
        const E3 r = dead[j] ? E3{{0, 0, 0}} : gl::e3_mul(inv, pre[j]);
        inv = gl::e3_mul(inv, d[j]);
        if (i < n) { res[i * 3] = r.v[0]; res[i * 3 + 1] = r.v[1]; res[i * 3 + 2] = r.v[2]; }
    }
}

int launch_batch_inverse3(mi_ctx *ctx, u64 *res, const u64 *src, uint64_t n)
{
    if (!n) return MI_OK;
    const uint64_t threads = (n + 3) / 4;
    MI_REQUIRE_1D_GRID(threads);
    hipLaunchKernelGGL(k_batch_inverse3, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, res, src, n);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void k_geom_seq(u64 *out, uint64_t n, u64 start, u64 ratio)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = gl::mul(start, gl::pow(ratio, i));
}

int launch_geom_seq(mi_ctx *ctx, u64 *out, uint64_t n, u64 start, u64 ratio)
{
    if (!n) return MI_OK;
    MI_REQUIRE_1D_GRID(n);
    hipLaunchKernelGGL(k_geom_seq, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, out, n, gl::canon(start),
                       gl::canon(ratio));
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void k_geom_seq3(u64 *out, uint64_t n, E3 ratio)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    E3 r = {{1, 0, 0}}, b = ratio;
    for (uint64_t e = i; e; e >>= 1) {
        if (e & 1) r = gl::e3_mul(r, b);
        b = gl::e3_mul(b, b);
    }
    out[i * 3] = r.v[0]; out[i * 3 + 1] = r.v[1]; out[i * 3 + 2] = r.v[2];
}

int launch_geom_seq3(mi_ctx *ctx, u64 *out, uint64_t n, const u64 ratio[3])
{
    if (!n) return MI_OK;
    const E3 r = {{gl::canon(ratio[0]), gl::canon(ratio[1]), gl::canon(ratio[2])}};
    MI_REQUIRE_1D_GRID(n);
    hipLaunchKernelGGL(k_geom_seq3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, out, n, r);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// out[k] = x[k] * (x[k] - xi)^-1, x in the base field, xi in the extension.  A thread's four elements (a grid stride apart, so the
// accesses stay coalesced) share ONE inversion, Montgomery's trick as in the reference's batchInverse (polinomial.hpp:698-720): exact
// arithmetic, the same field elements; a zero denominator (inverse defined as 0) is taken out of the chain.
__global__ __launch_bounds__(256) void k_x_div_x_sub(u64 *out, const u64 *x, uint64_t n, E3 xi)
{
    constexpr int XB = 4;
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, stride = (uint64_t)gridDim.x * 256;
    u64 xv[XB];
    E3 d[XB], pre[XB], acc = {{1, 0, 0}};
    bool dead[XB];
#pragma unroll
    for (int j = 0; j < XB; j++) {
        const uint64_t i = t + (uint64_t)j * stride;
        xv[j] = i < n ? gl::canon(x[i]) : 0;
        d[j] = E3{{gl::sub(xv[j], xi.v[0]), gl::neg(xi.v[1]), gl::neg(xi.v[2])}};
        dead[j] = i >= n || (d[j].v[0] | d[j].v[1] | d[j].v[2]) == 0;
        if (dead[j]) d[j] = E3{{1, 0, 0}};
        pre[j] = acc;
        acc = gl::e3_mul(acc, d[j]);
    }
    E3 inv = gl::e3_inv(acc);
#pragma unroll
    for (int j = XB - 1; j >= 0; j--) {
        const uint64_t i = t + (uint64_t)j * stride;
        const E3 dinv = gl::e3_mul(inv, pre[j]);
        inv = gl::e3_mul(inv, d[j]);
        if (i < n) {
            const E3 r = dead[j] ? E3{{0, 0, 0}} : gl::e3_mul1(dinv, xv[j]);
            out[i * 3] = r.v[0]; out[i * 3 + 1] = r.v[1]; out[i * 3 + 2] = r.v[2];
        }
    }
}

int launch_x_div_x_sub(mi_ctx *ctx, u64 *out, const u64 *x, uint64_t n, const u64 xi[3])
{
    if (!n) return MI_OK;
    const E3 e = {{gl::canon(xi[0]), gl::canon(xi[1]), gl::canon(xi[2])}};
    const uint64_t threads = (n + 3) / 4;
    MI_REQUIRE_1D_GRID(threads);
    hipLaunchKernelGGL(k_x_div_x_sub, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, out, x, n, e);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void k_zhinv(u64 *out, uint64_t cnt, u64 sn, u64 w)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) out[i] = gl::inv(gl::sub(gl::mul(sn, gl::pow(w, i)), 1));
}

int launch_zhinv(mi_ctx *ctx, u64 *out, uint64_t cnt, u64 sn, u64 w)
{
    MI_REQUIRE_1D_GRID(cnt);
    hipLaunchKernelGGL(k_zhinv, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, out, cnt, sn, w);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void k_fill_synthetic(u64 *out, uint64_t count, u64 seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL; // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        out[i] = gl::canon(z);
    }
}

int launch_fill_synthetic(mi_ctx *ctx, u64 *out, uint64_t count, u64 seed)
{
    if (!count) return MI_OK;
    MI_REQUIRE(count < (1ull << 39), "count too large");
    hipLaunchKernelGGL(k_fill_synthetic, dim3(mi_grid_256(count)), dim3(256), 0, ctx->stream, out, count, seed);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void k_fill_synthetic_2d(u64 *out, uint64_t out_pitch, uint64_t nrows, uint32_t ncols,
                                                           uint64_t global_cols, uint64_t col0, u64 seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nrows * ncols; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / ncols, c = i % ncols;
        u64 z = seed + (r * global_cols + col0 + c + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        out[r * out_pitch + c] = gl::canon(z);
    }
}

int launch_fill_synthetic_2d(mi_ctx *ctx, u64 *out, uint64_t out_pitch, uint64_t nrows, uint64_t ncols, uint64_t global_cols,
                             uint64_t col0, u64 seed)
{
    const uint64_t tot = nrows * ncols;
    if (!tot) return MI_OK;
    MI_REQUIRE(tot < (1ull << 39) && ncols < (1ull << 31), "matrix too large");
    hipLaunchKernelGGL(k_fill_synthetic_2d, dim3(mi_grid_256(tot)), dim3(256), 0, ctx->stream, out, out_pitch, nrows,
                       (uint32_t)ncols, global_cols, col0, seed);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// test hook: out[i] = a[i] * b[i] and out[n + i] = a[i] + b[i] - b[i]^2 ... through the DEVICE arithmetic (the device
// build of gl_math.h differs from the host build: inline asm, wave-uniform branches)
__global__ __launch_bounds__(256) void k_dbg_field_ops(u64 *out, const u64 *a, const u64 *b, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 x = a[i], y = b[i];
    out[i] = gl::mul(x, y);                                                   // any u64 encodings in, canonical out
    out[n + i] = gl::canon(gl::add_wc(x, gl::canon(y)));                      // x + y
    out[2 * n + i] = gl::canon(gl::sub_wc(x, gl::canon(y)));                  // x - y
    out[3 * n + i] = gl::canon(nttm::mul_pow2<12>(nttm::mul_pow2<84>(x)));    // x * 2^96 = -x : both shift forms
    out[4 * n + i] = gl::canon(nttm::mul_pow2<40>(x));                        // 128-bit shift form
}

extern "C" int mi_dbg_field_ops_dev(mi_ctx *c, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n)
{
    if (!c) return MI_ERR_INVALID;
    if (!n) return MI_OK;
    MI_REQUIRE_1D_GRID(n);
    hipLaunchKernelGGL(k_dbg_field_ops, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (u64 *)out, (const u64 *)a,
                       (const u64 *)b, n);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// verification hook: out[r] = (accumulate ? out[r] : 0) + sum_c coef[c] * src[r][c] mod p -- one column that depends on EVERY column of the matrix.
// The LDE is linear, so LDE(lincomb(trace)) == lincomb(LDE(trace)) ties all columns at all rows of a full-size
// extension to one single-column transform the CPU oracle can redo.  One wave per row, lanes strided over the columns.
__global__ __launch_bounds__(256) void k_dbg_lincomb_cols(u64 *__restrict__ out, const u64 *__restrict__ src, uint64_t pitch,
                                                          uint64_t nrows, uint32_t ncols, const u64 *__restrict__ coef,
                                                          int accumulate)
{
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (uint64_t)gridDim.x * 4) {
        const u64 *p = src + r * pitch;
        u64 acc = 0;
        for (uint32_t c = lane; c < ncols; c += 64) acc = gl::add(acc, gl::mul(coef[c], p[c]));
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const u32 lo = (u32)__shfl_xor((int)(u32)acc, m, 64), hi = (u32)__shfl_xor((int)(u32)(acc >> 32), m, 64);
            acc = gl::add(acc, ((u64)hi << 32) | lo);
        }
        if (lane == 0) out[r] = accumulate ? gl::add(gl::canon(out[r]), acc) : acc;
    }
}

extern "C" int mi_dbg_lincomb_cols_dev(mi_ctx *c, uint64_t *out, const uint64_t *src, uint64_t pitch, uint64_t nrows,
                                       uint64_t ncols, const uint64_t *coef, int accumulate)
{
    if (!c) return MI_ERR_INVALID;
    if (!nrows) return MI_OK;
    MI_REQUIRE(out && src && coef && ncols < (1ull << 31) && pitch >= ncols, "bad arguments");
    const uint64_t blocks = (nrows + 3) / 4;
    hipLaunchKernelGGL(k_dbg_lincomb_cols, dim3((unsigned)(blocks < (1u << 20) ? blocks : (1u << 20))), dim3(256), 0, c->stream,
                       (u64 *)out, (const u64 *)src, pitch, nrows, (uint32_t)ncols, (const u64 *)coef, accumulate);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

extern "C" uint64_t mi_dbg_host_mul(uint64_t a, uint64_t b) { return gl::mul(a, b); }
extern "C" void mi_dbg_host_e3_mul(uint64_t out[3], const uint64_t a[3], const uint64_t b[3])
{
    const E3 r = gl::e3_mul(E3{{gl::canon(a[0]), gl::canon(a[1]), gl::canon(a[2])}}, E3{{gl::canon(b[0]), gl::canon(b[1]), gl::canon(b[2])}});
    out[0] = r.v[0]; out[1] = r.v[1]; out[2] = r.v[2];
}
extern "C" void mi_dbg_host_e3_inv(uint64_t out[3], const uint64_t a[3])
{
    const E3 r = gl::e3_inv(E3{{gl::canon(a[0]), gl::canon(a[1]), gl::canon(a[2])}});
    out[0] = r.v[0]; out[1] = r.v[1]; out[2] = r.v[2];
}
