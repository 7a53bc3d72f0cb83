// ntt.hip -- Goldilocks NTT / INTT / LDE over a ROW-MAJOR n x ncols matrix on gfx950.
//
// Replaces NTT_Goldilocks::{NTT, INTT, extendPol} (call sites starks.cpp:52,133,214,261,284,325-326).
// Semantics (build_const_tree.cpp:42-140,160-196): natural order in and out, INTT scaled by 1/n,
// extendPol = INTT_n -> coefficient k times shift^k -> zero pad -> NTT_next.
//
// Design (MI355X-first, not the reference's block/phase recursion):
//   * Stockham autosort passes of up to 8 bits each (a 2^24 transform = 3 passes).  Pass with radix
//     r = 2^LOG_R on the current length n_cur = n/K (K = product of earlier radices):
//         in  row = i1 * (m*K) + beta,          beta = i' * K + kappa in [0, m*K), m = n_cur / r
//         out row = ((i' * r + k1) * K) + kappa
//         y[k1] = w_ncur^(i' k1) * sum_i1 w_r^(i1 k1) x[i1]
//     so every pass READS r strided rows and WRITES natural order for the digits produced so far --
//     no bit-reversal pass, and after the last pass the rows are in natural order.
//   * One workgroup owns a tile of r rows x 32 "batch" elements (TJ consecutive beta rows x TCP columns,
//     TJ * TCP = 32), i.e. each touched row contributes a contiguous 256-byte segment of the row-major
//     trace.  The tile lives in LDS (r * 256 B <= 64 KiB); the r-point DFT is two in-register radix-16
//     steps (w_16 = 2^12, so twiddles inside a step are powers of two) with one LDS round trip between.
//   * The inter-pass twiddle w_ncur^(i' k1) -- and for the last INTT pass the 1/n or shift^k/n scale --
//     is looked up once per (k1, beta) from two-level 2 x 4096-entry tables and kept in LDS for all
//     32 columns of the tile.
//   * blockIdx -> tile mapping is XCD-aware: each XCD walks a contiguous run of tiles with the column
//     tiles of one row set adjacent, so the 128-byte lines that straddle two column tiles (row pitch
//     665 * 8 B is not line aligned) are served from that XCD's L2.
#include "common.h"
#include "ntt_math.h"


struct NttPass {
    const u64 *src;
    u64 *dst;
    uint64_t src_pitch, dst_pitch;
    uint64_t in_valid_rows; // input rows >= this are read as zero (zero-padded LDE input)
    uint32_t ncols;
    uint32_t log_n, log_K;
    uint32_t apply_scale;
    uint32_t tj_log, tcp_log; // TJ = beta rows per tile, TCP = padded columns per tile, TJ*TCP = 32
    uint32_t n_col_tiles;
    uint64_t n_tiles;
    const u64 *tw_lo, *tw_hi;
    uint32_t tw_lo_bits;
    const u64 *sc_lo, *sc_hi;
    uint32_t sc_lo_bits;
    const u64 *w256;
};

// 16 bytes that are only 8-byte aligned (row pitch 665 is odd): gfx950 serves them with one dwordx4 access
struct __attribute__((packed, aligned(8))) U64x2 { u64 x, y; };

// value held by the neighbouring lane (lane ^ 1): two full-rate DPP moves, no LDS
__device__ __forceinline__ u64 from_lane_xor1(u64 v)
{
    const int lo = __builtin_amdgcn_mov_dpp((int)(u32)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(u32)(v >> 32), 0xB1, 0xF, 0xF, true);
    return ((u64)(u32)hi << 32) | (u32)lo;
}

// WIDE: every lane moves two adjacent columns (16 bytes) per access on both the load and the store side.  With
// 8-byte accesses the pass is limited by the address/L1 path, not by HBM (measured on the movement alone:
// 211 ms -> 173 ms for the loads).  Needs at least two columns per tile row (TCP >= 2); single-column
// transforms use the narrow form.
template <int LOG_R, bool INV, int LOG_B, bool WIDE>
__global__ __launch_bounds__(16 << LOG_B) void k_ntt_pass(const NttPass a)
{
    constexpr int B = 1 << LOG_B;               // batch elements per tile row (B * 8 bytes contiguous in HBM)
    constexpr int NTT_THREADS = 16 * B;         // one radix-16 work item per thread and step at r = 256
    constexpr int R = 1 << LOG_R;
    constexpr int LB = LOG_R >= 4 ? 4 : LOG_R; // second in-register step
    constexpr int LA = LOG_R - LB;             // first in-register step (0..4 bits)
    constexpr int RA = 1 << LA, RB = 1 << LB;
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    u64 *tile = smem;          // [R][B]
    u64 *w256 = tile + R * B;  // [256]
    u64 *tw = w256 + 256;      // [R][TJ]

    const uint32_t tid = threadIdx.x;
    const uint32_t TJ = 1u << a.tj_log, TCP = 1u << a.tcp_log;
    const uint64_t n = 1ull << a.log_n;
    const uint64_t K = 1ull << a.log_K;
    const uint64_t mK = n >> LOG_R;

    // XCD-aware remap: hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD a
    // contiguous run of logical tiles (bijective only when the grid is a multiple of 8, else identity).
    uint64_t lt = blockIdx.x;
    if ((a.n_tiles & 7) == 0) lt = (lt & 7) * (a.n_tiles >> 3) + (lt >> 3);
    const uint64_t beta0 = (lt / a.n_col_tiles) << a.tj_log;
    const uint32_t c0 = (uint32_t)(lt % a.n_col_tiles) << a.tcp_log;

    if (tid < 256) w256[tid] = a.w256[tid];
    auto fill_tw = [&]() {
        for (uint32_t e = tid; e < (uint32_t)R * TJ; e += NTT_THREADS) {
            const uint32_t k1 = e >> a.tj_log, tj = e & (TJ - 1);
            const uint64_t beta = beta0 + tj;
            const uint64_t ip = beta >> a.log_K, kappa = beta & (K - 1);
            uint64_t ex = (ip * k1) << a.log_K; // exponent of w_n, < n
            if (INV) ex = (n - ex) & (n - 1);
            u64 t = 1;
            if (ex) t = gl::mul(a.tw_hi[ex >> a.tw_lo_bits], a.tw_lo[ex & ((1ull << a.tw_lo_bits) - 1)]);
            if (a.apply_scale) {
                const uint64_t ro = (((ip << LOG_R) + k1) << a.log_K) + kappa;
                t = gl::mul(t, gl::mul(a.sc_hi[ro >> a.sc_lo_bits], a.sc_lo[ro & ((1ull << a.sc_lo_bits) - 1)]));
            }
            tw[e] = t;
        }
    };

    // ---- load (any encoding; dft_reg accepts it)
    if (WIDE) { // lanes run along column PAIRS: 16 lanes cover a 256-byte row segment, 32 rows per sweep
        constexpr int NW = (R * (B / 2) + NTT_THREADS - 1) / NTT_THREADS; // 16-byte loads per thread (8 at r = 256)
        const uint32_t bp = (tid & (B / 2 - 1)) * 2, tj = bp >> a.tcp_log, c = bp & (TCP - 1);
        const uint32_t col = c0 + c;
        const u64 *p = a.src + (beta0 + tj) * a.src_pitch + col;
        ulonglong2 v[NW];
        // all of the tile's loads are in flight before the twiddle table (dependent L2 loads + multiplies) is built
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const uint32_t i1 = tid / (B / 2) + k * (NTT_THREADS / (B / 2));
            const uint64_t row = (uint64_t)i1 * mK + beta0 + tj;
            const u64 *pr = p + (uint64_t)i1 * mK * a.src_pitch;
            v[k] = make_ulonglong2(0, 0);
            if (i1 < (uint32_t)R && row < a.in_valid_rows) {
                if (col + 1 < a.ncols) {
                    const U64x2 w = *reinterpret_cast<const U64x2 *>(pr);
                    v[k] = make_ulonglong2(w.x, w.y);
                } else if (col < a.ncols) {
                    v[k].x = pr[0];
                }
            }
        }
        fill_tw();
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const uint32_t i1 = tid / (B / 2) + k * (NTT_THREADS / (B / 2));
            if (i1 < (uint32_t)R) *reinterpret_cast<ulonglong2 *>(&tile[i1 * B + bp]) = v[k];
        }
    } else { // lanes run along the B batch elements, 16 rows per sweep
        fill_tw();
        const uint32_t b = tid & (B - 1), tj = b >> a.tcp_log, c = b & (TCP - 1);
        const uint32_t col = c0 + c;
        const bool active = col < a.ncols;
        const u64 *p = a.src + (beta0 + tj) * a.src_pitch + col;
#pragma unroll 8
        for (uint32_t i1 = tid >> LOG_B; i1 < (uint32_t)R; i1 += NTT_THREADS / B) {
            const uint64_t row = (uint64_t)i1 * mK + beta0 + tj;
            u64 v = 0;
            if (active && row < a.in_valid_rows) v = p[(uint64_t)i1 * mK * a.src_pitch];
            tile[i1 * B + b] = v;
        }
    }
    __syncthreads();

    // ---- step A: RA-point DFTs over the high bits, in place, then twiddle by w_r^(p' ka)
    if (LA > 0) {
        for (uint32_t item = tid; item < (uint32_t)RB * B; item += NTT_THREADS) {
            const uint32_t b = item & (B - 1), pp = item >> LOG_B;
            u64 x[RA];
#pragma unroll
            for (int i = 0; i < RA; i++) x[i] = tile[(i * RB + pp) * B + b];
            nttm::dft_reg<LA, INV>(x);
#pragma unroll
            for (int ka = 1; ka < RA; ka++) {
                uint32_t idx = (pp * ka) << (8 - LOG_R);
                if (INV) idx = (256 - idx) & 255;
                x[ka] = gl::mul_w(x[ka], w256[idx]);
            }
#pragma unroll
            for (int ka = 0; ka < RA; ka++) tile[(ka * RB + pp) * B + b] = x[ka];
        }
        __syncthreads();
    }

    // ---- step B: RB-point DFTs over the low bits, inter-pass twiddle / scale, store in natural order
    for (uint32_t item = tid; item < (uint32_t)RA * B; item += NTT_THREADS) { // trip count is wave-uniform
        const uint32_t b = item & (B - 1), kap = item >> LOG_B;
        const uint32_t tj = b >> a.tcp_log, c = b & (TCP - 1);
        const uint32_t col = c0 + c;
        u64 x[RB];
#pragma unroll
        for (int i = 0; i < RB; i++) x[i] = tile[(kap * RB + i) * B + b];
        nttm::dft_reg<LB, INV>(x);
        const uint64_t beta = beta0 + tj;
        const uint64_t ip = beta >> a.log_K, kappa = beta & (K - 1);
        // output row of k1 = kap + RA*kb is ((ip << LOG_R) + k1) * K + kappa: a constant stride in kb
        const uint64_t qstride = ((uint64_t)RA << a.log_K) * a.dst_pitch;
        const u64 *t = tw + ((kap << a.tj_log) + tj);
        const uint32_t tstride = (uint32_t)RA << a.tj_log;
#pragma unroll
        for (int kb = 0; kb < RB; kb++) x[kb] = gl::mul(x[kb], t[kb * tstride]);
        if (WIDE && RB >= 2) {
            // lanes b (even) and b+1 hold the same rows of two adjacent columns: swap halves so that the even lane
            // owns rows kb = 0,2,4.. and the odd lane rows 1,3,5.. of BOTH columns, then store 16 bytes per lane
            const bool odd = b & 1;
            const uint32_t col_e = col & ~1u;                      // first column of the pair
            const bool pair_ok = col_e + 1 < a.ncols;
            u64 *q = a.dst + col_e + ((((ip << LOG_R) + kap) << a.log_K) + kappa) * a.dst_pitch;
#pragma unroll
            for (int j = 0; j < RB / 2; j++) {
                const u64 give = odd ? x[2 * j] : x[2 * j + 1];
                const u64 got = from_lane_xor1(give);              // executed by every lane (no divergence here)
                U64x2 w;
                w.x = odd ? got : x[2 * j];
                w.y = odd ? x[2 * j + 1] : got;
                u64 *qr = q + (uint64_t)(2 * j + (odd ? 1 : 0)) * qstride;
                if (pair_ok) *reinterpret_cast<U64x2 *>(qr) = w;
                else if (!odd && col < a.ncols) {                  // last, unpaired column: this lane writes both rows
                    q[(uint64_t)(2 * j) * qstride] = x[2 * j];
                    q[(uint64_t)(2 * j + 1) * qstride] = x[2 * j + 1];
                }
            }
        } else if (col < a.ncols) {
            u64 *q = a.dst + col + ((((ip << LOG_R) + kap) << a.log_K) + kappa) * a.dst_pitch;
#pragma unroll
            for (int kb = 0; kb < RB; kb++) q[kb * qstride] = x[kb];
        }
    }
}

// out[j] = s0 * g^(j * stride_exp)
__global__ __launch_bounds__(256) void k_fill_pow(u64 *out, uint64_t count, u64 s0, u64 g, uint64_t stride_exp)
{
    uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    out[j] = gl::mul(s0, gl::pow(gl::pow(g, stride_exp), j));
}

int launch_fill_pow(mi_ctx *ctx, u64 *out, uint64_t count, u64 s0, u64 g, uint64_t stride_exp)
{
    hipLaunchKernelGGL(k_fill_pow, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, out, count, s0, g,
                       stride_exp);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_make_pow_table(mi_ctx *ctx, PowTable *t, uint64_t count_log, u64 s0, u64 g)
{
    // covers exponents e < 2^count_log:  g^e * s0 = hi[e >> lo_bits] * lo[e & mask]
    t->lo_bits = (uint32_t)((count_log + 1) / 2);
    const uint64_t nlo = 1ull << t->lo_bits, nhi = 1ull << (count_log - t->lo_bits);
    MI_HIP_CHECK(hipMalloc((void **)&t->lo, nlo * 8));
    ctx->owned.push_back(t->lo);
    MI_HIP_CHECK(hipMalloc((void **)&t->hi, nhi * 8));
    ctx->owned.push_back(t->hi);
    MI_TRY(launch_fill_pow(ctx, t->lo, nlo, s0, g, 1));
    MI_TRY(launch_fill_pow(ctx, t->hi, nhi, 1, g, nlo));
    return MI_OK;
}

// w(n) = 7277203076849721926^(2^(32-n))   (pinned by tests/golden)
static u64 root_of_unity(uint32_t nbits)
{
    u64 r = 7277203076849721926ULL;
    for (uint32_t i = nbits; i < 32; i++) r = gl::mul(r, r);
    return r;
}

int mi_get_plan(mi_ctx *ctx, uint32_t log_n, NttPlan **plan)
{
    auto it = ctx->plans.find(log_n);
    if (it == ctx->plans.end()) {
        NttPlan p;
        p.log_n = log_n;
        const u64 w = root_of_unity(log_n);
        const u64 ninv = gl::inv((1ull << log_n) % GL_P);
        MI_TRY(mi_make_pow_table(ctx, &p.tw, log_n, 1, w));
        MI_TRY(mi_make_pow_table(ctx, &p.inv_scale, log_n, ninv, 1));
        MI_TRY(mi_make_pow_table(ctx, &p.lde_scale, log_n, ninv, 49 /* Goldilocks::shift() */));
        it = ctx->plans.emplace(log_n, p).first;
    }
    if (!ctx->w256) {
        MI_HIP_CHECK(hipMalloc((void **)&ctx->w256, 256 * 8));
        MI_TRY(launch_fill_pow(ctx, ctx->w256, 256, 1, root_of_unity(8), 1));
    }
    *plan = &it->second;
    return MI_OK;
}

template <int LOG_R, int LOG_B, bool WIDE>
static int launch_pass_rbw(mi_ctx *ctx, const NttPass &a, bool inv, size_t lds)
{
    auto kf = k_ntt_pass<LOG_R, false, LOG_B, WIDE>;
    auto ki = k_ntt_pass<LOG_R, true, LOG_B, WIDE>;
    if (lds > 48 * 1024) {
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (inv) hipLaunchKernelGGL(ki, dim3((unsigned)a.n_tiles), dim3(16 << LOG_B), lds, ctx->stream, a);
    else hipLaunchKernelGGL(kf, dim3((unsigned)a.n_tiles), dim3(16 << LOG_B), lds, ctx->stream, a);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

template <int LOG_R>
static int launch_pass_r(mi_ctx *ctx, const NttPass &a, bool inv, size_t lds, uint32_t log_b)
{
    const bool wide = a.tcp_log >= 1; // two adjacent columns per lane need at least two columns per tile row
    if (log_b == 4) return wide ? launch_pass_rbw<LOG_R, 4, true>(ctx, a, inv, lds) : launch_pass_rbw<LOG_R, 4, false>(ctx, a, inv, lds);
    return wide ? launch_pass_rbw<LOG_R, 5, true>(ctx, a, inv, lds) : launch_pass_rbw<LOG_R, 5, false>(ctx, a, inv, lds);
}

static int launch_pass(mi_ctx *ctx, NttPass &a, uint32_t log_r, bool inv)
{
    const uint64_t n = 1ull << a.log_n;
    const uint64_t mK = n >> log_r;
    // tile geometry: B = batch elements per tile row (16 or 32), TCP = columns per tile (power of two <= B),
    // TJ = B / TCP consecutive beta rows (<= mK)
    const uint32_t log_b = ctx->ntt_log_b;
    uint32_t tcp_log = 0;
    while ((1u << tcp_log) < a.ncols && tcp_log < log_b) tcp_log++;
    uint32_t tj_log = log_b - tcp_log;
    while ((1ull << tj_log) > mK) tj_log--;
    a.tcp_log = tcp_log;
    a.tj_log = tj_log;
    a.n_col_tiles = (a.ncols + (1u << tcp_log) - 1) >> tcp_log;
    a.n_tiles = (mK >> tj_log) * a.n_col_tiles;
    MI_REQUIRE(a.n_tiles < (1ull << 31), "NTT grid too large");
    a.w256 = ctx->w256;
    const size_t lds = (((size_t)(1u << log_r) << log_b) + 256 + ((size_t)(1u << log_r) << tj_log)) * 8;
    switch (log_r) {
    case 1: return launch_pass_r<1>(ctx, a, inv, lds, log_b);
    case 2: return launch_pass_r<2>(ctx, a, inv, lds, log_b);
    case 3: return launch_pass_r<3>(ctx, a, inv, lds, log_b);
    case 4: return launch_pass_r<4>(ctx, a, inv, lds, log_b);
    case 5: return launch_pass_r<5>(ctx, a, inv, lds, log_b);
    case 6: return launch_pass_r<6>(ctx, a, inv, lds, log_b);
    case 7: return launch_pass_r<7>(ctx, a, inv, lds, log_b);
    case 8: return launch_pass_r<8>(ctx, a, inv, lds, log_b);
    }
    mi_set_error("bad radix");
    return MI_ERR_INVALID;
}

struct Buf {
    u64 *p;
    uint64_t pitch;
};

// Runs the passes of one length-n transform.  bufs[0] = source, bufs[i] = destination of pass i.
static int run_passes(mi_ctx *ctx, NttPlan *plan, const std::vector<Buf> &bufs, uint64_t ncols, bool inv,
                      uint64_t in_valid_rows, const PowTable *scale)
{
    const uint32_t L = plan->log_n;
    const uint32_t P = (uint32_t)bufs.size() - 1;
    uint32_t log_K = 0;
    for (uint32_t ps = 0; ps < P; ps++) {
        const uint32_t log_r = L / P + (ps < L % P ? 1 : 0);
        NttPass a = {};
        a.src = bufs[ps].p;
        a.src_pitch = bufs[ps].pitch;
        a.dst = bufs[ps + 1].p;
        a.dst_pitch = bufs[ps + 1].pitch;
        a.in_valid_rows = ps == 0 ? in_valid_rows : (1ull << L);
        a.ncols = (uint32_t)ncols;
        a.log_n = L;
        a.log_K = log_K;
        a.tw_lo = plan->tw.lo;
        a.tw_hi = plan->tw.hi;
        a.tw_lo_bits = plan->tw.lo_bits;
        a.apply_scale = (ps == P - 1 && scale) ? 1 : 0;
        if (a.apply_scale) {
            a.sc_lo = scale->lo;
            a.sc_hi = scale->hi;
            a.sc_lo_bits = scale->lo_bits;
        }
        MI_TRY(launch_pass(ctx, a, log_r, inv));
        log_K += log_r;
    }
    return MI_OK;
}

static uint32_t num_passes(uint32_t L) { return L == 0 ? 0 : (L + 7) / 8; }

__global__ __launch_bounds__(256) void k_copy_canon(u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch,
                                                    uint64_t nrows, uint32_t ncols)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * ncols) return;
    uint64_t r = i / ncols, c = i % ncols;
    dst[r * dst_pitch + c] = gl::canon(src[r * src_pitch + c]);
}

int launch_copy_2d(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t nrows, uint64_t ncols);
static int copy_canon(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t nrows,
                      uint64_t ncols)
{
    const uint64_t tot = nrows * ncols;
    if (!tot) return MI_OK;
    hipLaunchKernelGGL(k_copy_canon, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dst, dst_pitch, src,
                       src_pitch, nrows, (uint32_t)ncols);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int launch_copy_2d(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t nrows, uint64_t ncols)
{
    MI_REQUIRE(nrows * ncols < (1ull << 39) && ncols < (1ull << 31), "matrix too large");
    return copy_canon(ctx, dst, dst_pitch, src, src_pitch, nrows, ncols);
}

int launch_ntt(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t n, uint64_t ncols,
               int inverse)
{
    if (n == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(is_pow2(n), "n must be a power of two");
    MI_REQUIRE(n <= (1ull << 32), "n too large");
    MI_REQUIRE(dst_pitch >= ncols && src_pitch >= ncols, "pitch smaller than ncols");
    const uint32_t L = ilog2_u64(n);
    if (L == 0) return copy_canon(ctx, dst, dst_pitch, src, src_pitch, 1, ncols);
    NttPlan *plan;
    MI_TRY(mi_get_plan(ctx, L, &plan));
    const uint32_t P = num_passes(L);
    const PowTable *scale = inverse ? &plan->inv_scale : nullptr;
    // column chunks sized to the workspace (a chunk needs up to 2 * n * cols of scratch)
    uint64_t chunk = ncols;
    if (P > 1) {
        const uint64_t per_col = 2 * n * 8;
        uint64_t fit = ctx->workspace_limit / per_col;
        if (fit == 0) fit = 1;
        if (fit >= 32) fit &= ~31ull;
        if (fit < chunk) chunk = fit;
        MI_TRY(mi_ensure_workspace(ctx, (chunk < ncols ? chunk : ncols) * per_col));
    }
    for (uint64_t c0 = 0; c0 < ncols; c0 += chunk) {
        const uint64_t cw = (ncols - c0 < chunk) ? ncols - c0 : chunk;
        const Buf S = {const_cast<u64 *>(src) + c0, src_pitch}, D = {dst + c0, dst_pitch};
        const Buf W0 = {ctx->workspace, cw}, W1 = {ctx->workspace ? ctx->workspace + n * cw : nullptr, cw};
        std::vector<Buf> bufs;
        bufs.push_back(S);
        // intermediates alternate W0 / W1 and never alias the caller's buffers, so dst == src is safe
        for (uint32_t i = 1; i < P; i++) bufs.push_back((i & 1) ? W0 : W1);
        bufs.push_back(D);
        MI_TRY(run_passes(ctx, plan, bufs, cw, inverse != 0, n, scale));
    }
    return MI_OK;
}

int launch_lde(mi_ctx *ctx, u64 *out, uint64_t out_pitch, const u64 *in, uint64_t in_pitch, uint64_t n_ext, uint64_t n,
               uint64_t ncols)
{
    if (n == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(is_pow2(n) && is_pow2(n_ext) && n_ext >= n, "sizes must be powers of two with n_ext >= n");
    MI_REQUIRE(n_ext <= (1ull << 32), "n_ext too large");
    MI_REQUIRE(out_pitch >= ncols && in_pitch >= ncols, "pitch smaller than ncols");
    MI_REQUIRE(out != in, "extendPol cannot run in place");
    const uint32_t L1 = ilog2_u64(n), L2 = ilog2_u64(n_ext);
    NttPlan *p1 = nullptr, *p2 = nullptr;
    if (L1) MI_TRY(mi_get_plan(ctx, L1, &p1));
    if (L2) MI_TRY(mi_get_plan(ctx, L2, &p2));
    if (L2 == 0) return copy_canon(ctx, out, out_pitch, in, in_pitch, 1, ncols);
    const uint32_t P1 = num_passes(L1), P2 = num_passes(L2);
    // scratch per column: coefficients (n) + one n_ext ping-pong buffer + INTT intermediate (n)
    const uint64_t per_col = (2 * n + n_ext) * 8;
    uint64_t chunk = ctx->workspace_limit / per_col;
    if (chunk == 0) chunk = 1;
    if (chunk >= 32) chunk &= ~31ull;
    if (chunk > ncols) chunk = ncols;
    MI_TRY(mi_ensure_workspace(ctx, chunk * per_col));
    for (uint64_t c0 = 0; c0 < ncols; c0 += chunk) {
        const uint64_t cw = (ncols - c0 < chunk) ? ncols - c0 : chunk;
        const Buf S = {const_cast<u64 *>(in) + c0, in_pitch}, D = {out + c0, out_pitch};
        u64 *w = ctx->workspace;
        const Buf C = {w, cw};                 // coefficients, n rows
        const Buf T = {w + n * cw, cw};        // INTT intermediate, n rows
        const Buf X = {w + 2 * n * cw, cw};    // n_ext-row ping-pong partner of D
        const Buf Dlo = {D.p, D.pitch};        // first n rows of the output double as an INTT intermediate
        // ---- INTT_n with the shift^k / n scale folded into its last pass: S -> ... -> C
        if (L1 == 0) {
            MI_TRY(copy_canon(ctx, C.p, C.pitch, S.p, S.pitch, 1, cw)); // n = 1: the coefficient is the value
        } else {
            std::vector<Buf> b1;
            b1.push_back(S);
            for (uint32_t i = 1; i < P1; i++) b1.push_back(((P1 - i) & 1) ? T : Dlo); // last intermediate is T
            b1.push_back(C);
            MI_TRY(run_passes(ctx, p1, b1, cw, true, n, &p1->lde_scale));
        }
        // ---- NTT_next of the zero-padded coefficients: C -> ... -> D
        std::vector<Buf> b2;
        b2.push_back(C);
        for (uint32_t i = 1; i < P2; i++) b2.push_back(((P2 - i) & 1) ? X : D); // last intermediate is X
        b2.push_back(D);
        MI_TRY(run_passes(ctx, p2, b2, cw, false, n, nullptr));
    }
    return MI_OK;
}

extern "C" void mi_dbg_host_dft16(uint64_t x[16], int log_size, int inverse)
{
    u64 v[16];
    for (int i = 0; i < 16; i++) v[i] = x[i];
#define RUN(Q)                                                                     \
    {                                                                              \
        u64 t[1 << Q];                                                             \
        for (int i = 0; i < (1 << Q); i++) t[i] = v[i];                            \
        if (inverse) nttm::dft_reg<Q, true>(t); else nttm::dft_reg<Q, false>(t);   \
        for (int i = 0; i < (1 << Q); i++) v[i] = t[i];                            \
    }
    switch (log_size) {
    case 0: break;
    case 1: RUN(1) break;
    case 2: RUN(2) break;
    case 3: RUN(3) break;
    case 4: RUN(4) break;
    }
#undef RUN
    for (int i = 0; i < 16; i++) x[i] = gl::canon(v[i]);
}
