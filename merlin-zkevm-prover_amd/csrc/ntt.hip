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

// MI_NTT_NO_ARITH: A/B diagnosis build -- the passes move the same bytes through the same LDS round trips and barriers, the
// field arithmetic (butterflies in ntt_math.h, twiddle multiplies, canonical form) is replaced by an xor that keeps the data flow
// MI_NTT_SETPRIO (A/B experiment, profiles/r03_pmc_ntt.txt): 1 = the waves of a workgroup that is issuing its tile loads run at a higher
// issue priority than the ones computing (their few instructions put 64 KB in flight), 2 = the opposite
#if defined(MI_NTT_SETPRIO) && MI_NTT_SETPRIO == 1
#define NTT_PRIO_LOAD() __builtin_amdgcn_s_setprio(3)
#define NTT_PRIO_COMPUTE() __builtin_amdgcn_s_setprio(0)
#elif defined(MI_NTT_SETPRIO) && MI_NTT_SETPRIO == 2
#define NTT_PRIO_LOAD() __builtin_amdgcn_s_setprio(0)
#define NTT_PRIO_COMPUTE() __builtin_amdgcn_s_setprio(3)
#else
#define NTT_PRIO_LOAD() do { } while (0)
#define NTT_PRIO_COMPUTE() do { } while (0)
#endif
#ifdef MI_NTT_NO_ARITH
#define NTT_MULW(x, t) ((x) ^ (t))
#define NTT_CANON(x) (x)
#else
#ifdef MI_NTT_AB_NOMULW   // (class diagnosis builds: ntt_math.h)
#define NTT_MULW(x, t) ((x) ^ (t))
#else
#define NTT_MULW(x, t) gl::mul_w((x), (t))
#endif
#ifdef MI_NTT_AB_NOCANON
#define NTT_CANON(x) (x)
#else
#define NTT_CANON(x) gl::canon(x)
#endif
#endif

struct NttPass {
    const u64 *src;
    u64 *dst;
    uint64_t src_pitch, dst_pitch;
    uint64_t in_valid_rows; // input rows >= this are read as zero (zero-padded LDE input)
    uint32_t ncols;
    uint32_t log_n, log_K;
    uint32_t apply_scale;
    uint32_t unit_tw; // last pass of an unscaled transform: every inter-pass twiddle is 1, outputs are only canonicalised
    uint32_t weak_out; // not the last pass: the next pass accepts any encoding of a value, so the store skips the canonical form
    uint32_t tj_log, tcp_log; // TJ = beta rows per tile, TCP = padded columns per tile; TJ*TCP = B, or less when the
                              // transform has fewer than B/TCP beta rows: the surplus lanes of a tile row then own nothing
    uint32_t n_col_tiles;
    uint64_t n_tiles;
    const u64 *tw_lo, *tw_hi;
    uint32_t tw_lo_bits;
    const u64 *sc_lo, *sc_hi;
    uint32_t sc_lo_bits;
    const u64 *w256;
};

// column index of a lane that owns no element: fails every `col < ncols` / `col + 1 < ncols` guard
static constexpr uint32_t NO_COL = 0xFFFFFFF0u;

// 16 bytes that are only 8-byte aligned (row pitch 665 is odd): gfx950 serves them with one dwordx4 access
struct __attribute__((packed, aligned(8))) U64x2 { u64 x, y; };

// value held by the neighbouring lane (lane ^ 1): two full-rate DPP moves, no LDS
__device__ __forceinline__ u64 from_lane_xor1(u64 v)
{
    const int lo = __builtin_amdgcn_mov_dpp((int)(u32)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(u32)(v >> 32), 0xB1, 0xF, 0xF, true);
    return ((u64)(u32)hi << 32) | (u32)lo;
}

// ---- tile-level building blocks shared by the plain pass and the fused LDE middle pass.
// An r-point DFT over the rows of an LDS tile [r][B] is two in-register steps (LA + LB bits, LB <= 4):
template <int LOG_R>
struct TileRadix {
    static constexpr int R = 1 << LOG_R;
    static constexpr int LB = LOG_R >= 4 ? 4 : LOG_R; // second in-register step
    static constexpr int LA = LOG_R - LB;             // first in-register step (0..4 bits)
    static constexpr int RA = 1 << LA, RB = 1 << LB;
};

// step A: RA-point DFTs over the high bits, in place, then twiddle by w_r^(p' ka).  Caller syncs afterwards.
// NZ: only the first NZ of the RA inputs of a DFT can be non-zero (zero-padded first pass of the extended NTT): the
// others are neither read nor computed with -- the unrolled butterflies fold the constants away.
template <int LOG_R, bool INV, int LOG_B, int NZ = (1 << (LOG_R - (LOG_R >= 4 ? 4 : LOG_R)))>
__device__ __forceinline__ void tile_step_a(u64 *tile, const u64 *w256, uint32_t tid)
{
    using T = TileRadix<LOG_R>;
    constexpr int B = 1 << LOG_B, NTT_THREADS = 16 * B;
    if (T::LA > 0) {
        for (uint32_t item = tid; item < (uint32_t)T::RB * B; item += NTT_THREADS) {
            const uint32_t b = item & (B - 1), pp = item >> LOG_B;
            u64 x[T::RA];
#pragma unroll
            for (int i = 0; i < T::RA; i++) x[i] = i < NZ ? tile[(i * T::RB + pp) * B + b] : 0;
            nttm::dft_reg<T::LA, INV>(x);
#pragma unroll
            for (int ka = 1; ka < T::RA; ka++) {
                uint32_t idx = (pp * ka) << (8 - LOG_R);
                if (INV) idx = (256 - idx) & 255;
                x[ka] = NTT_MULW(x[ka], w256[idx]);
            }
#pragma unroll
            for (int ka = 0; ka < T::RA; ka++) tile[(ka * T::RB + pp) * B + b] = x[ka];
        }
    }
}

// step B for one work item (kap, b): RB-point DFT over the low bits; x[kb] is output k1 = kap + RA * kb
template <int LOG_R, bool INV, int LOG_B>
__device__ __forceinline__ void tile_step_b(const u64 *tile, uint32_t kap, uint32_t b, u64 (&x)[TileRadix<LOG_R>::RB])
{
    using T = TileRadix<LOG_R>;
    constexpr int B = 1 << LOG_B;
#pragma unroll
    for (int i = 0; i < T::RB; i++) x[i] = tile[(kap * T::RB + i) * B + b];
    nttm::dft_reg<T::LB, INV>(x);
}

// 16-byte stores of two adjacent columns: lanes b (even) and b+1 hold the same rows of two adjacent columns; they
// swap halves (DPP) so that the even lane owns rows kb = 0,2,4.. and the odd lane rows 1,3,5.. of BOTH columns.
// q_e points at (first output row of this item, first column of the pair); executed by every lane of the wave.
template <int RB>
__device__ __forceinline__ void store_pairs(u64 *q_e, uint64_t qstride, const u64 (&x)[RB], bool odd, bool pair_ok, bool lone_ok)
{
    u64 *qr = q_e + (odd ? qstride : 0); // this lane's rows: kb = 2 j + odd
    const uint64_t step = 2 * qstride;
#pragma unroll
    for (int j = 0; j < RB / 2; j++) {
        const u64 give = odd ? x[2 * j] : x[2 * j + 1];
        const u64 got = from_lane_xor1(give);
        U64x2 w;
        w.x = odd ? got : x[2 * j];
        w.y = odd ? x[2 * j + 1] : got;
        if (pair_ok) *reinterpret_cast<U64x2 *>(qr) = w;
        else if (lone_ok) { // last, unpaired column: the even lane writes both rows itself
            qr[0] = x[2 * j];
            qr[qstride] = x[2 * j + 1];
        }
        qr += step;
    }
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
        if (a.unit_tw) return;
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
    NTT_PRIO_LOAD();
    if (WIDE) { // lanes run along column PAIRS: 16 lanes cover a 256-byte row segment, 32 rows per sweep
        constexpr int NW = (R * (B / 2) + NTT_THREADS - 1) / NTT_THREADS; // 16-byte loads per thread (8 at r = 256)
        const uint32_t bp = (tid & (B / 2 - 1)) * 2, tj = bp >> a.tcp_log, c = bp & (TCP - 1);
        const uint32_t col = tj < TJ ? c0 + c : NO_COL; // TJ * TCP < B (few beta rows): the surplus lanes own nothing
        const u64 *p = a.src + (beta0 + tj) * a.src_pitch + col;
        ulonglong2 v[NW];
        // all of the tile's loads are in flight before the twiddle table (dependent L2 loads + multiplies) is built
        constexpr bool FULL_SWEEPS = (R * (B / 2)) % NTT_THREADS == 0; // every thread owns NW rows
        const bool plain = FULL_SWEEPS && a.in_valid_rows >= n && __builtin_amdgcn_ballot_w64(col + 1 >= a.ncols) == 0;
        if (plain) { // interior tile (wave-uniform): no guards, the row pointer advances by a constant stride
            const u64 *pr = p + (uint64_t)(tid / (B / 2)) * mK * a.src_pitch;
            const uint64_t rstride = (uint64_t)(NTT_THREADS / (B / 2)) * mK * a.src_pitch;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const U64x2 w = *reinterpret_cast<const U64x2 *>(pr);
                v[k] = make_ulonglong2(w.x, w.y);
                pr += rstride;
            }
        } else {
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
        const uint32_t col = tj < TJ ? c0 + c : NO_COL;
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
    NTT_PRIO_COMPUTE();

    // ---- step A (high bits, in place) and step B (low bits) + inter-pass twiddle / scale + store in natural order
    tile_step_a<LOG_R, INV, LOG_B>(tile, w256, tid);
    if (LA > 0) __syncthreads();
    for (uint32_t item = tid; item < (uint32_t)RA * B; item += NTT_THREADS) { // trip count is wave-uniform
        const uint32_t b = item & (B - 1), kap = item >> LOG_B;
        const uint32_t tj_raw = b >> a.tcp_log, c = b & (TCP - 1);
        const uint32_t tj = tj_raw < TJ ? tj_raw : 0; // surplus lanes (see the load) compute on zeros and store nothing
        const uint32_t col = tj_raw < TJ ? c0 + c : NO_COL;
        u64 x[RB];
        tile_step_b<LOG_R, INV, LOG_B>(tile, kap, b, x);
        const uint64_t beta = beta0 + tj;
        const uint64_t ip = beta >> a.log_K, kappa = beta & (K - 1);
        // output row of k1 = kap + RA*kb is ((ip << LOG_R) + k1) * K + kappa: a constant stride in kb
        const uint64_t qstride = ((uint64_t)RA << a.log_K) * a.dst_pitch;
        const u64 *t = tw + ((kap << a.tj_log) + tj);
        const uint32_t tstride = (uint32_t)RA << a.tj_log;
        if (a.unit_tw) {
#pragma unroll
            for (int kb = 0; kb < RB; kb++) x[kb] = NTT_CANON(x[kb]);
        } else {
#pragma unroll
            for (int kb = 0; kb < RB; kb++) x[kb] = NTT_MULW(x[kb], t[kb * tstride]);
            if (!a.weak_out) { // wave-uniform
#pragma unroll
                for (int kb = 0; kb < RB; kb++) x[kb] = NTT_CANON(x[kb]);
            }
        }
        if (WIDE && RB >= 2) {
            const bool odd = b & 1;
            const uint32_t col_e = col & ~1u; // first column of the pair
            u64 *q = a.dst + col_e + ((((ip << LOG_R) + kap) << a.log_K) + kappa) * a.dst_pitch;
            store_pairs<RB>(q, qstride, x, odd, col_e + 1 < a.ncols, !odd && col < a.ncols);
        } else if (col < a.ncols) {
            u64 *q = a.dst + col + ((((ip << LOG_R) + kap) << a.log_K) + kappa) * a.dst_pitch;
#pragma unroll
            for (int kb = 0; kb < RB; kb++) q[kb * qstride] = x[kb];
        }
    }
}

// ---- persistent, double-buffered form of the radix-256 pass (the one LDE route DESIGN left open).
// k_ntt_pass above: two 64 KB workgroups per CU, each loads -> syncs -> computes -> stores; the memory pipe idles while both compute and
// the ALUs idle while both load, and how the two drift against each other is the ~40 ms between max(movement, arithmetic) and the
// measured LDE (profiles/r03_pmc_ntt.txt).  Here ONE 512-thread workgroup per CU walks its share of the tiles with TWO tile buffers:
// the next tile's 64 KB are requested with LDS-DMA loads (global_load_lds_dwordx4: no registers, no LDS-write instructions, the data
// lands while the current tile is computed) at the top of an iteration and waited for -- one s_waitcnt vmcnt(0), by which time they
// have had a whole tile's arithmetic to arrive -- just before the current tile's stores are issued, so load and compute phases overlap
// by construction instead of by the luck of two workgroups' phases.  The LDS image is lane-linear per wave (4 rows x 256 B per
// instruction), which is the layout the steps already use.  Only interior work qualifies (full 32-element tile rows, no zero-padded
// input, 16-byte aligned source rows); everything else takes k_ntt_pass.
template <bool INV>
__global__ __launch_bounds__(512, 1) void k_ntt_pass_pers(const NttPass a)
{
    constexpr int LOG_R = 8, LOG_B = 5, B = 32, R = 256, RA = 16, RB = 16;
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    u64 *const tile0 = smem, *const tile1 = smem + R * B;
    u64 *const w256 = smem + 2 * R * B;
    u64 *const tw0 = w256 + 256, *const tw1 = tw0 + R; // [R] each: TJ == 1 (a tile row is 32 columns of one beta row)

    const uint32_t tid = threadIdx.x;
    const uint64_t n = 1ull << a.log_n;
    const uint64_t K = 1ull << a.log_K;
    const uint64_t mK = n >> LOG_R;
    if (tid < 256) w256[tid] = a.w256[tid];

    auto locate = [&](uint64_t v, uint64_t &beta0, uint32_t &c0) { // virtual block id -> tile, XCD-aware as in k_ntt_pass (n_tiles and the grid are multiples of 8)
        const uint64_t lt = (v & 7) * (a.n_tiles >> 3) + (v >> 3);
        beta0 = lt / a.n_col_tiles;
        c0 = (uint32_t)(lt % a.n_col_tiles) << 5;
    };
    // the inter-pass twiddle (and scale) of entry k1 = tid of a tile: the table reads are ISSUED with the tile's DMA and only consumed
    // after the wait that retires both, so their latency hides behind a tile's arithmetic like the DMA's
    struct TwRegs { u64 hi, lo, shi, slo; };
    auto tw_issue = [&](uint64_t beta, TwRegs &r) {
        r.hi = r.lo = r.shi = r.slo = 1;
        if (a.unit_tw || tid >= 256) return;
        const uint32_t k1 = tid;
        const uint64_t ip = beta >> a.log_K, kappa = beta & (K - 1);
        uint64_t ex = (ip * k1) << a.log_K;
        if (INV) ex = (n - ex) & (n - 1);
        r.hi = a.tw_hi[ex >> a.tw_lo_bits];
        r.lo = a.tw_lo[ex & ((1ull << a.tw_lo_bits) - 1)];
        if (a.apply_scale) {
            const uint64_t ro = (((ip << LOG_R) + k1) << a.log_K) + kappa;
            r.shi = a.sc_hi[ro >> a.sc_lo_bits];
            r.slo = a.sc_lo[ro & ((1ull << a.sc_lo_bits) - 1)];
        }
    };
    auto tw_finish = [&](u64 *tw, const TwRegs &r) {
        if (a.unit_tw || tid >= 256) return;
        u64 t = gl::mul(r.hi, r.lo); // (w^0 = hi[0] * lo[0] = 1 * 1)
        if (a.apply_scale) t = gl::mul(t, gl::mul(r.shi, r.slo));
        tw[tid] = t;
    };
    // a wave's instruction k fetches rows 4 w + 32 k .. + 3 of the tile: lane l -> row l / 16, 16-byte piece l % 16 = LDS bytes l * 16 on
    auto issue = [&](u64 *tile, uint64_t beta0, uint32_t c0) {
        const u64 *p = a.src + beta0 * a.src_pitch + c0 + (tid & 15) * 2 + (uint64_t)(tid >> 4) * mK * a.src_pitch;
        const uint64_t rstride = 32 * mK * a.src_pitch;
        u64 *wave_base = tile + (uint32_t)((tid >> 6) * 4) * B;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(wave_base + k * 32 * B), 16, 0, 0);
            p += rstride;
        }
    };

    uint64_t v = blockIdx.x;
    uint64_t beta0;
    uint32_t c0;
    TwRegs tr;
    locate(v, beta0, c0);
    tw_issue(beta0, tr);
    issue(tile0, beta0, c0);
    __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0): as an instruction the compiler sees (inline asm would leave it believing the loads pending) */
    tw_finish(tw0, tr);
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
    __builtin_amdgcn_s_barrier();
    for (uint32_t it = 0; v < a.n_tiles; it++, v += gridDim.x) {
        u64 *tile = (it & 1) ? tile1 : tile0, *tile_n = (it & 1) ? tile0 : tile1;
        u64 *tw = (it & 1) ? tw1 : tw0, *tw_n = (it & 1) ? tw0 : tw1;
        const uint64_t vn = v + gridDim.x;
        uint64_t bn = 0;
        uint32_t cn = 0;
        if (vn < a.n_tiles) { // the next tile: its table reads and its 64 KB request go out now and are waited for after this tile's arithmetic
            locate(vn, bn, cn);
            tw_issue(bn, tr);
            issue(tile_n, bn, cn);
        }
        asm volatile("" ::: "memory");
        // ---- step A (in place), step B, twiddle
        tile_step_a<LOG_R, INV, LOG_B>(tile, w256, tid);
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
        __builtin_amdgcn_s_barrier();
        const uint32_t b = tid & (B - 1), kap = tid >> LOG_B;
        const uint32_t col = c0 + b;
        u64 x[RB];
        tile_step_b<LOG_R, INV, LOG_B>(tile, kap, b, x);
        const uint64_t ip = beta0 >> a.log_K, kappa = beta0 & (K - 1);
        const uint64_t qstride = ((uint64_t)RA << a.log_K) * a.dst_pitch;
        const u64 *t = tw + kap;
        if (a.unit_tw) {
#pragma unroll
            for (int kb = 0; kb < RB; kb++) x[kb] = NTT_CANON(x[kb]);
        } else {
#pragma unroll
            for (int kb = 0; kb < RB; kb++) x[kb] = NTT_MULW(x[kb], t[kb * RA]);
            if (!a.weak_out) {
#pragma unroll
                for (int kb = 0; kb < RB; kb++) x[kb] = NTT_CANON(x[kb]);
            }
        }
        // the next tile (and its table words) have had this tile's arithmetic to arrive; the previous tile's stores are long gone
        __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0): as an instruction the compiler sees (inline asm would leave it believing the loads pending) */
        if (vn < a.n_tiles) tw_finish(tw_n, tr);
        {
            const bool odd = b & 1;
            const uint32_t col_e = col & ~1u;
            u64 *q = a.dst + col_e + ((((ip << LOG_R) + kap) << a.log_K) + kappa) * a.dst_pitch;
            store_pairs<RB>(q, qstride, x, odd, true, false);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
        __builtin_amdgcn_s_barrier(); // every wave has read this tile (its buffer is the target after next) and its share of the next has landed
        beta0 = bn;
        c0 = cn;
    }
}

// ---- fused middle pass of extendPol: the LAST pass of INTT_N (radix r1, with the shift^k / N scale) and the FIRST
// pass of NTT_Next on the zero-padded coefficients (radix r2 = blowup * r1) work on the same data: the INTT tile
// (kappa, column tile) produces coefficients k = k1 * K1 + kappa, k1 < r1, K1 = N / r1, and the NTT's first pass
// reads rows i1 * (Next / r2) + beta = i1 * K1 + beta, i.e. with beta = kappa exactly those coefficients as its
// rows i1 < r1 (rows r1.. are the zero padding).  Keeping them in LDS saves one write and one read of the
// N x cols coefficient matrix (2 of the 17 pass-volumes) and a launch.
struct LdeMid {
    const u64 *src;
    u64 *dst;
    uint64_t src_pitch, dst_pitch;
    uint32_t ncols, log_n1, log_n2;
    uint32_t weak_out; // the NTT has further passes: they accept any encoding, the store skips the canonical form
    uint32_t tj_log, tcp_log, n_col_tiles;
    uint64_t n_tiles;
    const u64 *sc_lo, *sc_hi; // shift^k / N over k < N
    uint32_t sc_lo_bits;
    const u64 *tw_lo, *tw_hi; // w_Next^e
    uint32_t tw_lo_bits;
    const u64 *w256;
};

template <int LOG_R1, int LOG_BLOW, int LOG_B>
__global__ __launch_bounds__(16 << LOG_B) void k_lde_mid(const LdeMid a)
{
    constexpr int LOG_R2 = LOG_R1 + LOG_BLOW;
    using T1 = TileRadix<LOG_R1>;
    using T2 = TileRadix<LOG_R2>;
    constexpr int B = 1 << LOG_B, NTT_THREADS = 16 * B, R1 = T1::R, R2 = T2::R;
    // rows i >= R1 of the NTT tile are the zero padding; step A reads rows ia * RB + pp, so with RA >= blowup exactly the
    // inputs ia >= RA / blowup of every DFT are zero
    constexpr bool PAD_KNOWN = (T2::RA >> LOG_BLOW) >= 1;
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    u64 *tile = smem;           // [R2][B]
    u64 *w256 = tile + R2 * B;  // [256]
    u64 *tw1 = w256 + 256;      // [R1][TJ] scale of coefficient k1*K1 + kappa
    u64 *tw2 = tw1 + ((size_t)R1 << a.tj_log); // [R2][TJ] w_Next^(kappa * k1')

    const uint32_t tid = threadIdx.x;
    const uint32_t TJ = 1u << a.tj_log, TCP = 1u << a.tcp_log;
    const uint32_t log_K1 = a.log_n1 - LOG_R1;
    const uint64_t K1 = 1ull << log_K1;
    uint64_t lt = blockIdx.x;
    if ((a.n_tiles & 7) == 0) lt = (lt & 7) * (a.n_tiles >> 3) + (lt >> 3);
    const uint64_t kappa0 = (lt / a.n_col_tiles) << a.tj_log;
    const uint32_t c0 = (uint32_t)(lt % a.n_col_tiles) << a.tcp_log;

    if (tid < 256) w256[tid] = a.w256[tid];
    NTT_PRIO_LOAD();
    // ---- R1 input rows i1 * K1 + kappa, two adjacent columns per lane
    constexpr int NW = (R1 * (B / 2) + NTT_THREADS - 1) / NTT_THREADS;
    const uint32_t bp = (tid & (B / 2 - 1)) * 2, ltj = bp >> a.tcp_log, lc = bp & (TCP - 1);
    const uint32_t lcol = ltj < TJ ? c0 + lc : NO_COL; // TJ * TCP < B (few kappa rows): the surplus lanes own nothing
    ulonglong2 v[NW];
    {
        const u64 *p = a.src + (kappa0 + ltj) * a.src_pitch + lcol;
        constexpr bool FULL_SWEEPS = (R1 * (B / 2)) % NTT_THREADS == 0; // every thread owns NW rows
        if (FULL_SWEEPS && __builtin_amdgcn_ballot_w64(lcol + 1 >= a.ncols) == 0) { // interior tile (wave-uniform)
            const u64 *pr = p + (uint64_t)(tid / (B / 2)) * K1 * a.src_pitch;
            const uint64_t rstride = (uint64_t)(NTT_THREADS / (B / 2)) * K1 * a.src_pitch;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const U64x2 w = *reinterpret_cast<const U64x2 *>(pr);
                v[k] = make_ulonglong2(w.x, w.y);
                pr += rstride;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const uint32_t i1 = tid / (B / 2) + k * (NTT_THREADS / (B / 2));
                const u64 *pr = p + (uint64_t)i1 * K1 * a.src_pitch;
                v[k] = make_ulonglong2(0, 0);
                if (i1 < (uint32_t)R1) {
                    if (lcol + 1 < a.ncols) {
                        const U64x2 w = *reinterpret_cast<const U64x2 *>(pr);
                        v[k] = make_ulonglong2(w.x, w.y);
                    } else if (lcol < a.ncols) {
                        v[k].x = pr[0];
                    }
                }
            }
        }
    }
    for (uint32_t e = tid; e < (uint32_t)R1 * TJ; e += NTT_THREADS) { // scale of coefficient k = k1 * K1 + kappa
        const uint64_t k = ((uint64_t)(e >> a.tj_log) << log_K1) + kappa0 + (e & (TJ - 1));
        tw1[e] = gl::mul(a.sc_hi[k >> a.sc_lo_bits], a.sc_lo[k & ((1ull << a.sc_lo_bits) - 1)]);
    }
    for (uint32_t e = tid; e < (uint32_t)R2 * TJ; e += NTT_THREADS) { // w_Next^(kappa * k1')
        const uint64_t ex = (kappa0 + (e & (TJ - 1))) * (uint64_t)(e >> a.tj_log); // < K1 * R2 = Next
        tw2[e] = ex ? gl::mul(a.tw_hi[ex >> a.tw_lo_bits], a.tw_lo[ex & ((1ull << a.tw_lo_bits) - 1)]) : 1;
    }
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint32_t i1 = tid / (B / 2) + k * (NTT_THREADS / (B / 2));
        if (i1 < (uint32_t)R1) *reinterpret_cast<ulonglong2 *>(&tile[i1 * B + bp]) = v[k];
    }
    __syncthreads();
    NTT_PRIO_COMPUTE();

    // ---- INTT over the R1 rows; results (times scale) go back to rows k1 in natural order, rows R1.. are zeroed
    tile_step_a<LOG_R1, true, LOG_B>(tile, w256, tid);
    if (T1::LA > 0) __syncthreads();
    {
        const bool has_item = tid < (uint32_t)T1::RA * B; // RA1 * B <= 16 * B = threads: at most one item per thread
        const uint32_t b = tid & (B - 1), kap = tid >> LOG_B, tj_raw = b >> a.tcp_log;
        const uint32_t tj = tj_raw < TJ ? tj_raw : 0;
        u64 x[T1::RB];
        if (has_item) {
            tile_step_b<LOG_R1, true, LOG_B>(tile, kap, b, x);
#pragma unroll
            for (int kb = 0; kb < T1::RB; kb++) x[kb] = NTT_MULW(x[kb], tw1[((kap + T1::RA * kb) << a.tj_log) + tj]);
        }
        __syncthreads(); // every read of the INTT input rows is done
        if (has_item) {
#pragma unroll
            for (int kb = 0; kb < T1::RB; kb++) tile[(kap + T1::RA * kb) * B + b] = x[kb];
        }
        // the zero padding: rows R1.. are never read when step A knows they are zero
        if (!PAD_KNOWN)
            for (uint32_t e = tid; e < (uint32_t)(R2 - R1) * B; e += NTT_THREADS) tile[R1 * B + e] = 0;
    }
    __syncthreads();

    // ---- first pass of NTT_Next: rows kappa * R2 + k1' of the destination
    tile_step_a<LOG_R2, false, LOG_B, PAD_KNOWN ? (T2::RA >> LOG_BLOW) : T2::RA>(tile, w256, tid);
    if (T2::LA > 0) __syncthreads();
    for (uint32_t item = tid; item < (uint32_t)T2::RA * B; item += NTT_THREADS) {
        const uint32_t b = item & (B - 1), kap = item >> LOG_B;
        const uint32_t tj_raw = b >> a.tcp_log, c = b & (TCP - 1);
        const uint32_t tj = tj_raw < TJ ? tj_raw : 0;
        const uint32_t col = tj_raw < TJ ? c0 + c : NO_COL;
        u64 x[T2::RB];
        tile_step_b<LOG_R2, false, LOG_B>(tile, kap, b, x);
#pragma unroll
        for (int kb = 0; kb < T2::RB; kb++) x[kb] = NTT_MULW(x[kb], tw2[((kap + T2::RA * kb) << a.tj_log) + tj]);
        if (!a.weak_out) { // a one-pass NTT: this is the result
#pragma unroll
            for (int kb = 0; kb < T2::RB; kb++) x[kb] = NTT_CANON(x[kb]);
        }
        const bool odd = b & 1;
        const uint32_t col_e = col & ~1u;
        u64 *q = a.dst + col_e + (((kappa0 + tj) << LOG_R2) + kap) * a.dst_pitch;
        store_pairs<T2::RB>(q, (uint64_t)T2::RA * a.dst_pitch, x, odd, col_e + 1 < a.ncols, !odd && col < a.ncols);
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
    MI_REQUIRE_1D_GRID(count);
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
    MI_REQUIRE(a.n_tiles < (1ull << 22), "NTT grid too large"); // 512 threads each: the grid must stay below 2^32 threads
    a.w256 = ctx->w256;
    const size_t lds = (((size_t)(1u << log_r) << log_b) + 256 + ((size_t)(1u << log_r) << tj_log)) * 8;
    // the persistent double-buffered form (MI_NTT_PERSISTENT=1; A/B against the default in tools/pmc_ntt.sh): interior work only
    static const int persistent = []() { const char *e = getenv("MI_NTT_PERSISTENT"); return e ? atoi(e) : 0; }();
    const uint64_t grid_p = (uint64_t)ctx->cu_count & ~7ull;
    if (persistent && log_r == 8 && log_b == 5 && tcp_log == 5 && tj_log == 0 && (a.ncols & 31) == 0 && a.in_valid_rows >= n && ((uintptr_t)a.src & 15) == 0 &&
        (a.src_pitch & 1) == 0 && (a.n_tiles & 7) == 0 && a.n_tiles >= 4 * grid_p && grid_p >= 8) {
        const size_t lds_p = ((size_t)2 * 256 * 32 + 256 + 2 * 256) * 8;
        auto kf = k_ntt_pass_pers<false>;
        auto ki = k_ntt_pass_pers<true>;
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        if (inv) hipLaunchKernelGGL(ki, dim3((unsigned)grid_p), dim3(512), lds_p, ctx->stream, a);
        else hipLaunchKernelGGL(kf, dim3((unsigned)grid_p), dim3(512), lds_p, ctx->stream, a);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
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
// Only passes ps_begin <= ps < ps_end are launched (the fused LDE middle pass replaces one at either end).
static int run_passes(mi_ctx *ctx, NttPlan *plan, const std::vector<Buf> &bufs, uint64_t ncols, bool inv,
                      uint64_t in_valid_rows, const PowTable *scale, uint32_t ps_begin = 0, uint32_t ps_end = ~0u)
{
    const uint32_t L = plan->log_n;
    const uint32_t P = (uint32_t)bufs.size() - 1;
    uint32_t log_K = 0;
    for (uint32_t ps = 0; ps < P; ps++) {
        const uint32_t log_r = L / P + (ps < L % P ? 1 : 0);
        if (ps < ps_begin || ps >= ps_end) {
            log_K += log_r;
            continue;
        }
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
        a.unit_tw = (ps == P - 1 && !scale) ? 1 : 0; // ip = 0 in every tile of the last pass: exponent 0
        a.weak_out = ps == P - 1 ? 0 : 1;
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
static uint32_t pass_log_r(uint32_t L, uint32_t P, uint32_t ps) { return L / P + (ps < L % P ? 1 : 0); }

// Widest column chunk whose passes stay inside the launch limit (n_tiles < 2^22 workgroups): a pass over a chunk of
// at least one full tile width has TJ = 1, i.e. (n >> log_r) row tiles per column tile; the smallest radix of the split
// gives the most.  0: even a single column tile exceeds the limit (n > 2^29).
static uint64_t grid_chunk_cap(uint32_t L, uint32_t log_b)
{
    if (L == 0) return ~0ull;
    const uint32_t P = num_passes(L), min_r = L / P;
    const uint64_t row_tiles = 1ull << (L - min_r);
    const uint64_t col_tiles = ((1ull << 22) - 1) / row_tiles;
    return col_tiles << log_b;
}

template <int LOG_R1, int LOG_BLOW>
static int launch_lde_mid_t(mi_ctx *ctx, const LdeMid &a, size_t lds)
{
    auto k = k_lde_mid<LOG_R1, LOG_BLOW, 5>;
    if (lds > 48 * 1024) MI_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)a.n_tiles), dim3(512), lds, ctx->stream, a);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// The fused middle pass exists for tile rows of 32 elements holding at least one column pair, radix 16..128 on the
// INTT side and blowup 2 or 4, when the two transforms' pass splits line up (r2 = blowup * r1).
static bool lde_mid_applies(const mi_ctx *ctx, uint32_t L1, uint32_t L2, uint64_t ncols)
{
    if (!ctx->lde_fuse_mid || ctx->ntt_log_b != 5 || ncols < 2 || L1 == 0 || L2 <= L1 || L2 - L1 > 2) return false;
    const uint32_t P1 = num_passes(L1), P2 = num_passes(L2);
    const uint32_t r1 = pass_log_r(L1, P1, P1 - 1), r2 = pass_log_r(L2, P2, 0);
    return r1 >= 4 && r2 == r1 + (L2 - L1) && r2 <= 8;
}

static int launch_lde_mid(mi_ctx *ctx, NttPlan *p1, NttPlan *p2, const Buf &src, const Buf &dst, uint64_t ncols)
{
    const uint32_t L1 = p1->log_n, L2 = p2->log_n, P1 = num_passes(L1);
    const uint32_t log_r1 = pass_log_r(L1, P1, P1 - 1), log_blow = L2 - L1;
    const uint64_t K1 = 1ull << (L1 - log_r1);
    LdeMid a = {};
    a.src = src.p;
    a.src_pitch = src.pitch;
    a.dst = dst.p;
    a.dst_pitch = dst.pitch;
    a.ncols = (uint32_t)ncols;
    a.log_n1 = L1;
    a.log_n2 = L2;
    a.weak_out = num_passes(L2) > 1 ? 1 : 0;
    uint32_t tcp_log = 0;
    while ((1u << tcp_log) < a.ncols && tcp_log < 5) tcp_log++;
    uint32_t tj_log = 5 - tcp_log;
    while ((1ull << tj_log) > K1) tj_log--;
    a.tcp_log = tcp_log;
    a.tj_log = tj_log;
    a.n_col_tiles = (a.ncols + (1u << tcp_log) - 1) >> tcp_log;
    a.n_tiles = (K1 >> tj_log) * a.n_col_tiles;
    MI_REQUIRE(a.n_tiles < (1ull << 22), "LDE grid too large"); // 512 threads each: the grid must stay below 2^32 threads
    a.sc_lo = p1->lde_scale.lo;
    a.sc_hi = p1->lde_scale.hi;
    a.sc_lo_bits = p1->lde_scale.lo_bits;
    a.tw_lo = p2->tw.lo;
    a.tw_hi = p2->tw.hi;
    a.tw_lo_bits = p2->tw.lo_bits;
    a.w256 = ctx->w256;
    const uint32_t log_r2 = log_r1 + log_blow;
    const size_t lds = (((size_t)(1u << log_r2) << 5) + 256 + ((size_t)(1u << log_r1) << tj_log) + ((size_t)(1u << log_r2) << tj_log)) * 8;
#define MID(R1, BL) if (log_r1 == R1 && log_blow == BL) return launch_lde_mid_t<R1, BL>(ctx, a, lds);
    MID(4, 1) MID(5, 1) MID(6, 1) MID(7, 1) MID(4, 2) MID(5, 2) MID(6, 2)
#undef MID
    mi_set_error("no fused LDE pass for this radix");
    return MI_ERR_INVALID;
}

__global__ __launch_bounds__(256) void k_copy_canon(u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch,
                                                    uint64_t nrows, uint32_t ncols)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nrows * ncols; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / ncols, c = i % ncols;
        dst[r * dst_pitch + c] = gl::canon(src[r * src_pitch + c]);
    }
}

int launch_copy_2d(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t nrows, uint64_t ncols);
static int copy_canon(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t nrows,
                      uint64_t ncols)
{
    const uint64_t tot = nrows * ncols;
    if (!tot) return MI_OK;
    hipLaunchKernelGGL(k_copy_canon, dim3(mi_grid_256(tot)), dim3(256), 0, ctx->stream, dst, dst_pitch, src,
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
    }
    const uint64_t cap = grid_chunk_cap(L, ctx->ntt_log_b);
    if (cap && cap < chunk) chunk = cap; // cap == 0: launch_pass reports "NTT grid too large"
    if (P > 1) MI_TRY(mi_ensure_workspace(ctx, chunk * 2 * n * 8));
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

// ------------------------------------------------------------------ EXPERIMENT (round 5, r04 next #4b): a radix-256 pass over COLUMN-MAJOR data
// x(row, col) = base[col * cpitch + row].  The tile is 256 strided groups x 32 CONSECUTIVE ROWS of one column (what a tile-major LDE needs:
// the last pass then writes 64-row tiles in 256-byte runs).  A workgroup keeps ITS 32 rows and walks over all the columns (the twiddle
// table of its rows is built once), and moves 16 bytes per lane by pairing adjacent rows.  tools/ntt_colmajor_probe.py measures it against the row-major pass of the same data; it is not
// part of the product's transforms.  Only passes with log_K >= 5 (not the first: there consecutive rows of a tile are 256 outputs apart).
struct NttPassCM {
    const u64 *src;
    u64 *dst;
    uint64_t src_cpitch, dst_cpitch;
    uint32_t ncols, log_n, log_K, apply_scale, unit_tw, weak_out;
    uint64_t n_tiles;
    const u64 *tw_lo, *tw_hi;
    uint32_t tw_lo_bits;
    const u64 *sc_lo, *sc_hi;
    uint32_t sc_lo_bits;
    const u64 *w256;
};

template <bool INV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ntt_pass_cm(const NttPassCM a) // (two workgroups per CU, like the row-major pass: at most 128 VGPRs)
{
    constexpr int LOG_R = 8, LOG_B = 5, B = 32, R = 256, RA = 16, RB = 16;
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    u64 *tile = smem;         // [R][B]
    u64 *w256 = tile + R * B; // [256]
    const uint32_t tid = threadIdx.x;
    const uint64_t n = 1ull << a.log_n, K = 1ull << a.log_K, mK = n >> LOG_R;
    uint64_t lt = blockIdx.x;
    if ((a.n_tiles & 7) == 0) lt = (lt & 7) * (a.n_tiles >> 3) + (lt >> 3);
    const uint64_t beta0 = lt << LOG_B;
    if (tid < 256) w256[tid] = a.w256[tid];
    // this thread's step-B item: row beta = beta0 + b, outputs k1 = kap + 16 kb.  With K >= 32 the 32 rows of a tile share i' = beta >> log_K,
    // so the inter-pass twiddle w_n^(i' k1 K) depends on k1 alone (and the inverse transform's scale 1 / n on nothing): ONE table of 256 per
    // workgroup, in LDS, for all of its columns -- the row-major pass builds the same table once per tile
    u64 *tw = w256 + 256; // [R]
    const uint32_t b = tid & (B - 1), kap = tid >> LOG_B;
    if (!a.unit_tw && tid < (uint32_t)R) {
        const uint64_t ip = beta0 >> a.log_K;
        uint64_t ex = (ip * tid) << a.log_K;
        if (INV) ex = (n - ex) & (n - 1);
        u64 v = 1;
        if (ex) v = gl::mul(a.tw_hi[ex >> a.tw_lo_bits], a.tw_lo[ex & ((1ull << a.tw_lo_bits) - 1)]);
        if (a.apply_scale) v = gl::mul(v, gl::mul(a.sc_hi[0], a.sc_lo[0])); // (a constant scale: 1 / n)
        tw[tid] = v;
    }
    // load item: row pair bp, bp + 1 of strided group i1 = tid / 16 + 32 k
    const uint32_t bp = (tid & 15) * 2, i1_0 = tid >> 4;
    const bool odd = b & 1;
    const uint64_t qstride = (uint64_t)RA << a.log_K; // rows between the outputs kb, kb + 1
    const uint64_t beta_e = beta0 + (b & ~1u); // the EVEN row of my pair (K >= 32: both rows of a pair have the same i')
    const uint64_t orow_e = ((((beta_e >> a.log_K) << LOG_R) + kap) << a.log_K) + (beta_e & (K - 1)); // its first output row (kb = 0)
    for (uint32_t c = 0; c < a.ncols; c++) {
        const u64 *p = a.src + (uint64_t)c * a.src_cpitch + beta0 + bp + (uint64_t)i1_0 * mK;
        ulonglong2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const U64x2 w = *reinterpret_cast<const U64x2 *>(p + (uint64_t)k * 32 * mK);
            v[k] = make_ulonglong2(w.x, w.y);
        }
        if (c) __syncthreads(); // the previous column's step B has read the tile
#pragma unroll
        for (int k = 0; k < 8; k++) *reinterpret_cast<ulonglong2 *>(&tile[(i1_0 + 32 * k) * B + bp]) = v[k];
        __syncthreads();
        tile_step_a<LOG_R, INV, LOG_B>(tile, w256, tid);
        __syncthreads();
        u64 x[RB];
        tile_step_b<LOG_R, INV, LOG_B>(tile, kap, b, x);
        if (a.unit_tw) {
#pragma unroll
            for (int kb = 0; kb < RB; kb++) x[kb] = NTT_CANON(x[kb]);
        } else {
#pragma unroll
            for (int kb = 0; kb < RB; kb++) x[kb] = NTT_MULW(x[kb], tw[kap + RA * kb]);
            if (!a.weak_out) {
#pragma unroll
                for (int kb = 0; kb < RB; kb++) x[kb] = NTT_CANON(x[kb]);
            }
        }
        // adjacent rows are adjacent addresses: the paired 16-byte stores of the row-major kernel, with "column pair" read as "row pair"
        store_pairs<RB>(a.dst + (uint64_t)c * a.dst_cpitch + orow_e, qstride, x, odd, true, false);
    }
}

// EXPERIMENT, continued: the FIRST radix-256 pass, ROW-major in (the caller's trace: 32 adjacent columns of one row per tile row, 16-byte
// loads as in k_ntt_pass) -> COLUMN-major out.  In the first pass (K = 1) the 256 outputs of a column are CONSECUTIVE rows beta * 256 + k1,
// but a thread holds 16 of them 16 apart for ONE column: the tile goes back through LDS as [column][k1] (pitch 257) and leaves as 2 KB
// runs, 16 bytes per lane, lanes along the rows.
template <bool INV>
__global__ __launch_bounds__(512) void k_ntt_first_rm2cm(const NttPassCM a, uint64_t src_pitch)
{
    constexpr int LOG_R = 8, LOG_B = 5, B = 32, R = 256, RA = 16, RB = 16, TP = R + 1;
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    u64 *tile = smem;           // [R][B], then [B][TP]
    u64 *w256 = tile + B * TP;  // [256]
    u64 *tw = w256 + 256;       // [R]
    const uint32_t tid = threadIdx.x;
    const uint64_t n = 1ull << a.log_n, mK = n >> LOG_R;
    const uint32_t n_col_tiles = (a.ncols + B - 1) / B;
    uint64_t lt = blockIdx.x;
    if ((a.n_tiles & 7) == 0) lt = (lt & 7) * (a.n_tiles >> 3) + (lt >> 3);
    const uint64_t beta = lt / n_col_tiles; // = i' (K = 1)
    const uint32_t c0 = (uint32_t)(lt % n_col_tiles) * B;
    if (tid < 256) {
        w256[tid] = a.w256[tid];
        uint64_t ex = beta * tid; // < n
        if (INV) ex = (n - ex) & (n - 1);
        tw[tid] = ex ? gl::mul(a.tw_hi[ex >> a.tw_lo_bits], a.tw_lo[ex & ((1ull << a.tw_lo_bits) - 1)]) : 1;
    }
    // load: lanes along column pairs, 32 rows per sweep (k_ntt_pass, WIDE)
    const uint32_t bp = (tid & 15) * 2, i1_0 = tid >> 4;
    const uint32_t col = c0 + bp;
    const u64 *p = a.src + (beta + (uint64_t)i1_0 * mK) * src_pitch + col;
    ulonglong2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v[k] = make_ulonglong2(0, 0);
        const u64 *pr = p + (uint64_t)k * 32 * mK * src_pitch;
        if (col + 1 < a.ncols) { const U64x2 w = *reinterpret_cast<const U64x2 *>(pr); v[k] = make_ulonglong2(w.x, w.y); }
        else if (col < a.ncols) v[k].x = pr[0];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) *reinterpret_cast<ulonglong2 *>(&tile[(i1_0 + 32 * k) * B + bp]) = v[k];
    __syncthreads();
    tile_step_a<LOG_R, INV, LOG_B>(tile, w256, tid);
    __syncthreads();
    const uint32_t b = tid & (B - 1), kap = tid >> LOG_B;
    u64 x[RB];
    tile_step_b<LOG_R, INV, LOG_B>(tile, kap, b, x);
#pragma unroll
    for (int kb = 0; kb < RB; kb++) x[kb] = NTT_MULW(x[kb], tw[kap + RA * kb]);
    __syncthreads(); // every thread has read its step-B inputs: the tile is rewritten as [column][k1]
#pragma unroll
    for (int kb = 0; kb < RB; kb++) tile[b * TP + kap + RA * kb] = x[kb];
    __syncthreads();
    // store: 32 columns x 128 row pairs; lanes along the row pairs of one column (2 KB runs)
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const uint32_t idx = it * 512 + tid, cc = idx >> 7, pr2 = (idx & 127) * 2;
        if (c0 + cc < a.ncols) {
            U64x2 w;
            w.x = tile[cc * TP + pr2];
            w.y = tile[cc * TP + pr2 + 1];
            *reinterpret_cast<U64x2 *>(a.dst + (uint64_t)(c0 + cc) * a.dst_cpitch + beta * R + pr2) = w;
        }
    }
}

// Probe entry (tools/ntt_colmajor_probe.py): the length-n transform of ncols contiguous columns (column c at src + c * n, dst + c * n;
// n = 2^24 or 2^16: radix-256 passes only), the first pass by the product's single-column kernel column by column, the others by
// k_ntt_pass_cm over all columns at once; *ms_cm = the time of those passes (HIP events).  Results equal mi_ntt_dev's per column.
extern "C" int mi_dbg_ntt_colmajor_dev(mi_ctx *ctx, uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t ncols, int inverse, float *ms_cm)
{
    if (!ctx) return MI_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    MI_HIP_CHECK(hipSetDevice(ctx->device));
    MI_REQUIRE(dst && src && ncols >= 1 && (n == (1ull << 24) || n == (1ull << 16)), "n = 2^16 or 2^24, at least one column");
    // inverse & 2: the SOURCE is row-major (n x ncols at pitch ncols) and the first pass is the experimental row-major -> column-major
    // kernel as well (timed with the others): the whole transform in the transposed form; the result stays column-major
    const bool rm_src = (inverse & 2) != 0;
    inverse &= 1;
    const uint32_t L = ilog2_u64(n), P = L / 8;
    NttPlan *plan;
    MI_TRY(mi_get_plan(ctx, L, &plan));
    const PowTable *scale = inverse ? &plan->inv_scale : nullptr;
    MI_TRY(mi_ensure_workspace(ctx, 2 * n * ncols * 8));
    u64 *W0 = ctx->workspace, *W1 = ctx->workspace + n * ncols;
    // pass 0, column by column, through the product's kernel (the single-column form: the same tile, twiddles per element)
    for (uint64_t c = 0; c < ncols && !rm_src; c++) {
        std::vector<Buf> bufs;
        bufs.push_back({const_cast<u64 *>((const u64 *)src) + c * n, 1});
        bufs.push_back({W0 + c * n, 1});
        for (uint32_t i = 2; i <= P; i++) bufs.push_back({nullptr, 1});
        MI_TRY(run_passes(ctx, plan, bufs, 1, inverse != 0, n, scale, 0, 1));
    }
    hipEvent_t e0, e1;
    MI_HIP_CHECK(hipEventCreate(&e0));
    MI_HIP_CHECK(hipEventCreate(&e1));
    MI_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    const size_t lds = ((size_t)256 * 32 + 256 + 256) * 8;
    auto kf = k_ntt_pass_cm<false>;
    auto ki = k_ntt_pass_cm<true>;
    MI_HIP_CHECK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MI_HIP_CHECK(hipFuncSetAttribute((const void *)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (rm_src) {
        NttPassCM a = {};
        a.src = (const u64 *)src; a.dst = W0; a.dst_cpitch = n; a.ncols = (uint32_t)ncols; a.log_n = L;
        a.tw_lo = plan->tw.lo; a.tw_hi = plan->tw.hi; a.tw_lo_bits = plan->tw.lo_bits;
        a.n_tiles = (n >> 8) * ((ncols + 31) / 32);
        MI_REQUIRE(a.n_tiles < (1ull << 22), "NTT grid too large");
        a.w256 = ctx->w256;
        const size_t lds1 = ((size_t)32 * 257 + 256 + 256) * 8;
        auto k1f = k_ntt_first_rm2cm<false>;
        auto k1i = k_ntt_first_rm2cm<true>;
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)k1f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)k1i, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        if (inverse) hipLaunchKernelGGL(k1i, dim3((unsigned)a.n_tiles), dim3(512), lds1, ctx->stream, a, ncols);
        else hipLaunchKernelGGL(k1f, dim3((unsigned)a.n_tiles), dim3(512), lds1, ctx->stream, a, ncols);
        MI_HIP_CHECK(hipGetLastError());
    }
    const u64 *cur = W0;
    for (uint32_t ps = 1; ps < P; ps++) {
        NttPassCM a = {};
        a.src = cur;
        a.dst = ps == P - 1 ? (u64 *)dst : (cur == W0 ? W1 : W0);
        a.src_cpitch = a.dst_cpitch = n;
        a.ncols = (uint32_t)ncols;
        a.log_n = L;
        a.log_K = 8 * ps;
        a.tw_lo = plan->tw.lo; a.tw_hi = plan->tw.hi; a.tw_lo_bits = plan->tw.lo_bits;
        a.apply_scale = (ps == P - 1 && scale) ? 1 : 0;
        a.unit_tw = (ps == P - 1 && !scale) ? 1 : 0;
        a.weak_out = ps == P - 1 ? 0 : 1;
        if (a.apply_scale) { a.sc_lo = scale->lo; a.sc_hi = scale->hi; a.sc_lo_bits = scale->lo_bits; }
        a.n_tiles = (n >> 8) >> 5;
        a.w256 = ctx->w256;
        if (inverse) hipLaunchKernelGGL(ki, dim3((unsigned)a.n_tiles), dim3(512), lds, ctx->stream, a);
        else hipLaunchKernelGGL(kf, dim3((unsigned)a.n_tiles), dim3(512), lds, ctx->stream, a);
        MI_HIP_CHECK(hipGetLastError());
        cur = a.dst;
    }
    MI_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    MI_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    MI_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_cm) *ms_cm = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
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
    const bool fuse = lde_mid_applies(ctx, L1, L2, ncols);
    // scratch per column: one n_ext ping-pong buffer + INTT intermediate (n) [+ coefficients (n) when not fused].
    // The caller's output doubles as the other ping-pong buffer -- unless it is a window of a wider matrix (row pitch
    // 5320 B for the 665-column trace: every 256-byte row segment straddles three 128-byte lines) and the workspace
    // has room for a second, compact n_ext buffer Y without narrowing the column chunks below two tiles: then only the
    // first read and the last write touch the strided matrices.
    const uint64_t per_col = ((fuse ? 1 : 2) * n + n_ext) * 8;
    auto chunk_for = [&](uint64_t pc) {
        uint64_t c = ctx->workspace_limit / pc;
        if (c == 0) c = 1;
        if (c >= 32) c &= ~31ull;
        const uint64_t cap = grid_chunk_cap(L2, ctx->ntt_log_b);
        if (cap && cap < c) c = cap;
        return c > ncols ? ncols : c;
    };
    uint64_t chunk = chunk_for(per_col);
    const uint64_t chunk_y = chunk_for(per_col + n_ext * 8);
    const bool use_y = out_pitch > chunk_y && chunk_y >= (ncols < 64 ? ncols : 64); // a chunk is a window narrower than the pitch
    if (use_y) chunk = chunk_y;
    MI_TRY(mi_ensure_workspace(ctx, chunk * (per_col + (use_y ? n_ext * 8 : 0))));
    for (uint64_t c0 = 0; c0 < ncols; c0 += chunk) {
        const uint64_t cw = (ncols - c0 < chunk) ? ncols - c0 : chunk;
        const Buf S = {const_cast<u64 *>(in) + c0, in_pitch}, Dout = {out + c0, out_pitch};
        u64 *w = ctx->workspace;
        const Buf T = {w, cw};                             // INTT intermediate, n rows
        const Buf X = {w + n * cw, cw};                    // n_ext-row ping-pong buffer
        const Buf C = {w + n * cw + n_ext * cw, cw};       // coefficients, n rows (unfused path only)
        const Buf Y = {C.p + (fuse ? 0 : n * cw), cw};     // second n_ext-row ping-pong buffer (use_y)
        const Buf D = use_y ? Y : Dout;                    // ping-pong partner of X: Y, or the output itself
        const Buf Dlo = {D.p, D.pitch};                    // its first n rows double as an INTT intermediate
        std::vector<Buf> b1, b2;
        b1.push_back(S);
        for (uint32_t i = 1; i < P1; i++) b1.push_back(((P1 - i) & 1) ? T : Dlo); // last intermediate is T
        b1.push_back(C);
        b2.push_back(C);
        for (uint32_t i = 1; i < P2; i++) b2.push_back(((P2 - i) & 1) ? X : D); // last intermediate is X
        b2.push_back(Dout);
        if (fuse) {
            // INTT passes 0..P1-2, [last INTT pass + first NTT pass] b1[P1-1] -> b2[1], NTT passes 1..P2-1.
            // b1[P1-1] is S or T and never overlaps b2[1] (X or D).
            MI_TRY(run_passes(ctx, p1, b1, cw, true, n, nullptr, 0, P1 - 1));
            MI_TRY(launch_lde_mid(ctx, p1, p2, b1[P1 - 1], b2[1], cw));
            MI_TRY(run_passes(ctx, p2, b2, cw, false, n_ext, nullptr, 1, P2));
            continue;
        }
        // ---- INTT_n with the shift^k / n scale folded into its last pass: S -> ... -> C
        if (L1 == 0) MI_TRY(copy_canon(ctx, C.p, C.pitch, S.p, S.pitch, 1, cw)); // n = 1: the coefficient is the value
        else MI_TRY(run_passes(ctx, p1, b1, cw, true, n, &p1->lde_scale));
        // ---- NTT_next of the zero-padded coefficients: C -> ... -> D
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
