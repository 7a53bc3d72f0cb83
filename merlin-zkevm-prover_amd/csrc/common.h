// common.h -- context, error plumbing and internal launcher prototypes of libmi_stark.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <vector>
#include <mutex>
#include "../../include/mi_stark.h"
#include "gl_math.h"

void mi_set_error(const char *fmt, ...);

#define MI_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            mi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MI_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define MI_REQUIRE(cond, msg)                                    \
    do {                                                         \
        if (!(cond)) {                                           \
            mi_set_error("%s: %s", __func__, msg);               \
            return MI_ERR_INVALID;                               \
        }                                                        \
    } while (0)

#define MI_TRY(expr)               \
    do {                           \
        int s_ = (expr);           \
        if (s_ != MI_OK) return s_; \
    } while (0)

// Blocks of 256 threads for an element-wise kernel over `total` elements.  A launch may not have 2^32 or more threads
// in one grid dimension (HIP truncates the global size: a 5.6e9-element fill silently covered 1.3e9 elements), so the
// grid is capped and the kernels below loop with stride gridDim.x * 256.
static inline unsigned mi_grid_256(uint64_t total)
{
    const uint64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks < (1ull << 22) ? blocks : (1ull << 22));
}

// one thread per element, no loop in the kernel: the element count must leave the grid below 2^32 threads
#define MI_REQUIRE_1D_GRID(total) MI_REQUIRE((uint64_t)(total) < (1ull << 32) - 1024, "too many elements for one launch")

// Column windows ("slabs") of the rows a leaf-hash launch absorbs, in order: window i holds `width[i]` consecutive
// columns of every row, row r at base[i] + r * pitch[i] (kernel argument, so a plain aggregate).
#define MI_MAX_SLABS 16
struct LeafSlabs {
    const u64 *base[MI_MAX_SLABS];
    uint64_t pitch[MI_MAX_SLABS];
    uint32_t width[MI_MAX_SLABS];
    uint32_t nslabs;
    uint32_t carry_in; // 1: the sponge capacity starts from digests[row] (earlier columns were absorbed by an earlier launch)
    // EMIT form (launch_linear_hash_absorb_emit): the one window is a compact chunk whose rows start on 128-byte lines, a lane owns one
    // of 64 CONSECUTIVE rows, and every word it absorbs is also written to columns [emit_col0, emit_col0 + width) of the tile-major
    // section emit = [nrows / 64][emit_cols][64]
    u64 *emit;
    uint32_t emit_cols, emit_col0;
};

// Two-level power table: g^e = hi[e >> lo_bits] * lo[e & mask]   (lo[j] = s0 * g^j, hi[j] = g^(j << lo_bits))
struct PowTable {
    u64 *lo = nullptr, *hi = nullptr;
    uint32_t lo_bits = 0;
};

struct NttPlan {
    uint32_t log_n = 0;
    PowTable tw;        // w_n^e, e < n
    PowTable inv_scale; // 1/n (constant table, g = 1)
    PowTable lde_scale; // shift^k / n
};

// Grow-only device buffers the entry points size per call.  A context owns one set; the contexts of shards that share a physical device
// (csrc/multi.hip, "device groups": a one-GPU box rehearsing eight shards at full size) point at their group leader's set instead --
// they also share the leader's streams, so the users of a buffer are ordered -- and eight shards do not hold eight 8 GiB operand copies.
struct CtxPool {
    // device scratch of the entry points that need a few hundred MB per call (plookup tables, evaluation partials): kept
    // for the life of the context -- memory that goes back to the driver, also through the stream-ordered pool, is wiped in the
    // background and slows down whatever runs next ("released and fresh device memory")
    char *scratch = nullptr;
    uint64_t scratch_bytes = 0;
    u64 *chelpers_stage = nullptr;   // constraint evaluators: per-workgroup transposed operand staging
    uint64_t chelpers_stage_bytes = 0;
    // native-code constraint evaluators (chelpers_native.hip): constants table, tile-major operand copy of one batch of rows, chunk spill
    u64 *chelpers_cst = nullptr, *chelpers_tiled = nullptr, *chelpers_spill = nullptr, *chelpers_lin = nullptr;
    uint64_t chelpers_cst_bytes = 0, chelpers_tiled_bytes = 0, chelpers_spill_bytes = 0, chelpers_lin_bytes = 0;
};

struct mi_ctx {
    int device = 0;
    // The LOGICAL shard this context works for (csrc/multi.hip sets it: shard g's context; 0 for every other context, the home of a
    // proof's image included).  Several shards may sit on one physical device; with MI_MULTI_CHECK=1 every device pointer an entry point
    // is handed must belong to the context's logical shard (mi_own_check), so that a one-GPU rehearsal catches what only a real multi-GPU
    // node would otherwise show: a buffer, a program or an event of one shard used on behalf of another.
    int logical = 0;
    CtxPool own_pool;
    CtxPool *pool = &own_pool;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int poseidon_variant = 2; // 2 = full rounds on 32-bit halves (v_mad_u64_u32) + grouped optimised partial rounds; 0 / 1 = naive rounds
    uint64_t workspace_limit = 32ULL << 30;
    u64 *workspace = nullptr;
    uint64_t workspace_bytes = 0;
    // mi_ctx_lend_workspace: while a caller's buffer is lent, `workspace` / `workspace_bytes` / `workspace_limit` describe IT and
    // the context's own allocation waits here
    bool workspace_lent = false;
    u64 *own_workspace = nullptr;
    uint64_t own_workspace_bytes = 0, own_workspace_limit = 0;
    u64 *w256 = nullptr; // w_256^j, j < 256
    u64 *small = nullptr; // 4 KiB device scratch for host-pointer single hashes
    u64 *pinned = nullptr; // 4 KiB of page-locked host memory: the few words an entry point hands back (a hash, a flag) are copied
                           // through it -- pageable async copies go through the runtime's staging pool, whose housekeeping after a
                           // few dozen of them stalled a later launch by ~40 ms (seen in an earlier benchmark's FRI phase)
    std::map<uint32_t, NttPlan> plans;
    std::vector<void *> owned; // tables to free
    static constexpr int N_TIMERS = 64;
    hipEvent_t ev_start[N_TIMERS] = {}, ev_stop[N_TIMERS] = {};
    int cu_count = 256;
    uint64_t poseidon_coop_max = 16384; // at most this many independent permutations per launch take the wave-cooperative form (0 = never)
    bool leaf_line_aligned = true; // leaf sponge fetches whole aligned 128-byte lines (k_linear_hash_rows_lines)
    bool lde_fuse_mid = true; // extendPol: last INTT pass and first NTT pass in one kernel (k_lde_mid) when the splits line up
    uint32_t ntt_log_b = 5; // log2 of the NTT tile's batch width (elements per row segment): 4 or 5
    bool poseidon_constants_uploaded = false; // c_rc / c_sparse on this context's device
    u64 *chelpers_scratch = nullptr; // challenges / public inputs / ZhInv of the running constraint-evaluator program
    // mi_lde_merkle_host: upload stream, two staging buffers [n x chunk] and their hand-over events
    hipStream_t copy_stream[2] = {};
    // Transcript::put runs on a stream of its own (mi_transcript_put): its operands come from the host, so it depends on nothing that is
    // queued on `stream`, and a long absorb (the 5 304 words of a zkEVM proof's evaluations: 663 chained permutations, 10 ms on one
    // wave) runs beside whatever the caller queued before it instead of behind it
    hipStream_t tr_stream = nullptr;
    u64 *tr_dev = nullptr;
    uint64_t tr_dev_bytes = 0;
    u64 *stage = nullptr;
    uint64_t stage_bytes = 0;
    static constexpr int N_STAGE = 3; // device staging buffers of the upload: the copy of chunk k + 2 must not wait for the kernels of chunk k
    hipEvent_t ev_uploaded[N_STAGE][2] = {}, ev_consumed[N_STAGE] = {}; // [staging buffer][copy stream], [staging buffer]
    // ... second form of the upload: host threads pack a column chunk into page-locked staging (1-D), which then moves at the full
    // PCIe rate instead of the 2-D copies' 39-53 GB/s (the default); 0 threads = 2-D copies
    int pack_threads = -1; // -1: min(16, hardware threads)
    u64 *pack_stage[3] = {};
    uint64_t pack_stage_bytes = 0; // of each
    hipEvent_t ev_pack_sent[3] = {};
    uint64_t chelpers_batch_rows = 0; // rows per batch (0: sized for about 8 GiB of operand copy)
    uint64_t chelpers_min_words = 0; // benchmarking: LDS words per row to allocate at least (occupancy of a bigger program)
    // Entry points serialise on the context (scratch, plans, workspace and timers are shared state) and make
    // ctx->device current first, so one context may be called from several host threads like the reference's
    // static Poseidon / NTT methods; recursive because host-pointer wrappers call other entry points.
    std::recursive_mutex mu;
};

int mi_ensure_workspace(mi_ctx *ctx, uint64_t bytes);

// ---- MI_MULTI_CHECK=1: which logical shard owns which device address range (capi.hip).  Every allocation the library makes for a shard
// (mi_dev_alloc, mi_vmm_reserve, the pools and buffers of csrc/multi.hip) is entered; an entry point then refuses a pointer that belongs
// to ANOTHER shard than the one its context works for.  Addresses nobody entered (a caller's own allocations: torch tensors in the tests)
// pass and are counted.  Off (the default): one relaxed load per entry point.
bool mi_check_on();
void mi_own_add(const void *p, uint64_t bytes, int shard, const char *what);
void mi_own_del(const void *p);
void mi_own_set_leaders(const uint32_t *lead, uint32_t n); // device groups: a group's shards share buffers (the check compares leaders)
int mi_own_check(int shard, const void *p, const char *what); // MI_OK, or MI_ERR_INVALID with the two shards and the range's name in mi_last_error
#define MI_OWN(c, p)                                                                   \
    do {                                                                               \
        if (mi_check_on()) MI_TRY(mi_own_check((c)->logical, (const void *)(p), __func__)); \
    } while (0)
int mi_scratch(mi_ctx *ctx, uint64_t bytes, void **p); // *p = ctx->scratch, at least `bytes` long; stream-ordered use only
int mi_get_plan(mi_ctx *ctx, uint32_t log_n, NttPlan **plan);
int mi_make_pow_table(mi_ctx *ctx, PowTable *t, uint64_t count_log, u64 s0, u64 g);

static inline uint32_t ilog2_u64(uint64_t n)
{
    uint32_t b = 0;
    while ((1ULL << b) < n) b++;
    return b;
}
static inline bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }

// ---- internal launchers (defined in the .hip files)
int launch_permute(mi_ctx *ctx, u64 *out, const u64 *in, uint64_t count);
int launch_transcript_put(mi_ctx *ctx, u64 *io, const u64 *input, uint64_t n, hipStream_t stream);
int launch_linear_hash_rows(mi_ctx *ctx, u64 *digests, const u64 *src, uint64_t pitch, uint64_t ncols, uint64_t nrows);
int launch_linear_hash_absorb(mi_ctx *ctx, u64 *digests, uint32_t nslabs, const u64 *const *bases, const uint64_t *pitches,
                              const uint64_t *widths, uint64_t nrows, bool first, bool final);
// the same over ONE compact window (pitch a multiple of 16 elements, base on a 128-byte line, nrows a multiple of 64), leaving the absorbed
// columns tile-major in dst = [nrows / 64][dst_cols][64] at columns [col0, col0 + width) as well
int launch_linear_hash_absorb_emit(mi_ctx *ctx, u64 *digests, const u64 *base, uint64_t pitch, uint64_t width, uint64_t nrows, bool first, bool final,
                                   u64 *dst, uint64_t dst_cols, uint64_t col0);
int launch_merkle_levels(mi_ctx *ctx, u64 *nodes, uint64_t nleaves);
int launch_merkle_zero_width(mi_ctx *ctx, u64 *nodes, uint64_t nleaves); // the tree over rows of width 0: one value per level
int launch_group_proofs(mi_ctx *ctx, u64 *proofs, const u64 *nodes, const u64 *src, uint64_t pitch, uint64_t height,
                        uint64_t width, const u64 *idx_dev, uint64_t nq, bool tiled = false);
int launch_ntt(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t n, uint64_t ncols,
               int inverse);
int launch_lde(mi_ctx *ctx, u64 *out, uint64_t out_pitch, const u64 *in, uint64_t in_pitch, uint64_t n_ext, uint64_t n,
               uint64_t ncols);
int launch_fill_pow(mi_ctx *ctx, u64 *out, uint64_t count, u64 s0, u64 g, uint64_t stride_exp);
int launch_fri_fold(mi_ctx *ctx, u64 *out, const u64 *pol, unsigned prev_bits, unsigned cur_bits, unsigned nbits_ext,
                    const u64 x[3], uint64_t g0, uint64_t g_count);
int launch_fri_transpose(mi_ctx *ctx, u64 *aux, const u64 *pol, uint64_t degree, unsigned tbits);
int launch_q_split(mi_ctx *ctx, u64 *qq2, const u64 *qq1, uint64_t n, uint64_t n_ext, unsigned qdeg);
int launch_evmap(mi_ctx *ctx, u64 *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits, const u64 *const *pol_ptr,
                 const uint32_t *pol_dim, const u64 *pol_stride, const uint8_t *prime, const u64 *lev, const u64 *lpev, uint64_t row0, uint64_t nrows,
                 const uint64_t *tile_cols = nullptr);
int launch_batch_inverse3(mi_ctx *ctx, u64 *res, const u64 *src, uint64_t n);
int launch_calculate_h1h2(mi_ctx *ctx, u64 *h1, uint64_t h1_stride, u64 *h2, uint64_t h2_stride, const u64 *f, uint64_t f_stride, const u64 *t,
                          uint64_t t_stride, unsigned dim, uint64_t n);
int launch_calculate_z(mi_ctx *ctx, u64 *z, uint64_t z_stride, const u64 *num, uint64_t num_stride, const u64 *den, uint64_t den_stride,
                       uint64_t n, int *closes);
int launch_calculate_z_batch(mi_ctx *ctx, uint32_t nprod, u64 *const *z, const uint64_t *z_stride, const u64 *const *num, const uint64_t *num_stride,
                             const u64 *const *den, const uint64_t *den_stride, uint64_t n, int *closes);
int launch_geom_seq(mi_ctx *ctx, u64 *out, uint64_t n, u64 start, u64 ratio);
int launch_geom_seq3(mi_ctx *ctx, u64 *out, uint64_t n, const u64 ratio[3]);
int launch_x_div_x_sub(mi_ctx *ctx, u64 *out, const u64 *x, uint64_t n, const u64 xi[3]);
int launch_zhinv(mi_ctx *ctx, u64 *out, uint64_t cnt, u64 sn, u64 w);
int launch_fill_synthetic(mi_ctx *ctx, u64 *out, uint64_t count, u64 seed);
int launch_fill_synthetic_2d(mi_ctx *ctx, u64 *out, uint64_t out_pitch, uint64_t nrows, uint64_t ncols, uint64_t global_cols,
                             uint64_t col0, u64 seed);
int launch_copy_2d(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t src_pitch, uint64_t nrows, uint64_t ncols);
// row-major [nrows x ncols] at src_pitch -> columns [col0, col0 + ncols) of the tile-major section dst = [nrows / 64][ncols_total][64], canonical
int launch_tile_major(mi_ctx *ctx, u64 *dst, uint64_t ncols_total, uint64_t col0, const u64 *src, uint64_t src_pitch, uint64_t nrows, uint64_t ncols);
int launch_untile(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t ncols_total, uint64_t col0, uint64_t row0, uint64_t nrows, uint64_t ncols);
