// lookup.hip -- what Starks::genProof computes between the base-domain constraint steps, on polynomials that already live in HBM:
//
//   plookup h1 / h2     starks.cpp:92-128  -> Polinomial::calculateH1H2_opt1 / _opt3 (polinomial.hpp:349-584; :303-347 is the plain form)
//   grand product z     starks.cpp:174-187 -> Polinomial::calculateZ (polinomial.hpp:586-607)
//
// The reference transposes the columns into contiguous buffers first (transposeH1H2Columns / transposeZColumns) because its
// loops are sequential per polynomial; here a polynomial is read through its strided view (element i at p[i * stride .. + dim))
// by one thread per row, so the transposes disappear.
//
// h1 / h2: every row of t counts once plus once per row of f holding the same value, rows of f being credited to the LAST row of t
// with that value; walking t in order and repeating each row by its count gives 2n values, alternately h1[i] and h2[i].
//   1. an insert-only open-addressing table (linear probing, 2n..4n slots of 32-bit row numbers, compare-and-swap on the slot,
//      the key compared by reading t itself) maps a value to the last row of t that holds it (atomicMax on the slot);
//   2. every row of f probes the table and adds one to its row's counter; a value that is absent aborts the call with the row;
//   3. an exclusive prefix sum of the counters gives the first output position of every row of t;
//   4. one thread per output pair finds its row of t by bisection on the prefix sums.
// Runs of equal values are the normal case (padding rows, selector-gated lookups), i.e. thousands of atomics on one address:
// each wave first merges the lanes that target the same slot / counter (a few rounds of readfirstlane + ballot).
//
// z: z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1] in F_p^3 -- an exclusive scan under the extension field's product.  Field
// arithmetic is exact, so the scan's association order cannot change a bit: block products, a one-block scan of those, then each
// block rescans its rows (the quotients are recomputed instead of stored: two Fermat inversions per row are ~1 ms at 2^23 rows).
#include "common.h"

using gl::E3;

namespace {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;

template <int DIM> struct Key { u64 k[DIM]; };

template <int DIM> __device__ __forceinline__ Key<DIM> load_key(const u64 *p)
{
    Key<DIM> r;
#pragma unroll
    for (int j = 0; j < DIM; j++) r.k[j] = gl::canon_sel(p[j]); // keys are field VALUES: x and x + p are the same key (polinomial.hpp CompareFe)
    return r;
}
template <int DIM> __device__ __forceinline__ bool same(const Key<DIM> &a, const Key<DIM> &b)
{
    bool e = true;
#pragma unroll
    for (int j = 0; j < DIM; j++) e = e && a.k[j] == b.k[j];
    return e;
}
template <int DIM> __device__ __forceinline__ uint64_t hash_key(const Key<DIM> &a)
{
    uint64_t h = a.k[0] * 0x9E3779B97F4A7C15ull;
    if (DIM > 1) h ^= a.k[1] * 0xC2B2AE3D27D4EB4Full + (h >> 31);
    if (DIM > 2) h ^= a.k[2] * 0x165667B19E3779F9ull + (h >> 29);
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}

// Lanes of the wave that carry the same `target` merge their contributions: up to ROUNDS groups are peeled off (the group of the
// first remaining lane each time, one lane acting for it); what is left acts lane by lane.  `pending` = this lane still has work.
// F(target, merged_value): merged_value = the largest `val` of the group (MAX) or the number of lanes in it (!MAX).
template <bool MAX, typename F> __device__ __forceinline__ void wave_merge(bool pending, uint32_t target, uint32_t val, F act)
{
    constexpr int ROUNDS = 4;
    for (int r = 0; r < ROUNDS; r++) {
        const uint64_t live = __ballot(pending);
        if (!live) return;
        const int first = __ffsll((long long)live) - 1;
        const uint32_t lead = (uint32_t)__shfl((int)target, first);
        const bool mine = pending && target == lead;
        const uint64_t grp = __ballot(mine);
        uint32_t merged;
        if (MAX) { // rows are dealt to lanes in increasing order, so the last lane of the group holds the largest row
            const int last = 63 - __clzll((long long)grp);
            merged = (uint32_t)__shfl((int)val, last);
        } else {
            merged = (uint32_t)__popcll(grp);
        }
        if (mine && (int)(threadIdx.x & 63) == first) act(lead, merged);
        pending = pending && !mine;
    }
    if (pending) act(target, MAX ? val : 1u);
}

template <int DIM>
__global__ __launch_bounds__(256) void k_h1h2_insert(uint32_t *__restrict__ table, uint32_t mask, const u64 *__restrict__ t, uint64_t t_stride,
                                                     uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    Key<DIM> key = load_key<DIM>(t + (uint64_t)(active ? i : 0) * t_stride);
    uint32_t h = (uint32_t)hash_key<DIM>(key) & mask;
    bool found = false; // the slot of this value is known (it holds a row with the same value) and only needs max(row)
    bool done = !active;
    while (!done) {
        uint32_t cur = table[h];
        if (cur == EMPTY) {
            cur = atomicCAS(&table[h], EMPTY, i);
            if (cur == EMPTY) { done = true; break; }
        }
        if (same<DIM>(load_key<DIM>(t + (uint64_t)cur * t_stride), key)) { found = true; done = true; break; }
        h = (h + 1) & mask;
    }
    // (the loop is divergent; the merge below runs with the whole wave converged again)
    // The plain read only saves atomics: the vector L1 is not coherent within a launch, so it may return an older value -- a smaller
    // row (the atomic then runs, harmlessly) or still EMPTY although the compare-and-swap above has seen the slot taken (EMPTY is the
    // largest 32-bit value: it must not be mistaken for "already larger").
    wave_merge<true>(found, h, i, [&](uint32_t slot, uint32_t row) {
        const uint32_t seen = table[slot];
        if (seen == EMPTY || seen < row) atomicMax(&table[slot], row);
    });
}

template <int DIM>
__global__ __launch_bounds__(256) void k_h1h2_count(uint32_t *__restrict__ counter, const uint32_t *__restrict__ table, uint32_t mask,
                                                    const u64 *__restrict__ f, uint64_t f_stride, const u64 *__restrict__ t, uint64_t t_stride,
                                                    uint32_t n, unsigned long long *__restrict__ bad_row)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    Key<DIM> key = load_key<DIM>(f + (uint64_t)(active ? i : 0) * f_stride);
    uint32_t h = (uint32_t)hash_key<DIM>(key) & mask;
    uint32_t row = EMPTY;
    bool done = !active;
    while (!done) {
        const uint32_t cur = table[h];
        if (cur == EMPTY) { atomicMin(bad_row, (unsigned long long)i); done = true; break; } // polinomial.hpp:321-325: "Number not included"
        if (same<DIM>(load_key<DIM>(t + (uint64_t)cur * t_stride), key)) { row = cur; done = true; break; }
        h = (h + 1) & mask;
    }
    wave_merge<false>(row != EMPTY, row, 0u, [&](uint32_t r, uint32_t cnt) { atomicAdd(&counter[r], cnt); });
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = v;
}

// ---- exclusive prefix sum of n 32-bit counters (their total, 2n, fits): SCAN_PER_BLOCK consecutive items per block
constexpr uint32_t SCAN_ITEMS = 16, SCAN_PER_BLOCK = 256 * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan_u32(uint32_t v, uint32_t *lds, uint32_t &total)
{
    const uint32_t tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t x = tid >= d ? lds[tid - d] : 0;
        __syncthreads();
        lds[tid] += x;
        __syncthreads();
    }
    total = lds[255];
    const uint32_t incl = lds[tid];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(uint32_t *__restrict__ sums, const uint32_t *__restrict__ in, uint32_t n)
{
    __shared__ uint32_t lds[256];
    const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) s += base + j < n ? in[base + j] : 0;
    uint32_t total;
    block_exclusive_scan_u32(s, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one block: sums[0..nb) -> their exclusive prefix sums, in place
__global__ __launch_bounds__(256) void k_scan_top(uint32_t *sums, uint32_t nb)
{
    __shared__ uint32_t lds[256];
    const uint32_t per = (nb + 255) / 256, lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t s = 0;
    for (uint32_t j = lo; j < hi; j++) s += sums[j];
    uint32_t total;
    uint32_t run = block_exclusive_scan_u32(s, lds, total);
    for (uint32_t j = lo; j < hi; j++) { const uint32_t v = sums[j]; sums[j] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_scan_apply(uint32_t *__restrict__ out, const uint32_t *__restrict__ in, const uint32_t *__restrict__ sums,
                                                    uint32_t n)
{
    __shared__ uint32_t lds[256];
    const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    uint32_t total;
    uint32_t run = block_exclusive_scan_u32(s, lds, total) + sums[blockIdx.x];
    for (uint32_t j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

// output positions 2i, 2i+1 -> rows of t (polinomial.hpp:330-346); start[] strictly increasing, start[0] = 0
template <int DIM>
__global__ __launch_bounds__(256) void k_h1h2_expand(u64 *__restrict__ h1, uint64_t h1_stride, u64 *__restrict__ h2, uint64_t h2_stride,
                                                     const u64 *__restrict__ t, uint64_t t_stride, const uint32_t *__restrict__ start, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = 2 * i;
    uint32_t lo = 0, hi = p < n - 1 ? p : n - 1; // start[id] >= id, so the row is at most p
    while (lo < hi) {                           // the largest id with start[id] <= p
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (start[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const uint32_t id1 = lo, id2 = (id1 + 1 < n && start[id1 + 1] <= p + 1) ? id1 + 1 : id1;
    const Key<DIM> a = load_key<DIM>(t + (uint64_t)id1 * t_stride), b = load_key<DIM>(t + (uint64_t)id2 * t_stride);
#pragma unroll
    for (int j = 0; j < DIM; j++) {
        h1[(uint64_t)i * h1_stride + j] = a.k[j];
        h2[(uint64_t)i * h2_stride + j] = b.k[j];
    }
}

// ---- grand product
constexpr uint32_t Z_ROWS = 4, Z_PER_BLOCK = 256 * Z_ROWS; // (measured at 30 products x 2^23 rows: 8 rows a thread 24.6 ms, 4 rows 19.1, 2 rows 25.0: registers against inversions)

__device__ __forceinline__ E3 load3(const u64 *p) { return E3{{p[0], p[1], p[2]}}; }
__device__ __forceinline__ E3 one3() { return E3{{1, 0, 0}}; }

// inclusive scan of one E3 per thread under the field product; returns the exclusive value, total = product of the block
__device__ __forceinline__ E3 block_exclusive_scan_e3(const E3 &v, u64 *lds, E3 &total)
{
    const uint32_t tid = threadIdx.x;
    E3 cur = v;
    lds[tid] = cur.v[0]; lds[256 + tid] = cur.v[1]; lds[512 + tid] = cur.v[2];
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        E3 x = one3();
        const bool has = tid >= d;
        if (has) x = E3{{lds[tid - d], lds[256 + tid - d], lds[512 + tid - d]}};
        __syncthreads();
        if (has) {
            cur = gl::e3_mul(x, cur);
            lds[tid] = cur.v[0]; lds[256 + tid] = cur.v[1]; lds[512 + tid] = cur.v[2];
        }
        __syncthreads();
    }
    total = E3{{lds[255], lds[511], lds[767]}};
    E3 excl = one3();
    if (tid) excl = E3{{lds[tid - 1], lds[256 + tid - 1], lds[512 + tid - 1]}};
    __syncthreads();
    return excl;
}

// FINAL = false: z = the rows' quotients num / den, prods[block] = their product over the block.  FINAL = true: z (holding the quotients)
// becomes the running product over the block's rows, from prods[block] = the product of everything before the block.  z may not
// overlap num or den.
template <bool FINAL>
__device__ __forceinline__ void z_block(u64 *__restrict__ z, uint64_t z_stride, const u64 *__restrict__ num, uint64_t num_stride,
                                        const u64 *__restrict__ den, uint64_t den_stride, uint64_t n, u64 *__restrict__ prods, uint32_t blk)
{
    __shared__ u64 lds[768];
    const uint64_t base = (uint64_t)blk * Z_PER_BLOCK + (uint64_t)threadIdx.x * Z_ROWS;
    E3 q[Z_ROWS], local = one3();
    if (!FINAL) {
        // The thread's Z_ROWS denominators share ONE inversion (Montgomery's trick: invert the running product, peel the factors off
        // backwards): 1 inversion (an adjugate + a 96-multiplication Fermat power) + 3 (Z_ROWS - 1) products instead of Z_ROWS
        // inversions.  Exact arithmetic, so the quotients are the same field elements; a zero denominator (inverse defined as 0, like
        // the reference's Goldilocks3::inv) is taken out of the chain and its quotient set to 0.
        // The quotients are parked in z (which the second pass overwrites with the running product), so the second pass reads 3 words
        // a row instead of 6 and inverts nothing.
        E3 pre[Z_ROWS], acc = one3();
        bool dead[Z_ROWS];
#pragma unroll
        for (uint32_t j = 0; j < Z_ROWS; j++) {
            E3 d = one3();
            if (base + j < n) d = load3(den + (base + j) * den_stride);
            dead[j] = (d.v[0] | d.v[1] | d.v[2]) == 0;
            if (dead[j]) d = one3();
            pre[j] = acc;
            acc = gl::e3_mul(acc, d);
            q[j] = d; // the denominator for now
        }
        E3 inv = gl::e3_inv(acc);
#pragma unroll
        for (int j = Z_ROWS - 1; j >= 0; j--) {
            const E3 dinv = gl::e3_mul(inv, pre[j]);
            inv = gl::e3_mul(inv, q[j]);
            q[j] = one3();
            if (base + j < n) {
                q[j] = dead[j] ? E3{{0, 0, 0}} : gl::e3_mul(load3(num + (base + j) * num_stride), dinv);
                u64 *o = z + (base + j) * z_stride;
                o[0] = q[j].v[0]; o[1] = q[j].v[1]; o[2] = q[j].v[2];
            }
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < Z_ROWS; j++) {
            q[j] = one3();
            if (base + j < n) q[j] = load3(z + (base + j) * z_stride);
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < Z_ROWS; j++) local = gl::e3_mul(local, q[j]);
    E3 total;
    E3 run = block_exclusive_scan_e3(local, lds, total);
    if (!FINAL) {
        if (threadIdx.x == 0) { prods[(uint64_t)blk * 3] = total.v[0]; prods[(uint64_t)blk * 3 + 1] = total.v[1]; prods[(uint64_t)blk * 3 + 2] = total.v[2]; }
        return;
    }
    run = gl::e3_mul(load3(prods + (uint64_t)blk * 3), run);
#pragma unroll
    for (uint32_t j = 0; j < Z_ROWS; j++) {
        if (base + j < n) {
            u64 *o = z + (base + j) * z_stride;
            o[0] = run.v[0]; o[1] = run.v[1]; o[2] = run.v[2];
        }
        run = gl::e3_mul(run, q[j]);
    }
}
template <bool FINAL>
__global__ __launch_bounds__(256) void k_z_blocks(u64 *__restrict__ z, uint64_t z_stride, const u64 *__restrict__ num, uint64_t num_stride,
                                                  const u64 *__restrict__ den, uint64_t den_stride, uint64_t n, u64 *__restrict__ prods)
{
    z_block<FINAL>(z, z_stride, num, num_stride, den, den_stride, n, prods, blockIdx.x);
}

// Several grand products over the same rows in one launch (a stage's products read their numerators and denominators from columns of
// the SAME rows -- 24 bytes of each polynomial out of a row of hundreds of words -- so a cache line a product fetches holds its
// neighbours' operands too).  Workgroups are numbered product-fastest within a block of rows, and so that every product of one block
// of rows lands on the same XCD (workgroup id mod 8): the lines the first product pulls into that XCD's L2 serve the others.
constexpr uint32_t Z_BATCH = 32;
struct ZBatch {
    u64 *z[Z_BATCH];
    const u64 *num[Z_BATCH], *den[Z_BATCH];
    uint32_t z_stride[Z_BATCH], num_stride[Z_BATCH], den_stride[Z_BATCH];
};
template <bool FINAL>
__global__ __launch_bounds__(256) void k_z_blocks_batch(const ZBatch b, uint32_t nprod, uint32_t nb, uint64_t n, u64 *__restrict__ prods)
{
    const uint32_t per = 8 * nprod, g = blockIdx.x / per, r = blockIdx.x % per, prod = r >> 3, blk = g * 8 + (r & 7);
    if (blk >= nb) return; // (uniform over the workgroup)
    z_block<FINAL>(b.z[prod], b.z_stride[prod], b.num[prod], b.num_stride[prod], b.den[prod], b.den_stride[prod], n,
                   prods + (uint64_t)prod * ((uint64_t)nb + 1) * 3, blk);
}
// one block per product: prods[0..nb) -> exclusive products in place; prods[nb] = the product of all
__global__ __launch_bounds__(256) void k_z_scan_top(u64 *prods_all, uint32_t nb)
{
    __shared__ u64 lds[768];
    u64 *prods = prods_all + (uint64_t)blockIdx.x * ((uint64_t)nb + 1) * 3, *total_out = prods + (uint64_t)nb * 3;
    const uint32_t per = (nb + 255) / 256, lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    E3 s = one3();
    for (uint32_t j = lo; j < hi; j++) s = gl::e3_mul(s, load3(prods + (uint64_t)j * 3));
    E3 total;
    E3 run = block_exclusive_scan_e3(s, lds, total);
    for (uint32_t j = lo; j < hi; j++) {
        const E3 v = load3(prods + (uint64_t)j * 3);
        prods[(uint64_t)j * 3] = run.v[0]; prods[(uint64_t)j * 3 + 1] = run.v[1]; prods[(uint64_t)j * 3 + 2] = run.v[2];
        run = gl::e3_mul(run, v);
    }
    if (threadIdx.x == 0) { total_out[0] = total.v[0]; total_out[1] = total.v[1]; total_out[2] = total.v[2]; }
}

template <int DIM>
int h1h2_impl(mi_ctx *ctx, u64 *h1, uint64_t h1_stride, u64 *h2, uint64_t h2_stride, const u64 *f, uint64_t f_stride, const u64 *t,
              uint64_t t_stride, uint32_t n)
{
    uint32_t slots = 2;
    while (slots < 2 * (uint64_t)n) slots <<= 1; // load factor in (1/4, 1/2]
    const uint32_t nb_scan = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK, nb = (n + 255) / 256;
    const uint64_t words = (uint64_t)slots + 2 * (uint64_t)n + nb_scan + 4;
    uint32_t *scratch = nullptr;
    MI_TRY(mi_scratch(ctx, words * 4, (void **)&scratch));
    uint32_t *table = scratch, *counter = table + slots, *start = counter + n, *sums = start + n;
    unsigned long long *bad = (unsigned long long *)(scratch + ((words - 2) & ~1ull));
    hipLaunchKernelGGL(k_fill_u32, dim3(mi_grid_256(slots)), dim3(256), 0, ctx->stream, table, (uint64_t)slots, EMPTY);
    hipLaunchKernelGGL(k_fill_u32, dim3(mi_grid_256(n)), dim3(256), 0, ctx->stream, counter, (uint64_t)n, 1u); // polinomial.hpp:351
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, ctx->stream, (uint32_t *)bad, (uint64_t)2, EMPTY);
    hipLaunchKernelGGL(k_h1h2_insert<DIM>, dim3(nb), dim3(256), 0, ctx->stream, table, slots - 1, t, t_stride, n);
    hipLaunchKernelGGL(k_h1h2_count<DIM>, dim3(nb), dim3(256), 0, ctx->stream, counter, table, slots - 1, f, f_stride, t, t_stride, n, bad);
    hipError_t e = hipMemcpyAsync(ctx->pinned, bad, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    const unsigned long long bad_host = ctx->pinned[0];
    if (e != hipSuccess || bad_host != ~0ull) {
        if (e != hipSuccess) { mi_set_error("calculateH1H2 failed: %s", hipGetErrorString(e)); return MI_ERR_HIP; }
        mi_set_error("calculateH1H2: number not included: w=%llu", bad_host);
        return MI_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb_scan), dim3(256), 0, ctx->stream, sums, counter, n);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, ctx->stream, sums, nb_scan);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb_scan), dim3(256), 0, ctx->stream, start, counter, sums, n);
    hipLaunchKernelGGL(k_h1h2_expand<DIM>, dim3(nb), dim3(256), 0, ctx->stream, h1, h1_stride, h2, h2_stride, t, t_stride, start, n);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

} // namespace

int launch_calculate_h1h2(mi_ctx *ctx, u64 *h1, uint64_t h1_stride, u64 *h2, uint64_t h2_stride, const u64 *f, uint64_t f_stride, const u64 *t,
                          uint64_t t_stride, unsigned dim, uint64_t n)
{
    if (!n) return MI_OK;
    MI_REQUIRE(dim == 1 || dim == 3, "polynomial dim must be 1 or 3");
    MI_REQUIRE(n < (1ull << 30), "too many rows");
    MI_REQUIRE(h1_stride >= dim && h2_stride >= dim && f_stride >= dim && t_stride >= dim, "stride smaller than dim");
    if (dim == 1) return h1h2_impl<1>(ctx, h1, h1_stride, h2, h2_stride, f, f_stride, t, t_stride, (uint32_t)n);
    return h1h2_impl<3>(ctx, h1, h1_stride, h2, h2_stride, f, f_stride, t, t_stride, (uint32_t)n);
}

int launch_calculate_z(mi_ctx *ctx, u64 *z, uint64_t z_stride, const u64 *num, uint64_t num_stride, const u64 *den, uint64_t den_stride,
                       uint64_t n, int *closes)
{
    if (!n) { if (closes) *closes = 1; return MI_OK; }
    MI_REQUIRE(n < (1ull << 32), "too many rows");
    MI_REQUIRE(z_stride >= 3 && num_stride >= 3 && den_stride >= 3, "stride smaller than dim");
    const uint32_t nb = (uint32_t)((n + Z_PER_BLOCK - 1) / Z_PER_BLOCK);
    u64 *prods = nullptr;
    MI_TRY(mi_scratch(ctx, ((uint64_t)nb + 1) * 24, (void **)&prods));
    u64 *total = prods + (uint64_t)nb * 3;
    hipLaunchKernelGGL(k_z_blocks<false>, dim3(nb), dim3(256), 0, ctx->stream, z, z_stride, num, num_stride, den, den_stride, n, prods);
    hipLaunchKernelGGL(k_z_scan_top, dim3(1), dim3(256), 0, ctx->stream, prods, nb);
    hipLaunchKernelGGL(k_z_blocks<true>, dim3(nb), dim3(256), 0, ctx->stream, z, z_stride, num, num_stride, den, den_stride, n, prods);
    MI_HIP_CHECK(hipGetLastError());
    if (closes) { // polinomial.hpp:603-606 (a zkassert there): does the product return to one?
        const u64 *tot = ctx->pinned;
        MI_HIP_CHECK(hipMemcpyAsync(ctx->pinned, total, 24, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *closes = tot[0] == 1 && tot[1] == 0 && tot[2] == 0;
    }
    return MI_OK;
}

int launch_calculate_z_batch(mi_ctx *ctx, uint32_t nprod, u64 *const *z, const uint64_t *z_stride, const u64 *const *num, const uint64_t *num_stride,
                             const u64 *const *den, const uint64_t *den_stride, uint64_t n, int *closes)
{
    if (!n || !nprod) {
        for (uint32_t i = 0; closes && i < nprod; i++) closes[i] = 1;
        return MI_OK;
    }
    MI_REQUIRE(n < (1ull << 32), "too many rows");
    const uint32_t nb = (uint32_t)((n + Z_PER_BLOCK - 1) / Z_PER_BLOCK), nb8 = (nb + 7) / 8 * 8;
    for (uint32_t i = 0; i < nprod; i++) {
        MI_REQUIRE(z[i] && num[i] && den[i], "null buffer");
        MI_REQUIRE(z_stride[i] >= 3 && num_stride[i] >= 3 && den_stride[i] >= 3, "stride smaller than dim");
        MI_REQUIRE(z_stride[i] < (1ull << 32) && num_stride[i] < (1ull << 32) && den_stride[i] < (1ull << 32), "stride too large");
    }
    const uint32_t chunk_max = nprod < Z_BATCH ? nprod : Z_BATCH;
    MI_REQUIRE((uint64_t)nb8 * chunk_max < (1ull << 31), "too many workgroups");
    u64 *prods = nullptr;
    MI_TRY(mi_scratch(ctx, (uint64_t)chunk_max * ((uint64_t)nb + 1) * 24, (void **)&prods));
    for (uint32_t first = 0; first < nprod; first += Z_BATCH) {
        const uint32_t cnt = nprod - first < Z_BATCH ? nprod - first : Z_BATCH;
        ZBatch b;
        for (uint32_t i = 0; i < Z_BATCH; i++) {
            const uint32_t k = first + (i < cnt ? i : 0);
            b.z[i] = z[k]; b.num[i] = num[k]; b.den[i] = den[k];
            b.z_stride[i] = (uint32_t)z_stride[k]; b.num_stride[i] = (uint32_t)num_stride[k]; b.den_stride[i] = (uint32_t)den_stride[k];
        }
        hipLaunchKernelGGL(k_z_blocks_batch<false>, dim3(nb8 * cnt), dim3(256), 0, ctx->stream, b, cnt, nb, n, prods);
        hipLaunchKernelGGL(k_z_scan_top, dim3(cnt), dim3(256), 0, ctx->stream, prods, nb);
        hipLaunchKernelGGL(k_z_blocks_batch<true>, dim3(nb8 * cnt), dim3(256), 0, ctx->stream, b, cnt, nb, n, prods);
        MI_HIP_CHECK(hipGetLastError());
        if (closes) { // the totals of this chunk in one copy (3 words each, (nb + 1) * 3 words apart)
            MI_HIP_CHECK(hipMemcpy2DAsync(ctx->pinned, 24, prods + (uint64_t)nb * 3, ((uint64_t)nb + 1) * 24, 24, cnt, hipMemcpyDeviceToHost, ctx->stream));
            MI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (uint32_t i = 0; i < cnt; i++) {
                const u64 *tot = ctx->pinned + 3 * i;
                closes[first + i] = tot[0] == 1 && tot[1] == 0 && tot[2] == 0;
            }
        } // (the next chunk reuses prods: same stream, so it runs after this one)
    }
    return MI_OK;
}
