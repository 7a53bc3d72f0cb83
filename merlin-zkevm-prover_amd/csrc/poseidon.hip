// poseidon.hip -- Poseidon sponge / Merkle-tree kernels for gfx950.
//
// Replaces PoseidonGoldilocks::{hash_full_result, hash, linear_hash, merkletree_avx} as called from
// merkleTreeGL.cpp:37-44 and transcript.cpp:23,46, and MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35).
//
// Mapping: ONE ROW (one sponge) PER LANE.  A leaf of the 665-column trace is 84 chained permutations, so
// the kernel is VALU-bound by three orders of magnitude (~1.7e4 integer instructions per 64 B absorbed).
// Default leaf kernel (k_linear_hash_rows_lines): every lane pulls whole aligned 128-byte lines of its row,
// one permutation ahead of their use, straight into a per-lane ring in LDS (global_load_lds_dwordx4) so that
// each line crosses the fabric once; k_linear_hash_rows (plain 8-byte loads of the next blocks) is kept for
// rows of at most 4 columns and for A/B runs (mi_set_leaf_mode 0).  Tree levels read two sibling digests
// (64 B contiguous per lane, fully coalesced) and write 32 B.
#include "common.h"
#include "poseidon_math.h"

__constant__ u64 c_rc[360];
__constant__ pos::SparseTables c_sparse;

// __constant__ symbols exist once per device; every context uploads them to ITS device the first time it hashes
// (entry points make ctx->device current before they get here).
static int upload_rc_once(mi_ctx *ctx)
{
    if (ctx->poseidon_constants_uploaded) return MI_OK;
    MI_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_rc), MI_POS_RC, sizeof(MI_POS_RC)));
    pos::SparseTables *host_tables = new pos::SparseTables();
    pos::fill_sparse_tables(*host_tables);
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_sparse), host_tables, sizeof(*host_tables));
    delete host_tables;
    MI_HIP_CHECK(e);
    ctx->poseidon_constants_uploaded = true;
    return MI_OK;
}

template <int MDS>
__global__ __launch_bounds__(256) void k_permute(u64 *__restrict__ out, const u64 *__restrict__ in, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = in[i * 12 + j];
    pos::permute<MDS>(s, c_rc, &c_sparse);
#pragma unroll
    for (int j = 0; j < 12; j++) out[i * 12 + j] = s[j];
}

// ---- wave-cooperative permutation: ONE STATE ACROSS 12 LANES (north_star's "warp-cooperative Poseidon round").
// A whole state per lane is the right mapping when there are millions of sponges (the leaves): every VALU slot does useful work.  It
// is the wrong one for the few DEPENDENT hashes a proof also needs -- a transcript permutation, the top of a Merkle tree, the levels of
// the small FRI trees: there a lane walks through all ~17 000 instructions alone (58-66 us measured) while the chip idles.  Here
// element j of a state lives in lane 16 q + j (j < 12; four states per wave, lanes 12-15 of a row compute on zeros): the S-box is one
// element per lane (in the 22 partial rounds only lane 0's result is kept -- the other lanes would idle either way), the MDS is
// circulant, out_j = sum_k MC[k] s[(j + k) mod 12] (+ 8 s_0 for j = 0), so the coefficient of the k-th rotation is a compile-time
// constant and the rotation is a ds_bpermute of the two 32-bit halves: 22 cross-lane reads, 24 multiply-adds and one closing per round.
// ~3 600 instructions per wave and permutation instead of ~17 000 per lane: a dependent hash in ~10 us.  Naive round structure
// (poseidon_g_executor.cpp:174-205), bit-identical with the per-lane variants (same tests).
__device__ __forceinline__ u64 permute_coop(u64 x, uint32_t j /* lane & 15 */)
{
    const uint32_t lane = __lane_id(), row = lane & ~15u;
    const bool live = j < 12;
    const uint32_t jj = live ? j : 0;
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        x = gl::add_wc(x, live ? c_rc[r * 12 + jj] : 0);
        const u64 sb = pos::sbox(x);
        x = (r < 4 || r >= 26 || j == 0) ? sb : x;
        // MDS on 32-bit halves, rotations through the LDS crossbar
        const u32 lo = (u32)x, hi = (u32)(x >> 32);
        u64 al = (u64)lo * (u32)(pos::MC[0] + (j == 0 ? pos::MD0 : 0)), ah = (u64)hi * (u32)(pos::MC[0] + (j == 0 ? pos::MD0 : 0));
        pos::nttm_static_for<1, 12>([&](auto K) {
            constexpr int k = decltype(K)::value;
            uint32_t src = jj + k;
            src = (src >= 12 ? src - 12 : src) + row;
            const u32 l = (u32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)lo);
            const u32 h = (u32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)hi);
            al += (u64)l * (u32)pos::MC[k];
            ah += (u64)h * (u32)pos::MC[k];
        });
        // value = al + ah * 2^32, 2^64 = 2^32 - 1 (as mds_half32)
        const u32 ahl = (u32)ah, ahh = (u32)(ah >> 32);
        const u64 t = (u64)ahh * 0xFFFFFFFFu + al;
        const u32 thi = (u32)(t >> 32) + ahl;
        x = (((u64)thi << 32) | (u32)t) + (thi < ahl ? GL_EPS : 0);
    }
    return gl::canon_sel(x);
}

// count states (12 words each), one per 16 lanes
__global__ __launch_bounds__(256) void k_permute_coop(u64 *__restrict__ out, const u64 *__restrict__ in, uint64_t count)
{
    const uint64_t q = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const uint32_t j = threadIdx.x & 15;
    const bool mine = q < count && j < 12;
    const u64 x = permute_coop(mine ? in[q * 12 + j] : 0, j);
    if (mine) out[q * 12 + j] = x;
}

// Transcript::put (transcript.cpp:12-29) of n elements as ONE launch: the sponge is sequential (a permutation per 8 elements, each fed
// the previous one's first four outputs), so the whole absorb runs in the first 16 lanes of one wave with the cooperative permutation.
// io = state[4] | pending[8] | out[12] | pending_cursor | out_cursor (the class's members, in and out).
__global__ __launch_bounds__(64) void k_transcript_put(u64 *io, const u64 *__restrict__ input, uint64_t n)
{
    __shared__ u64 st[4], pend[8], outp[12];
    const uint32_t j = threadIdx.x & 15;
    const bool row0 = threadIdx.x < 16;
    if (threadIdx.x < 4) st[threadIdx.x] = io[threadIdx.x];
    if (threadIdx.x < 8) pend[threadIdx.x] = io[4 + threadIdx.x];
    if (threadIdx.x < 12) outp[threadIdx.x] = io[12 + threadIdx.x];
    uint32_t pc = (uint32_t)io[24], oc = (uint32_t)io[25];
    __syncthreads();
    // up to eight elements at a time, lane j < take holding element i + j, and the next block's elements requested before this block's
    // permutation (element by element the absorb paid one dependent global load each: 10 ms for the 5 304 words of a zkEVM proof's evals)
    uint64_t i = 0;
    uint32_t take = (uint32_t)(n < 8 - pc ? n : 8 - pc);
    u64 nxt = threadIdx.x < take ? input[threadIdx.x] : 0;
    while (i < n) { // uniform over the wave
        if (threadIdx.x < take) pend[pc + threadIdx.x] = nxt;
        pc += take;
        i += take;
        oc = 0;
        take = pc == 8 ? (uint32_t)(n - i < 8 ? n - i : 8) : 0; // (pc < 8: the input is used up)
        if (threadIdx.x < take) nxt = input[i + threadIdx.x];
        __syncthreads();
        if (pc == 8) {
            const u64 x = permute_coop(row0 ? (j < 8 ? pend[j] : j < 12 ? st[j - 8] : 0) : 0, j);
            __syncthreads();
            if (row0 && j < 12) outp[j] = x;
            if (row0 && j < 4) st[j] = x;
            if (row0 && j < 8) pend[j] = 0;
            pc = 0;
            oc = 12;
            __syncthreads();
        }
    }
    if (threadIdx.x < 4) io[threadIdx.x] = st[threadIdx.x];
    if (threadIdx.x < 8) io[4 + threadIdx.x] = pend[threadIdx.x];
    if (threadIdx.x < 12) io[12 + threadIdx.x] = outp[threadIdx.x];
    if (threadIdx.x == 0) { io[24] = pc; io[25] = oc; }
}

int launch_transcript_put(mi_ctx *ctx, u64 *io, const u64 *input, uint64_t n, hipStream_t stream)
{
    MI_TRY(upload_rc_once(ctx));
    hipLaunchKernelGGL(k_transcript_put, dim3(1), dim3(64), 0, stream, io, input, n);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// The tree over rows of width 0 (a STARK without stage-2 polynomials commits to it all the same: root2 of the recursive STARKs'
// golden proofs, starks.cpp:133-137): every leaf is the zero digest, so every level is one value repeated -- log2(n) dependent
// hashes in one wave, then a fill, instead of n - 1 hashes.
__global__ __launch_bounds__(64) void k_uniform_tree_values(u64 *vals, uint32_t nlevels) // vals[l * 4 ..]: the node value of level l; level 0 = 0
{
    __shared__ u64 cur[4];
    const uint32_t j = threadIdx.x & 15;
    const bool row0 = threadIdx.x < 16;
    if (threadIdx.x < 4) { cur[threadIdx.x] = 0; vals[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t l = 1; l < nlevels; l++) { // uniform over the wave
        const u64 x = permute_coop(row0 && j < 8 ? cur[j & 3] : 0, j);
        __syncthreads();
        if (row0 && j < 4) { cur[j] = x; vals[l * 4 + j] = x; }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_uniform_tree_fill(u64 *__restrict__ nodes, const u64 *__restrict__ vals, uint64_t nleaves)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; // node index, level-major: level l at [2n - 2(n >> l) ... )
    if (g >= 2 * nleaves - 1) return;
    uint32_t l = 0;
    while (g >= 2 * nleaves - 2 * (nleaves >> (l + 1))) l++;
    u64 *o = nodes + g * 4;
    const u64 *v = vals + l * 4;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
int launch_merkle_zero_width(mi_ctx *ctx, u64 *nodes, uint64_t nleaves)
{
    MI_TRY(upload_rc_once(ctx));
    uint32_t nlevels = 1;
    while ((nleaves >> (nlevels - 1)) > 1) nlevels++;
    u64 *vals = nullptr;
    MI_TRY(mi_scratch(ctx, (uint64_t)nlevels * 32, (void **)&vals));
    hipLaunchKernelGGL(k_uniform_tree_values, dim3(1), dim3(64), 0, ctx->stream, vals, nlevels);
    hipLaunchKernelGGL(k_uniform_tree_fill, dim3(mi_grid_256(2 * nleaves - 1)), dim3(256), 0, ctx->stream, nodes, (const u64 *)vals, nleaves);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// parent = hash(left || right || 0^4)[0..4), one node per 16 lanes: the levels of small trees
__global__ __launch_bounds__(256) void k_merkle_level_coop(u64 *__restrict__ out, const u64 *__restrict__ in, uint64_t n_out)
{
    const uint64_t q = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const uint32_t j = threadIdx.x & 15;
    const u64 x = permute_coop((q < n_out && j < 8) ? in[q * 8 + j] : 0, j);
    if (q < n_out && j < 4) out[q * 4 + j] = x;
}

// the last levels (n <= 64 nodes) in one 512-thread workgroup, one node per 16 lanes: six dependent levels at ~20 us each instead of 58 us
// (wider levels are spread over the chip by k_merkle_level_coop: one CU would serialise their sweeps)
__global__ __launch_bounds__(512) void k_merkle_top_coop(u64 *level, uint64_t n)
{
    const uint32_t j = threadIdx.x & 15, slot = threadIdx.x >> 4; // 32 nodes per sweep
    while (n > 1) {
        const uint64_t n_out = n >> 1;
        u64 *nxt = level + n * 4;
        for (uint64_t i0 = 0; i0 < n_out; i0 += 32) { // trip count is workgroup-uniform
            const uint64_t i = i0 + slot;
            const u64 x = permute_coop((i < n_out && j < 8) ? level[i * 8 + j] : 0, j);
            if (i < n_out && j < 4) nxt[i * 4 + j] = x;
        }
        __threadfence_block();
        __syncthreads();
        level = nxt;
        n = n_out;
    }
}

// linear_hash of every row: ncols <= 4 -> copy + zero pad, else rate-8 sponge with the
// previous block's out[0..4) carried in the capacity.
template <int MDS>
__global__ __launch_bounds__(256) void k_linear_hash_rows(u64 *__restrict__ digests, const u64 *__restrict__ src,
                                                          uint64_t pitch, uint32_t ncols, uint64_t nrows)
{
    uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    const u64 *p = src + row * pitch;
    u64 *o = digests + row * 4;
    if (ncols <= 4) {
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) o[i] = i < ncols ? gl::canon(p[i]) : 0;
        return;
    }
    // Two 64-byte blocks are fetched back to back every second absorb (128 contiguous bytes per lane), so the
    // 128-byte line that straddles them is touched twice within one burst instead of one permutation apart.
    u64 s[12], b0[8], b1[8];
#pragma unroll
    for (int i = 0; i < 4; i++) s[8 + i] = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) b0[i] = i < ncols ? p[i] : 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) b1[i] = 8 + i < ncols ? p[8 + i] : 0;
    for (uint32_t c = 0; c < ncols; c += 16) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = b0[i];
        pos::permute<MDS, 0>(s, c_rc, &c_sparse);
        if (c + 8 >= ncols) break;
#pragma unroll
        for (int i = 0; i < 4; i++) s[8 + i] = s[i];
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = b1[i];
        const uint32_t c2 = c + 16;
        if (c2 < ncols) { // prefetch the next two absorbs while this permutation runs
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) b0[i] = (c2 + i < ncols) ? p[c2 + i] : 0;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) b1[i] = (c2 + 8 + i < ncols) ? p[c2 + 8 + i] : 0;
        }
        pos::permute<MDS, 0>(s, c_rc, &c_sparse);
        if (c2 >= ncols) break;
#pragma unroll
        for (int i = 0; i < 4; i++) s[8 + i] = s[i];
    }
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(o);
    o2[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    o2[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// ---- line-aligned variant of the leaf sponge (default).
// HBM is fetched in 128-byte lines, but a row of 665 elements starts at an arbitrary 8-byte offset inside its
// first line, so a lane reading "its next 64 bytes" touches most lines twice, one permutation (~35 us) apart;
// with 32768 such half-consumed lines per XCD (= the whole 4 MiB L2) a third of them are fetched again
// (measured: 1.36x the algorithmic bytes).  Here every lane fetches whole ALIGNED lines, one permutation ahead of
// their use, straight into a per-lane ring in LDS (global_load_lds_dwordx4: the data never passes through VGPRs,
// which the permutation needs all of), and takes its 8-element blocks out of the ring: each line crosses the
// fabric once.  Ring occupancy: a line is requested when fewer than 8 unconsumed elements would remain and lands
// (16 more) before the next take: at most 7 + 16 = 23 elements, hence 24 slots = 12 sixteen-byte pairs.
// 12 pairs x 256 threads x 16 B = 48 KiB per workgroup -> three workgroups (three waves per SIMD) per CU.
// All stream positions are wave-uniform (scalar control flow, LDS addresses = lane offset + scalar): wave (q, j)
// takes the rows r = 1024 q + 16 lane + j, whose offsets (r * pitch + base) mod 16 depend on j only.
static constexpr uint32_t LEAF_RING = 24;

// Requests the aligned 16-element line at `line` for every lane; it lands in ring pairs (f24 / 2 + i) mod 12, i < 8.
// wave_ring: this wave's 64 consecutive entries of pair 0 (wave-uniform).  Lines that stick out of the matrix (the
// very first / last line) are read element-wise and stored through registers instead: never read outside [lo, hi).
__device__ __forceinline__ void line_fetch(ulonglong2 *wave_ring, uint32_t lane, uint32_t f24, const u64 *line, const u64 *lo,
                                           const u64 *hi)
{
    const bool inside = line >= lo && line + 16 <= hi;
    if (__builtin_amdgcn_ballot_w64(!inside) == 0) {
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            uint32_t pr = (f24 >> 1) + i;
            pr = pr >= LEAF_RING / 2 ? pr - LEAF_RING / 2 : pr;
            // LDS destination = M0 base (wave-uniform) + lane * 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(line + 2 * i),
                                             (__attribute__((address_space(3))) void *)(wave_ring + pr * 256), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            uint32_t pr = (f24 >> 1) + i;
            pr = pr >= LEAF_RING / 2 ? pr - LEAF_RING / 2 : pr;
            ulonglong2 v;
            v.x = (line + 2 * i >= lo && line + 2 * i < hi) ? line[2 * i] : 0;
            v.y = (line + 2 * i + 1 >= lo && line + 2 * i + 1 < hi) ? line[2 * i + 1] : 0;
            wave_ring[pr * 256 + lane] = v;
        }
    }
}

// every requested line has landed in LDS (LDS-direct loads are counted by vmcnt) and is visible to the reads below
__device__ __forceinline__ void lines_landed() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// The columns of a row may come from several "slabs" (column windows stored as separate row-major matrices: the
// row-sharded multi-GPU path receives one slab per peer and never repacks them), absorbed in order; a plain matrix
// is one slab.  carry_in: the sponge capacity starts from digests[row] -- an earlier call absorbed the columns
// before these -- instead of zero.  Only the last slab may have a width that is not a multiple of 8 (zero padding
// exists at the end of a row only).
//
// EMIT (launch_linear_hash_absorb_emit): the window is a compact chunk whose rows start on 128-byte lines, so the offset is zero for
// every row and a wave may take 64 CONSECUTIVE rows -- one tile of a tile-major section [tile][column][64 rows], in which the words a
// lane absorbs are 512-byte runs of the wave (a row at gl::tile_pos of its index): each word is stored there as it is taken out of the ring (one store per word beside
// the ~2 100 instructions a word's share of the permutation costs; HBM is idle under this kernel).  That is how Starks::genProof gets
// its extended sections in the layout the constraint kernels read, without a transposing copy (host/starks.hpp).
template <int MDS, bool EMIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_linear_hash_rows_lines(
    u64 *__restrict__ digests, const LeafSlabs sl, uint64_t nrows)
{
    __shared__ ulonglong2 ring[(LEAF_RING / 2) * 256]; // [pair][thread]
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    const uint64_t row = EMIT ? gw * 64 + lane : (gw >> 4) * 1024 + (uint64_t)lane * 16 + (gw & 15);
    const bool active = row < nrows;
    // idle lanes shadow a valid row (of the same residue class when there is one: then their line requests coincide
    // with an active lane's; every load of a shadow is bounds-checked like any other)
    const uint64_t srow = active ? row : (EMIT ? lane : ((gw & 15) < nrows ? (gw & 15) : 0));
    // EMIT: this wave's tile, this lane's row of it (nrows is a multiple of 64: a wave is active or idle as a whole)
    u64 *const etile = EMIT ? sl.emit + (gw * sl.emit_cols + sl.emit_col0) * 64 + gl::tile_pos(lane) : nullptr;
    ulonglong2 *wave_ring = ring + wave * 64;
    const u64 *my = reinterpret_cast<const u64 *>(wave_ring + lane); // element in slot a: my[(a >> 1) * 512 + (a & 1)]
    u64 s[12];
    if (sl.carry_in) {
        const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(digests + srow * 4);
        const ulonglong2 c0 = c2[0], c1 = c2[1];
        s[8] = c0.x, s[9] = c0.y, s[10] = c1.x, s[11] = c1.y;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) s[8 + i] = 0;
    }
    for (uint32_t si = 0; si < sl.nslabs; si++) {
        const u64 *src = sl.base[si];
        const uint64_t pitch = sl.pitch[si];
        const uint32_t ncols = sl.width[si];
        const u64 *lo = src, *hi = src + (nrows - 1) * pitch + ncols;
        const u64 *p = src + srow * pitch;
        // offset of the row inside its 128-byte line, in elements: identical in every lane of the wave
        const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(((uintptr_t)p >> 3) & 15));
        const u64 *lp = p - o;     // the row's line stream: element e of the stream is lp[e], the row is [o, end)
        const uint32_t end = o + ncols;
        uint32_t fetched = 0, f24 = 0; // the ring holds stream elements [pos, fetched); f24 = fetched mod 24
        auto fetch_next = [&]() {      // request stream elements [fetched, fetched + 16)
            line_fetch(wave_ring, lane, f24, lp + fetched, lo, hi);
            fetched += 16;
            f24 = f24 + 16 >= LEAF_RING ? f24 + 16 - LEAF_RING : f24 + 16;
        };
        fetch_next();
        if (o > 8 && end > 16) fetch_next(); // the first block already crosses into the second line (occupancy 32 - o <= 23)
        uint32_t p24 = o; // pos mod 24
        for (uint32_t pos = o; pos < end; pos += 8) {
            lines_landed();
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) { // take the block; past the end of the row: zero padding
                uint32_t a = p24 + i;
                a = a >= LEAF_RING ? a - LEAF_RING : a;
                s[i] = (pos + i < end) ? my[(a >> 1) * 512 + (a & 1)] : 0;
            }
            if (EMIT && active) {
#pragma unroll
                for (uint32_t i = 0; i < 8; i++)
                    if (pos + i < end) etile[(uint64_t)(pos - o + i) * 64] = s[i];
            }
            p24 = p24 + 8 >= LEAF_RING ? p24 + 8 - LEAF_RING : p24 + 8;
            // fewer than 8 unconsumed elements left and the row goes on: request the next line during this permutation
            // (it lands before the next take).  The request is issued from inside the permutation, ahead of its last
            // four rounds: ~10 us of arithmetic follow, enough to hide the HBM latency.
            const bool want = (int32_t)(fetched - pos - 8) < 8 && fetched < end;
            pos::permute<MDS, 0>(s, c_rc, &c_sparse, [&]() {
                if (want) fetch_next();
            });
#pragma unroll
            for (int i = 0; i < 4; i++) s[8 + i] = s[i]; // capacity of the next block = this digest
        }
    }
    if (active) {
        ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(digests + row * 4);
        o2[0] = make_ulonglong2(gl::canon(s[8]), gl::canon(s[9]));
        o2[1] = make_ulonglong2(gl::canon(s[10]), gl::canon(s[11]));
    }
}

// parent = hash(left || right || 0^4)[0..4)
template <int MDS>
__device__ __forceinline__ void hash_pair(u64 *__restrict__ out, const u64 *__restrict__ in)
{
    const ulonglong2 *i2 = reinterpret_cast<const ulonglong2 *>(in);
    ulonglong2 a = i2[0], b = i2[1], c = i2[2], d = i2[3];
    u64 s[12] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y, 0, 0, 0, 0};
    pos::permute<MDS, 4>(s, c_rc, &c_sparse);
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(out);
    o2[0] = make_ulonglong2(s[0], s[1]);
    o2[1] = make_ulonglong2(s[2], s[3]);
}

template <int MDS>
__global__ __launch_bounds__(256) void k_merkle_level(u64 *__restrict__ out, const u64 *__restrict__ in, uint64_t n_out)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    hash_pair<MDS>(out + i * 4, in + i * 8);
}

// the last levels (n <= 512 nodes) in one workgroup: removes ~9 launch + drain latencies per tree
template <int MDS>
__global__ __launch_bounds__(256) void k_merkle_top(u64 *level, uint64_t n)
{
    while (n > 1) {
        const uint64_t n_out = n >> 1;
        u64 *nxt = level + n * 4;
        for (uint64_t i = threadIdx.x; i < n_out; i += 256) hash_pair<MDS>(nxt + i * 4, level + i * 8);
        __threadfence_block();
        __syncthreads();
        level = nxt;
        n = n_out;
    }
}

// one workgroup per query: row values then the sibling of every level (merkleTreeGL.cpp:12-35)
__global__ __launch_bounds__(64) void k_group_proofs(u64 *__restrict__ proofs, const u64 *__restrict__ nodes,
                                                     const u64 *__restrict__ src, uint64_t pitch, uint64_t height,
                                                     uint32_t width, uint32_t levels, const u64 *__restrict__ idx, uint32_t tiled)
{
    const uint64_t q = blockIdx.x;
    const uint64_t id = idx[q];
    u64 *out = proofs + q * ((uint64_t)width + 4ull * levels);
    if (tiled) // the source is a tile-major section [height / 64][pitch columns][64 rows]
        for (uint32_t i = threadIdx.x; i < width; i += 64) out[i] = gl::canon(src[((id >> 6) * pitch + i) * 64 + gl::tile_pos((uint32_t)(id & 63))]);
    else
        for (uint32_t i = threadIdx.x; i < width; i += 64) out[i] = gl::canon(src[id * pitch + i]);
    if (!nodes) return; // values only: the siblings come from elsewhere (a tree sharded over several devices, csrc/multi.hip)
    for (uint32_t e = threadIdx.x; e < levels * 4; e += 64) {
        const uint32_t l = e >> 2, k = e & 3;
        // offset of level l = 4 * (h + h/2 + ... ) = 4 * (2h - (h >> (l-1)))   for l >= 1
        const uint64_t off = l == 0 ? 0 : 4 * (2 * height - (height >> (l - 1)));
        const uint64_t sib = (id >> l) ^ 1;
        out[width + e] = nodes[off + sib * 4 + k];
    }
}

template <typename F>
static int by_variant(mi_ctx *ctx, F f)
{
    MI_TRY(upload_rc_once(ctx));
    if (ctx->poseidon_variant == pos::MDS_HALF32) f(std::integral_constant<int, pos::MDS_HALF32>());
    else if (ctx->poseidon_variant == pos::MDS_SPARSE) f(std::integral_constant<int, pos::MDS_SPARSE>());
    else f(std::integral_constant<int, pos::MDS_LIMB22>());
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// Below this many independent permutations (nodes of a level, states of a batch) the cooperative form wins: 16 lanes per state keep
// the chip busy where one lane per state would leave most of it idle behind a 58 us dependent chain (crossover measured: DESIGN.md).
// The threshold is ctx->poseidon_coop_max (common.h; mi_set_poseidon_coop_max).

int launch_permute(mi_ctx *ctx, u64 *out, const u64 *in, uint64_t count)
{
    if (count == 0) return MI_OK;
    MI_REQUIRE_1D_GRID(count);
    if (count <= ctx->poseidon_coop_max) {
        MI_TRY(upload_rc_once(ctx));
        hipLaunchKernelGGL(k_permute_coop, dim3((unsigned)((count * 16 + 255) / 256)), dim3(256), 0, ctx->stream, out, in, count);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
    const unsigned grid = (unsigned)((count + 255) / 256);
    return by_variant(ctx, [&](auto v) {
        hipLaunchKernelGGL((k_permute<decltype(v)::value>), dim3(grid), dim3(256), 0, ctx->stream, out, in, count);
    });
}

static int launch_leaf_slabs(mi_ctx *ctx, u64 *digests, const LeafSlabs &sl, uint64_t nrows)
{
    const uint64_t waves = sl.emit ? nrows / 64 : ((nrows + 1023) / 1024) * 16;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (sl.emit)
        return by_variant(ctx, [&](auto v) {
            hipLaunchKernelGGL((k_linear_hash_rows_lines<decltype(v)::value, true>), dim3(grid), dim3(256), 0, ctx->stream, digests, sl, nrows);
        });
    return by_variant(ctx, [&](auto v) {
        hipLaunchKernelGGL((k_linear_hash_rows_lines<decltype(v)::value, false>), dim3(grid), dim3(256), 0, ctx->stream, digests, sl, nrows);
    });
}

int launch_linear_hash_rows(mi_ctx *ctx, u64 *digests, const u64 *src, uint64_t pitch, uint64_t ncols, uint64_t nrows)
{
    if (nrows == 0) return MI_OK;
    MI_REQUIRE(ncols < (1ull << 30), "ncols too large");
    MI_REQUIRE_1D_GRID(nrows + 1024);
    if (ncols > 4 && ctx->leaf_line_aligned) {
        LeafSlabs sl = {};
        sl.base[0] = src;
        sl.pitch[0] = pitch;
        sl.width[0] = (uint32_t)ncols;
        sl.nslabs = 1;
        return launch_leaf_slabs(ctx, digests, sl, nrows);
    }
    const unsigned grid = (unsigned)((nrows + 255) / 256);
    return by_variant(ctx, [&](auto v) {
        hipLaunchKernelGGL((k_linear_hash_rows<decltype(v)::value>), dim3(grid), dim3(256), 0, ctx->stream, digests, src, pitch,
                           (uint32_t)ncols, nrows);
    });
}

// Streaming form of the leaf sponge: absorbs the next columns of every row, given as nslabs column windows, into
// the running capacity kept in digests[row].  first: the capacity starts at zero; final: these are the row's last
// columns (only then may the last window's width be other than a multiple of 8).
int launch_linear_hash_absorb(mi_ctx *ctx, u64 *digests, uint32_t nslabs, const u64 *const *bases, const uint64_t *pitches,
                              const uint64_t *widths, uint64_t nrows, bool first, bool final)
{
    if (nrows == 0 || nslabs == 0) return MI_OK;
    MI_REQUIRE(nslabs <= MI_MAX_SLABS, "too many column windows in one call");
    MI_REQUIRE_1D_GRID(nrows + 1024);
    LeafSlabs sl = {};
    uint32_t k = 0;
    for (uint32_t i = 0; i < nslabs; i++) {
        if (widths[i] == 0) continue; // an empty window contributes nothing
        MI_REQUIRE(widths[i] < (1ull << 30) && pitches[i] >= widths[i], "bad column window");
        sl.base[k] = bases[i];
        sl.pitch[k] = pitches[i];
        sl.width[k] = (uint32_t)widths[i];
        k++;
    }
    if (k == 0) { // nothing to absorb; a first call still has to leave the zero capacity behind for the next one
        if (first) MI_HIP_CHECK(hipMemsetAsync(digests, 0, nrows * 32, ctx->stream));
        return MI_OK;
    }
    for (uint32_t i = 0; i + 1 < k; i++) MI_REQUIRE(sl.width[i] % 8 == 0, "only a row's last column window may have a width that is not a multiple of 8");
    MI_REQUIRE(final || sl.width[k - 1] % 8 == 0, "a window that is not the row's last must have a width that is a multiple of 8");
    sl.nslabs = k;
    sl.carry_in = first ? 0 : 1;
    return launch_leaf_slabs(ctx, digests, sl, nrows);
}

int launch_linear_hash_absorb_emit(mi_ctx *ctx, u64 *digests, const u64 *base, uint64_t pitch, uint64_t width, uint64_t nrows, bool first, bool final,
                                   u64 *dst, uint64_t dst_cols, uint64_t col0)
{
    if (nrows == 0 || width == 0) return MI_OK;
    MI_REQUIRE_1D_GRID(nrows + 1024);
    MI_REQUIRE(dst && nrows % 64 == 0, "a tile-major section has a multiple of 64 rows");
    MI_REQUIRE(width < (1ull << 30) && pitch >= width && pitch % 16 == 0 && ((uintptr_t)base & 127) == 0,
               "the emitting leaf kernel reads a compact chunk whose rows start on 128-byte lines");
    MI_REQUIRE(final || width % 8 == 0, "a window that is not the row's last must have a width that is a multiple of 8");
    MI_REQUIRE(col0 + width <= dst_cols && dst_cols < (1ull << 32), "columns outside the tile-major section");
    LeafSlabs sl = {};
    sl.base[0] = base;
    sl.pitch[0] = pitch;
    sl.width[0] = (uint32_t)width;
    sl.nslabs = 1;
    sl.carry_in = first ? 0 : 1;
    sl.emit = dst;
    sl.emit_cols = (uint32_t)dst_cols;
    sl.emit_col0 = (uint32_t)col0;
    return launch_leaf_slabs(ctx, digests, sl, nrows);
}

int launch_merkle_levels(mi_ctx *ctx, u64 *nodes, uint64_t nleaves)
{
    MI_REQUIRE(is_pow2(nleaves), "number of leaves must be a power of two");
    MI_REQUIRE_1D_GRID(nleaves);
    u64 *level = nodes;
    uint64_t n = nleaves;
    const bool coop = ctx->poseidon_coop_max > 0;
    MI_TRY(upload_rc_once(ctx));
    while (n > (coop ? 64 : 512)) {
        const uint64_t n_out = n >> 1;
        u64 *nxt = level + n * 4;
        if (n_out <= ctx->poseidon_coop_max) { // a small level: one node per 16 lanes
            hipLaunchKernelGGL(k_merkle_level_coop, dim3((unsigned)((n_out * 16 + 255) / 256)), dim3(256), 0, ctx->stream, nxt, level, n_out);
            MI_HIP_CHECK(hipGetLastError());
        } else {
            const unsigned grid = (unsigned)((n_out + 255) / 256);
            MI_TRY(by_variant(ctx, [&](auto v) {
                hipLaunchKernelGGL((k_merkle_level<decltype(v)::value>), dim3(grid), dim3(256), 0, ctx->stream, nxt, level, n_out);
            }));
        }
        level = nxt;
        n = n_out;
    }
    if (n > 1) {
        if (coop) {
            hipLaunchKernelGGL(k_merkle_top_coop, dim3(1), dim3(512), 0, ctx->stream, level, n);
            MI_HIP_CHECK(hipGetLastError());
        } else {
            MI_TRY(by_variant(ctx, [&](auto v) {
                hipLaunchKernelGGL((k_merkle_top<decltype(v)::value>), dim3(1), dim3(256), 0, ctx->stream, level, n);
            }));
        }
    }
    return MI_OK;
}

int launch_group_proofs(mi_ctx *ctx, u64 *proofs, const u64 *nodes, const u64 *src, uint64_t pitch, uint64_t height,
                        uint64_t width, const u64 *idx_dev, uint64_t nq, bool tiled)
{
    if (nq == 0) return MI_OK;
    MI_REQUIRE(is_pow2(height), "tree height must be a power of two");
    MI_REQUIRE(!tiled || height % 64 == 0, "a tile-major section has a multiple of 64 rows");
    const uint32_t levels = ilog2_u64(height);
    hipLaunchKernelGGL(k_group_proofs, dim3((unsigned)nq), dim3(64), 0, ctx->stream, proofs, nodes, src, pitch, height,
                       (uint32_t)width, levels, idx_dev, tiled ? 1u : 0u);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ---- host-side debug hook: the same inline permutation, run on the CPU (tests only)
extern "C" void mi_dbg_host_poseidon_permute(uint64_t st[12], int variant)
{
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = st[i];
    static pos::SparseTables tables;
    static bool filled = false;
    if (!filled) { pos::fill_sparse_tables(tables); filled = true; }
    if (variant == pos::MDS_HALF32) pos::permute<pos::MDS_HALF32>(s, (const u64 *)MI_POS_RC);
    else if (variant == pos::MDS_SPARSE) pos::permute<pos::MDS_SPARSE>(s, (const u64 *)MI_POS_RC, &tables);
    else pos::permute<pos::MDS_LIMB22>(s, (const u64 *)MI_POS_RC);
    for (int i = 0; i < 12; i++) st[i] = s[i];
}
