/*
 * mi_stark.h -- C ABI of the MI355X-native STARK hot path (libmi_stark.so).
 *
 * This is the drop-in boundary for the zkevm-prover's batch-proof loop.  The reference has no FFI:
 * its boundary is the link-time C++ surface of the (un-vendored) src/goldilocks submodule plus three
 * src/starkpil classes.  Each entry point below names the reference interface it
 * replaces; the C++ header shims in merlin-zkevm-prover_amd/host/ (NTT_Goldilocks, PoseidonGoldilocks,
 * MerkleTreeGL, FRIProve, Transcript ...) forward to these, so src/starkpil compiles unchanged.
 *
 * Conventions
 *   - Elements are uint64_t Goldilocks values (p = 2^64 - 2^32 + 1), row-major: element (row r, col c)
 *     of a matrix lives at base[r * pitch + c] (commit_pols.hpp:1461, stark_info.cpp:473-482).
 *     Inputs may be any u64 encoding; outputs are canonical (< p).
 *   - "_dev" functions take HBM pointers (hipMalloc / torch device memory) and enqueue work on the
 *     context's stream without synchronising.  Functions without the suffix take HOST pointers exactly
 *     like the reference's methods, stage through HBM and return after the result is back in host memory.
 *   - The caller owns every buffer; the library never frees or retains them.
 *   - Return value: 0 on success, negative mi_status otherwise; mi_last_error() gives the message.
 *     The reference's convention is log + exitProcess() (starks.hpp:99-100): the C++ shims map a
 *     non-zero status onto that.  There is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with MI_ERR_NO_DEVICE.
 */
#ifndef MI_STARK_H
#define MI_STARK_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx mi_ctx;

typedef enum {
    MI_OK = 0,
    MI_ERR_NO_DEVICE = -1,
    MI_ERR_INVALID = -2,   /* bad argument (non power-of-two size, null pointer, ...) */
    MI_ERR_HIP = -3,       /* a HIP runtime call failed */
    MI_ERR_NOMEM = -4,
    MI_ERR_UNSUPPORTED = -5
} mi_status;

#define MI_HASH_SIZE 4        /* HASH_SIZE / CAPACITY   (merklehash_goldilocks.hpp) */
#define MI_RATE 8
#define MI_SPONGE_WIDTH 12
#define MI_FIELD_EXTENSION 3  /* FIELD_EXTENSION        (goldilocks_cubic_extension.hpp) */

/* ------------------------------------------------------------------ context */
/* One context per GPU/process (one process per GPU).  device < 0 keeps the current device. */
int mi_ctx_create(mi_ctx **out, int device);
void mi_ctx_destroy(mi_ctx *ctx);
/* Use a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL (the initial
 * value) is the device's default stream. */
int mi_ctx_device(const mi_ctx *ctx); /* the HIP device this context runs on */
int mi_ctx_set_stream(mi_ctx *ctx, void *hip_stream);
int mi_ctx_sync(mi_ctx *ctx);
/* Scratch HBM the library may allocate lazily for NTT/LDE ping-pong buffers (default 32 GiB,
 * clamped to what the problem needs).  Wide LDEs are processed in column chunks that fit it. */
int mi_ctx_set_workspace_limit(mi_ctx *ctx, uint64_t bytes);
/* The `buf` argument of NTT_Goldilocks::extendPol / NTT (starks.cpp:52 lends p_cm2_2ns, :133,214 pBuffer): device scratch the
 * CALLER owns -- e.g. a section of its polynomial area that is not live yet.  While a buffer is lent (16-byte aligned, >= 1 MiB)
 * every transform sizes its column chunks to it and allocates nothing; its contents are undefined afterwards.  ptr = NULL
 * returns to the context's own workspace.  Synchronises the stream. */
int mi_ctx_lend_workspace(mi_ctx *ctx, void *ptr, uint64_t bytes);
/* hipMemGetInfo of the context's device (the HBM plan of host/starks.hpp is checked against it) */
int mi_dev_mem_info(mi_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
const char *mi_last_error(void);
const char *mi_version(void);
int mi_device_count(void);

/* ------------------------------------------------------------------ NTT / LDE
 * Replaces NTT_Goldilocks::{NTT, INTT, extendPol} (call sites starks.cpp:52,133,214,261,284,325-326;
 * friProve.cpp:100-102).  Natural order in and out; columns independent; dst == src allowed.
 * The reference's buffer / nphase / nblock arguments are CPU blocking hints and have no equivalent. */
int mi_ntt_dev(mi_ctx *ctx, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src, uint64_t src_pitch,
               uint64_t n, uint64_t ncols, int inverse);
/* out[i] = P_col(shift * w_ext^i), i < n_ext  (= INTT_n, scale by shift^k, zero-pad, NTT_n_ext) */
int mi_lde_dev(mi_ctx *ctx, uint64_t *out, uint64_t out_pitch, const uint64_t *in, uint64_t in_pitch,
               uint64_t n_ext, uint64_t n, uint64_t ncols);
int mi_ntt(mi_ctx *ctx, uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t ncols, int inverse);
int mi_lde(mi_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t n_ext, uint64_t n, uint64_t ncols);

/* ------------------------------------------------------------------ Poseidon / Merkle
 * Replaces PoseidonGoldilocks::{hash_full_result, hash, linear_hash, merkletree_avx, merkletree_avx512,
 * merkletree} (transcript.cpp:23,46; merkleTreeGL.cpp:37-44; build_const_tree.cpp:382). */
int mi_poseidon_hash_full_result(mi_ctx *ctx, uint64_t out[12], const uint64_t in[12]);          /* host ptrs */
int mi_poseidon_hash(mi_ctx *ctx, uint64_t out[4], const uint64_t in[12]);
/* Transcript::put (transcript.cpp:4-29) as one call: the class's members (state, pending, out, the two cursors; HOST pointers, updated in
 * place) absorb n input elements; every completed block of 8 is hashed on the device, all inside one launch. */
int mi_transcript_put(mi_ctx *ctx, uint64_t state[4], uint64_t pending[8], uint64_t out[12], uint32_t *pending_cursor,
                      uint32_t *out_cursor, const uint64_t *input, uint64_t n);                       /* host ptrs */
int mi_poseidon_linear_hash(mi_ctx *ctx, uint64_t out[4], const uint64_t *in, uint64_t size);    /* host ptrs */
/* count independent permutations, in/out: count x 12 (device) */
int mi_poseidon_permute_dev(mi_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t count);
/* digests[r*4..] = linear_hash(row r)  (device) */
int mi_linear_hash_rows_dev(mi_ctx *ctx, uint64_t *digests, const uint64_t *src, uint64_t pitch,
                            uint64_t ncols, uint64_t nrows);
/* Streaming form of the same sponge, for rows whose columns arrive in pieces (the row-sharded multi-GPU path
 * receives one column window per peer and per pipeline round and never repacks them): absorbs, for every row r,
 * the columns of nwindows column windows in order -- window i holds widths[i] consecutive columns, row r at
 * bases[i] + r * pitches[i] (device pointers; the three arrays themselves are host arrays of nwindows entries,
 * nwindows <= 16) -- into the running capacity kept in digests[r*4..].  first != 0: start from the zero capacity
 * (the row's first columns); final != 0: these are the row's last columns, only then may the last window's width
 * be other than a multiple of 8.  After the final call digests[r*4..] = linear_hash(row r) provided the row has
 * more than 4 columns in total (linear_hash copies shorter rows instead of hashing them: use the entry point
 * above for those). */
int mi_linear_hash_absorb_dev(mi_ctx *ctx, uint64_t *digests, uint32_t nwindows, const uint64_t *const *bases,
                              const uint64_t *pitches, const uint64_t *widths, uint64_t nrows, int first, int final);
/* nodes: (2*nrows-1)*4 u64, leaves first then each level appended, root = last 4
 * (merkleTreeGL.hpp:61-68).  nrows must be a power of two. */
int mi_merkle_build_dev(mi_ctx *ctx, uint64_t *nodes, const uint64_t *src, uint64_t pitch,
                        uint64_t ncols, uint64_t nrows);
/* Finish a tree whose level-0 digests (nleaves*4) are already at nodes[0..]; used by the multi-GPU path
 * to hash the top log2(G) levels over the all-gathered subtree roots. */
int mi_merkle_levels_dev(mi_ctx *ctx, uint64_t *nodes, uint64_t nleaves);
int mi_merkle_build(mi_ctx *ctx, uint64_t *nodes, const uint64_t *src, uint64_t ncols, uint64_t nrows);
static inline uint64_t mi_merkle_num_nodes_elems(uint64_t nrows) { return (2 * nrows - 1) * MI_HASH_SIZE; }
static inline uint64_t mi_merkle_proof_levels(uint64_t nrows)
{
    uint64_t l = 0;
    while ((1ULL << l) < nrows) l++;
    return l;
}
/* Replaces MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35) for a batch of queries:
 * proofs[q] = row idx[q] (width values) followed by levels x 4 siblings; proof stride = width + 4*levels.
 * idx is a HOST array; src/nodes/proofs are device pointers.  width = 0 (src may then be NULL) returns the sibling paths
 * alone: the row-sharded multi-GPU tree opens a row's values from its column windows and its path from a rank's subtree.  nodes = NULL
 * returns the row values alone (the sibling words of the proofs are left untouched). */
int mi_merkle_group_proofs_dev(mi_ctx *ctx, uint64_t *proofs, const uint64_t *nodes, const uint64_t *src,
                               uint64_t pitch, uint64_t height, uint64_t width, const uint64_t *idx,
                               uint64_t nqueries);
/* The same over a TILE-MAJOR source ([height / 64][ncols_total][64], mi_lde_merkle_dev_tiled): the first `width` columns of row idx[q]. */
int mi_merkle_group_proofs_tiled_dev(mi_ctx *ctx, uint64_t *proofs, const uint64_t *nodes, const uint64_t *src_tiled,
                                     uint64_t ncols_total, uint64_t height, uint64_t width, const uint64_t *idx, uint64_t nqueries);

/* mi_lde_merkle_host, how a column chunk crosses PCIe: `threads` host threads pack it out of the row-major host trace into
 * page-locked staging (streaming stores), which then moves as one contiguous copy at the full rate -- a strided 2-D copy of a
 * 32- / 64- / 128-column chunk runs at 39 / 49 / 53 GB/s against 57 -- in 32-column chunks.  -1 (default) = min(16, hardware
 * threads) when the host has at least 8 of them, else 0; 0 = strided 2-D copies on two copy streams, no host threads
 * (measured per zkEVM stage-1 step: 0.86 s packed by 12 or more threads, 0.91 s by 8, 1.00 s with 2-D copies). */
int mi_set_host_pack_threads(mi_ctx *ctx, int threads);
int mi_get_host_pack_threads(mi_ctx *ctx); /* the count in effect (the default resolved against this host's threads) */
/* ------------------------------------------------------------------ stage driver (host trace in, resident result out)
 * Starks::genProof step 1 (starks.cpp:48-59: extendPol of p_cm1_n, then treesGL[0]->merkelize()) for a caller that holds
 * the trace in HOST memory and wants the extension and the tree to STAY on the device: the n x ncols row-major host trace is
 * uploaded in column chunks of at most chunk_cols (0 = 64; a multiple of 8) on two copy streams while the chunks already on the device
 * are extended into ext (device, n_ext x ncols at row pitch ext_pitch) and their columns absorbed into the leaf sponges, so
 * the PCIe transfer overlaps the kernels.  nodes (device, (2 n_ext - 1) * 4) receives the tree; root = its last 4 elements.
 * Work is enqueued; mi_ctx_sync (or reading the root) waits for it.  For full PCIe speed the host range should be
 * page-locked (mi_host_register once per buffer, or hipHostMalloc). */
int mi_lde_merkle_host(mi_ctx *ctx, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, const uint64_t *trace_host,
                       uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols);
/* The same, and the uploaded base-domain section stays in HBM as well: base (device, n x ncols at row pitch base_pitch) receives
 * the trace itself (canonical) -- what the base-domain steps of stages 2 and 3 read (starks.cpp:66-90,150-210 over p_cm1_n). */
int mi_lde_merkle_host_keep(mi_ctx *ctx, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, uint64_t *base, uint64_t base_pitch,
                            const uint64_t *trace_host, uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols);
/* The same with the base-domain section kept TILE-MAJOR (n x ncols, n a multiple of 64; see mi_chelpers_set_tiled_section): the layout
 * the compiled base-domain steps read in place.  Costs what the row-major copy costs and hides behind the upload like it. */
int mi_lde_merkle_host_keep_tiled(mi_ctx *ctx, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, uint64_t *base_tiled,
                                  const uint64_t *trace_host, uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols);
/* The extension TILE-MAJOR (ext_tiled: n_ext x ncols, n_ext a multiple of 64, ncols > 4); base (may be NULL) receives the trace itself,
 * row-major at base_pitch, or tile-major too with base_pitch = 0 (n a multiple of 64 then).  The extension of a
 * column chunk goes into a compact buffer (out of the lent workspace's tail) and the leaf kernel, absorbing the chunk's words, writes
 * them into the section -- a lane owns one of 64 consecutive rows, so a word of every lane is a 512-byte run of a tile -- where the
 * row-major form (extendPol's: starks.cpp:52) would have been written.  What Starks::genProof's constraint kernels, its linear
 * combination, evmap and openings read in place (host/starks.hpp); the tree is the same tree. */
int mi_lde_merkle_host_tiled(mi_ctx *ctx, uint64_t *nodes, uint64_t *ext_tiled, uint64_t *base, uint64_t base_pitch,
                             const uint64_t *trace_host, uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols);
/* The same for a section that is in HBM already (src: n x ncols at row pitch src_pitch, row-major): starks.cpp:133-138,214-219. */
int mi_lde_merkle_dev_tiled(mi_ctx *ctx, uint64_t *nodes, uint64_t *ext_tiled, const uint64_t *src, uint64_t src_pitch, uint64_t n,
                            uint64_t n_ext, uint64_t ncols);
int mi_host_register(mi_ctx *ctx, void *p, uint64_t bytes);   /* hipHostRegister (portable: every device may read it): page-lock a host range for DMA */
int mi_host_unregister(mi_ctx *ctx, void *p);

/* ------------------------------------------------------------------ one process, several devices (csrc/multi.hip)
 * The stage commit -- NTT_Goldilocks::extendPol + MerkleTreeGL::merkelize of one section (starks.cpp:52-57,133-138,214-219) -- sharded over
 * the GPUs of a node from ONE process, the form the reference's Prover has (one process, one proof in flight: prover.cpp:187-260;
 * merlin-zkevm-prover_amd/shard.py is the same plan as torch.distributed ranks).  A shard is a device; several shards may name the same
 * device (a single-GPU box rehearses the path that way).  Columns are dealt to the shards in rounds of tiles of at most 32, each shard
 * extends its tile, the tile's rows go to the shards that own them (hipMemcpyPeerAsync per peer: all xGMI links at once), every shard
 * absorbs the rounds' columns of its rows into the leaf sponges as they arrive and builds the subtree over its n_ext / G rows; the G
 * subtree roots are hashed on shard 0.  Result: the single-device tree, node for node. */
typedef struct mi_multi mi_multi;
typedef struct mi_multi_tree mi_multi_tree;
int mi_multi_create(mi_multi **out, const int *devices, int n_shards /* a power of two, at most 16 */);
/* group_same_device != 0: the shards that name one physical device form a GROUP -- one set of streams, one tile ring, staging and NTT
 * workspace, shared per-context pools, everything ordered by the one stream -- so that a one-GPU box can rehearse eight shards at the
 * full 2^23 rows (8 x 36 GB of per-shard buffers do not fit beside the proof).  With one shard per device it changes nothing.
 * mi_multi_create takes the choice from MI_MULTI_GROUP_SAME_DEVICE=1. */
int mi_multi_create2(mi_multi **out, const int *devices, int n_shards, int group_same_device);
int mi_multi_lead(const mi_multi *m, int shard); /* the first shard of this shard's device group (itself when ungrouped) */
void mi_multi_destroy(mi_multi *m);
int mi_multi_shards(const mi_multi *m);
/* What the driver answered when mi_multi_create asked for direct (xGMI) access between the shards' devices: out (G x G ints, may be
 * NULL) [a * G + b] = 2 same device, 1 peer access enabled, 0 the devices cannot reach each other directly, -1 enabling it failed.
 * Returns the number of ordered pairs that are neither 1 nor 2 (their exchanges are staged through host memory by the runtime);
 * warning (optional, warning_cap bytes) receives the sentence mi_multi_create printed about them, "" when every pair is direct. */
int mi_multi_peer_access(const mi_multi *m, int *out, char *warning, uint64_t warning_cap);
mi_ctx *mi_multi_ctx(mi_multi *m, int shard);
int mi_multi_set_pack_threads(mi_multi *m, int threads); /* host threads that pack a tile for its upload (default min(64, hardware threads)) */
/* How a HOST source reaches the shards: -1 (default) = a page-locked source (mi_host_register, hipHostMalloc) is read in place by each
 * device's own DMA engines, strided 2-D copies over that device's PCIe link and no host thread in the data path, when the shards sit on
 * at least two devices; a pageable source, or shards sharing one device, is packed by host threads into page-locked staging.  0 / 1 force the packed / the strided form (1 on a pageable source: the commit refuses).
 * The environment's MI_MULTI_UPLOAD=packed|strided overrides it (A/B runs). */
int mi_multi_set_upload_mode(mi_multi *m, int mode);
int mi_multi_last_upload_mode(const mi_multi *m); /* the last commit: -1 device source, 0 host-packed, 1 strided DMA out of the page-locked source */
/* src: the n x ncols row-major base-domain section at row pitch src_pitch (elements): HOST memory when src_device < 0 (page-locked: read
 * in place by the DMA engines of the shard that extends the tile; pageable is fine too: host threads pack each tile into page-locked
 * staging and it crosses that shard's PCIe link), else memory of that device.  image / base (either may be NULL; on device image_device): receive the whole extension (n_ext x ncols at row pitch
 * image_pitch) and the section itself (n x ncols at base_pitch), row-major -- what a caller that evaluates constraints on one device
 * keeps.  root: 4 words (host).  The call returns when everything has arrived.  *out: the sharded tree, for the openings. */
int mi_multi_commit(mi_multi *m, mi_multi_tree **out, const uint64_t *src, uint64_t src_pitch, int src_device, uint64_t n, uint64_t n_ext,
                    uint64_t ncols, uint64_t *image, uint64_t image_pitch, uint64_t *base, uint64_t base_pitch, int image_device, uint64_t root[4]);
/* MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35) for nq rows: proofs (HOST) = nq x (ncols + 4 * log2(n_ext)) words, the row's
 * values (zeros with with_values == 0: a caller that kept the image opens them there) then the siblings, leaves upward */
int mi_multi_group_proofs(mi_multi_tree *t, uint64_t *proofs, const uint64_t *idx, uint64_t nq, int with_values);
/* the NEXT commit carves shard `shard`'s row buffers, staging and NTT workspace from [ptr, ptr + bytes) (memory of that shard's device, not
 * live for the duration of the commit) instead of allocating them: for a caller whose own plan fills the device (host/starks.hpp) */
int mi_multi_lend(mi_multi *m, int shard, void *ptr, uint64_t bytes);
/* the NEXT commit also leaves shard q's own rows [q R, (q + 1) R) of the extension and the halo_rows rows after them (wrapping at n_ext:
 * the shifted reads of a constraint program, starks.cpp:240) row-major at row pitch `pitch` in imgs[q] -- the start of a FULL-HEIGHT
 * section (n_ext x pitch) in memory of shard q's device, of which only those rows are written -- for every non-null imgs[q] (G entries):
 * what a row-sharded constraint evaluation on that device reads (host/chelpers_steps.hpp: step42ns over its rows on every device) */
int mi_multi_set_row_images(mi_multi *m, uint64_t *const *imgs, uint64_t pitch, uint64_t halo_rows);
/* the NEXT commit is TRANSIENT: a row image for EVERY shard (mi_multi_set_row_images, no whole-extension image), no row values will be
 * opened from its tree.  A tile's rows are then written ONCE, by a kernel of the extending shard, straight into each owner's row image
 * and absorbed there at the image's pitch (no contiguous windows: half the bytes on the links); the extended tiles live in a ring of
 * two.  Per device group it takes mi_multi_transient_need(...) elements (21.5 GB at 2^23 x 665 and 8 shards against 36.5 GB): lent to the
 * group's leader (mi_multi_lend) or from its pool.  What a row-sharded Starks::genProof asks for (host/starks.hpp). */
int mi_multi_set_transient(mi_multi *m, int on);
uint64_t mi_multi_transient_need(uint64_t n, uint64_t n_ext, uint64_t ncols, uint32_t shards);
uint64_t mi_multi_windowed_need(uint64_t n, uint64_t n_ext, uint64_t ncols, uint32_t shards); /* elements one shard of an ordinary commit takes (tiles, row windows, staging, workspace) */
/* MI_MULTI_CHECK=1 (environment, read once): logical-shard discipline.  Every device range the library allocates for a shard
 * (mi_dev_alloc and mi_vmm_reserve through that shard's context, the buffers of csrc/multi.hip) is entered with its shard; an entry
 * point then refuses (MI_ERR_INVALID, the reason on stderr) a device pointer that belongs to ANOTHER shard than the one its context
 * works for, a compiled program run for two shards, an event recorded on another group's stream; operands that legitimately cross
 * (mi_multi_copy, row images, a device source) are declared by the call that takes them.  So that a one-GPU rehearsal, where every
 * shard is device 0 and every pointer is valid everywhere, fails where an eight-GPU node would.  out: [enabled, checks made, pointers
 * nobody entered (they pass), violations].  mi_multi_own enters (bytes > 0) or withdraws a range a caller allocated itself. */
int mi_multi_check_stats(uint64_t out[4]);
int mi_multi_own(const void *p, uint64_t bytes, int shard, const char *what);
int mi_multi_set_device(mi_multi *m, int shard); /* the calling thread's current device := the shard's (to drive mi_multi_ctx(m, shard) directly) */
int mi_multi_copy(mi_multi *m, void *dst, int dst_shard, const void *src, int src_shard, uint64_t bytes); /* behind the work queued on src_shard's context */
int mi_multi_sync(mi_multi *m, int shard); /* wait for the shard's context */
int mi_multi_tree_release_rows(mi_multi_tree *t); /* give the shards' row buffers back, keep the subtrees (siblings can still be opened) */
void mi_multi_tree_free(mi_multi_tree *t);
int mi_multi_tree_info(const mi_multi_tree *t, uint64_t out[6]); /* shards, rows per shard, columns per shard, rounds, columns, rows */
const uint64_t *mi_multi_tree_nodes(const mi_multi_tree *t, int shard); /* device pointer: the shard's subtree, level-0 digests first */
int mi_multi_gather_rows(mi_multi_tree *t, uint64_t *out_host, uint64_t row0, uint64_t nrows); /* rows of the sharded extension, row-major */
/* per shard of the last commit: lde_ms, absorb_ms, exchange_wait_ms (the compute stream stood still for the links), host_pack_ms, bytes
 * sent to shard 0 .. G-1: (4 + G) doubles per shard */
int mi_multi_last_stats(const mi_multi *m, double *out, double *wall_ms);
/* the dealing of columns to shards (no device needed): out = [rounds, columns per shard, rows per shard, then per round and shard (first
 * global column, width)]; returns the words written, or minus the words needed */
int64_t mi_multi_plan_debug(uint64_t n, uint64_t n_ext, uint64_t ncols, uint32_t shards, uint64_t *out, uint64_t cap);

/* ------------------------------------------------------------------ FRI
 * Replaces the fold loop of FRIProve::prove (friProve.cpp:44-108): pol holds 2^prev_bits cubic-extension
 * elements (3 u64 each), out receives 2^cur_bits.  nbits_ext is the size of the first FRI domain, so the
 * coset factor is shift^-(2^(nbits_ext - prev_bits)) (friProve.cpp:143-147). */
int mi_fri_fold_dev(mi_ctx *ctx, uint64_t *out, const uint64_t *pol, unsigned prev_bits, unsigned cur_bits,
                    unsigned nbits_ext, const uint64_t special_x[3] /* host */);
/* The same fold restricted to outputs g in [g0, g0 + g_count) (out is still the base of the whole folded polynomial): the
 * outputs are independent, so the ranks of a multi-GPU run fold disjoint ranges and all-gather them. */
int mi_fri_fold_range_dev(mi_ctx *ctx, uint64_t *out, const uint64_t *pol, unsigned prev_bits, unsigned cur_bits,
                          unsigned nbits_ext, const uint64_t special_x[3] /* host */, uint64_t g0, uint64_t g_count);
/* FRIProve::getTransposed (friProve.cpp:252-271): aux[i*h+j] = pol[j*w+i], w = 2^transpose_bits */
int mi_fri_transpose_dev(mi_ctx *ctx, uint64_t *aux, const uint64_t *pol, uint64_t degree, unsigned transpose_bits);

/* ------------------------------------------------------------------ the rest of Starks::genProof's
 * device-friendly loops */
/* step-4 split (starks.cpp:265-280): qq2 (n_ext x qdeg*3, zero-filled by the call) from qq1 (n_ext x 3) */
int mi_q_split_dev(mi_ctx *ctx, uint64_t *qq2, const uint64_t *qq1, uint64_t n, uint64_t n_ext, unsigned qdeg);
/* evmap (starks.cpp:555-668): evals[i] = sum_k L[k] * pol_i[(k << ext_bits) * stride_i], L = lpev if
 * prime[i] else lev.  pol_ptr are device pointers to element (row 0) of each polynomial; the four
 * descriptor arrays are HOST arrays of length n_evals. */
int mi_evmap_dev(mi_ctx *ctx, uint64_t *evals /* device, n_evals*3 */, uint64_t n_evals, uint64_t n,
                 unsigned ext_bits, const uint64_t *const *pol_ptr, const uint32_t *pol_dim,
                 const uint64_t *pol_stride, const uint8_t *prime, const uint64_t *lev, const uint64_t *lpev);
/* The same with some polynomials in TILE-MAJOR sections: tile_cols[i] != 0 is the width of polynomial i's section ([n_ext / 64]
 * [tile_cols[i]][64]) and pol_ptr[i] its element of row 0 (section + 64 * column); pol_stride[i] is ignored for it.  Such polynomials
 * are summed by a kernel whose lanes take rows (a column's rows are the contiguous ones there); the evaluations are the same. */
int mi_evmap_tiled_dev(mi_ctx *ctx, uint64_t *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits, const uint64_t *const *pol_ptr,
                       const uint32_t *pol_dim, const uint64_t *pol_stride, const uint8_t *prime, const uint64_t *tile_cols,
                       const uint64_t *lev, const uint64_t *lpev);
/* the same sums over rows [row0, row0 + nrows) of the base domain only (lev / lpev still indexed by the absolute row): a row shard's
 * share -- the shares of a partition of [0, n) add up (in F_p^3) to mi_evmap_dev's evaluations */
int mi_evmap_range_dev(mi_ctx *ctx, uint64_t *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits, const uint64_t *const *pol_ptr,
                       const uint32_t *pol_dim, const uint64_t *pol_stride, const uint8_t *prime, const uint64_t *lev, const uint64_t *lpev,
                       uint64_t row0, uint64_t nrows);
/* element-wise inverse of n cubic-extension elements (Polinomial::batchInverse[Parallel],
 * polinomial.hpp:612-720); res == src allowed; inverse of 0 is 0 */
int mi_batch_inverse3_dev(mi_ctx *ctx, uint64_t *res, const uint64_t *src, uint64_t n);
/* plookup h1 / h2 of one lookup (Polinomial::calculateH1H2_opt1 / _opt3, polinomial.hpp:349-584; plain form :303-347; called at
 * starks.cpp:92-128 between step2prev and the stage-2 commitment).  Device pointers to strided views as Polinomial: element i at
 * p[i * stride .. + dim), dim 1 or 3, n rows in each of f, t, h1, h2.  Every row of t counts once plus once per row of f with the
 * same value (credited to the last such row of t); walking t in order, each row repeated by its count, yields h1[0], h2[0],
 * h1[1], ...  A row of f whose value is not in t fails the call like the reference does (MI_ERR_INVALID, mi_last_error():
 * "calculateH1H2: number not included: w=<row>"); h1 / h2 are then left untouched.  The transposes the reference wraps around
 * it (transposeH1H2Columns / Rows, starks.cpp:405-455) are not needed: the views are read in place. */
int mi_calculate_h1h2_dev(mi_ctx *ctx, uint64_t *h1, uint64_t h1_stride, uint64_t *h2, uint64_t h2_stride, const uint64_t *f,
                          uint64_t f_stride, const uint64_t *t, uint64_t t_stride, unsigned dim, uint64_t n);
/* grand product (Polinomial::calculateZ, polinomial.hpp:586-607; starks.cpp:174-187 between step3prev and step3): z[0] = 1,
 * z[i] = z[i-1] * num[i-1] / den[i-1] in the cubic extension; strided device views of dim 3.  *closes (HOST, optional) receives 1
 * when z[n-1] * num[n-1] / den[n-1] == 1 -- the reference's zkassert -- else 0.  z may not overlap num or den (it holds the
 * quotients between the two passes). */
int mi_calculate_z_dev(mi_ctx *ctx, uint64_t *z, uint64_t z_stride, const uint64_t *num, uint64_t num_stride, const uint64_t *den,
                       uint64_t den_stride, uint64_t n, int *closes);
/* nprod grand products over the same n rows in one pass (the loop of starks.cpp:174-187 over puCtx / peCtx / ciCtx: every product of
 * the stage reads 2 x 3 words of the same rows of tmpExp_n, so one pass over those rows serves them all).  z / num / den and the
 * strides are HOST arrays of nprod device pointers / word strides; results identical to nprod calls of mi_calculate_z_dev.
 * closes: HOST array of nprod flags, optional. */
int mi_calculate_z_batch_dev(mi_ctx *ctx, unsigned nprod, uint64_t *const *z, const uint64_t *z_stride, const uint64_t *const *num,
                             const uint64_t *num_stride, const uint64_t *const *den, const uint64_t *den_stride, uint64_t n, int *closes);
/* out[i] = start * ratio^i  (x_n, x_2ns: starks.hpp:149-160,176-183) */
int mi_geom_seq_dev(mi_ctx *ctx, uint64_t *out, uint64_t n, uint64_t start, uint64_t ratio);
/* out[k] = ratio^k in the cubic extension (LEv / LpEv: starks.cpp:311-323) */
int mi_geom_seq3_dev(mi_ctx *ctx, uint64_t *out, uint64_t n, const uint64_t ratio[3] /* host */);
/* xDivXSubXi (starks.cpp:350-365): out[k] = x[k] / (x[k] - xi) in the cubic extension, x base field */
int mi_x_div_x_sub_dev(mi_ctx *ctx, uint64_t *out, const uint64_t *x, uint64_t n, const uint64_t xi[3] /* host */);
/* ZhInv table (zhInv.cpp:7-31), HOST output of 2^(nbits_ext-nbits) values */
int mi_zhinv(mi_ctx *ctx, uint64_t *out, unsigned nbits, unsigned nbits_ext);

/* ------------------------------------------------------------------ constraint evaluators ("chelpers")
 * Replaces ZkevmSteps::step42ns_parser_first_avx / _avx512 (zkevm.chelpers.step42ns.parser.cpp:10-760; call site
 * starks.cpp:237-248): the reference's generated PROGRAM -- the tables op42[NOPS_] / args42[NARGS_] of
 * zkevm.chelpers.step42ns.parser.hpp, handed over as data -- is run once per row of the extended domain on the GPU, so that
 * the extended polynomials never leave HBM for this step.  mi_chelpers_compile translates the tables once (host side:
 * decode by the reference interpreter's opcode numbering, copy forwarding, Sethi-Ullman reschedule, temp re-allocation);
 * a null ctx compiles for mi_dbg_host_chelpers_run only.  `step` names the opcode numbering of the tables. */
#define MI_CHELPERS_STEP42NS 42
#define MI_CHELPERS_STEP52NS 52 /* ZkevmSteps::step52ns_parser_first_avx (zkevm.chelpers.step52ns.parser.cpp; starks.cpp:370): the FRI polynomial f_2ns */
/* The base-domain steps (zkevm.chelpers.step{2prev,3prev,3}.parser.cpp; starks.cpp:93,165,178): ONE opcode numbering for the three --
 * 0-83 as step42ns (with params.pConstPols / x_n), 84-114 the forms whose result is stored into params.pols at the row or at a
 * shifted row, 115 a fusion.  They write params.pols (pass the same device memory, it is read and written; q / zhinv are unused,
 * const_pols = pConstPols over the N base-domain rows, nrows_ext = N) and run through the compiled kernels only:
 * mi_chelpers_build_native before mi_chelpers_run_dev. */
#define MI_CHELPERS_STEP2PREV 20
#define MI_CHELPERS_STEP3PREV 30
#define MI_CHELPERS_STEP3 31
typedef struct mi_chelpers_prog mi_chelpers_prog;
/* A section of the polynomial area the program reads (stark_info mapOffsets / mapSectionsN, e.g. cm1_2ns): element (row, col)
 * at pols[offset + row * ncols + col], row < nrows.  Every polynomial operand of the program must lie in a declared section:
 * the kernel stages the sections, 64 rows at a time, in column-major order so that its per-row reads are coalesced. */
typedef struct { uint64_t offset, ncols, nrows; } mi_chelpers_section;
typedef struct {
    uint64_t *pols;             /* device: params.pols, the base every polynomial offset of the program is relative to; the base-domain
                                 * steps (STEP2PREV / 3PREV / 3) also WRITE their results into it */
    const uint64_t *const_pols; /* device: params.pConstPols2ns, element (col, row) at const_pols[col + row * n_const] */
    uint64_t n_const;           /* pConstPols2ns->numPols() */
    const uint64_t *challenges; /* HOST: params.challenges, n_challenges x 3 */
    uint64_t n_challenges;
    const uint64_t *publics;    /* HOST: params.publicInputs */
    uint64_t n_publics;
    const uint64_t *x;          /* device: params.x_2ns, element i at x[i * x_stride] (Polinomial offset()) */
    uint64_t x_stride;
    const uint64_t *zhinv;      /* HOST: the ZhInv table (mi_zhinv); zi.zhInv(i) = zhinv[i % n_zhinv] (zhInv.hpp:22-25) */
    uint64_t n_zhinv;
    uint64_t *q;                /* device: params.q_2ns, row i at q[3 i .. 3 i + 3) (step42ns output) */
    /* step52ns only */
    const uint64_t *evals;      /* HOST: params.evals, n_evals x 3 */
    uint64_t n_evals;
    const uint64_t *xdiv;       /* device: params.xDivXSubXi, rows x 3 */
    const uint64_t *xdivw;      /* device: params.xDivXSubWXi, rows x 3 */
    uint64_t *f;                /* device: params.f_2ns, row i at f[3 i .. 3 i + 3) (step52ns output) */
} mi_chelpers_params;
/* sections: the n_sections (<= 4) sections of params.pols the program reads; n_const = pConstPols2ns->numPols();
 * nrows_ext = rows of the extended domain (the constant polynomials and x_2ns have that many rows). */
int mi_chelpers_compile(mi_ctx *ctx, mi_chelpers_prog **out, int step, const uint64_t *ops, uint64_t nops,
                        const uint64_t *args, uint64_t nargs, const mi_chelpers_section *sections, uint64_t n_sections,
                        uint64_t n_const, uint64_t nrows_ext);
/* The same, from the program as FIELD OPERATIONS rather than as one of the reference's opcode tables: what host/steps_tracer.hpp
 * records when it runs a Steps class's generated per-row code (recursive1.chelpers.step3.cpp etc.: straight-line
 * Goldilocks::add / sub / mul / copy, Goldilocks3::... on params.pols, constants, challenges; starks.cpp:84-88 calls them row by row)
 * once.  An operation is dst = a (cls) b; temporaries are numbered freely (dst_slot; an operand T1 / T3 names the latest
 * operation that wrote that slot); operand words v[] by kind:
 *   T1 T3: slot | NUM: value | CONST: column | CHAL PUB EVAL: index | POL POL3 DPOL: offset, row stride |
 *   CONSTS: column, row shift, modulus | POLS POL3S DPOLS: offset, row shift, modulus, row stride | X ZHINV XD XDW: none.
 * STOREQ: q[row] = a (T3) * ZhInv (b = ZHINV), dst_kind Q; STOREF: f[row] = a (T3), dst_kind Q; STOREP: params.pols[b] = a
 * (T1 / T3), b = DPOL / DPOLS, dst_kind = b's kind.  `step` says which domain and outputs (as for mi_chelpers_compile). */
#define MI_CHP_NONE 0
#define MI_CHP_T1 1
#define MI_CHP_T3 2
#define MI_CHP_POL 3
#define MI_CHP_POLS 4
#define MI_CHP_NUM 5
#define MI_CHP_CONST 6
#define MI_CHP_CONSTS 7
#define MI_CHP_CHAL 8
#define MI_CHP_PUB 9
#define MI_CHP_POL3 10
#define MI_CHP_POL3S 11
#define MI_CHP_X 12
#define MI_CHP_ZHINV 13
#define MI_CHP_Q 14
#define MI_CHP_EVAL 15
#define MI_CHP_XD 16
#define MI_CHP_XDW 17
#define MI_CHP_DPOL 18
#define MI_CHP_DPOLS 19
#define MI_CHP_ADD 0
#define MI_CHP_SUB 1
#define MI_CHP_MUL 2
#define MI_CHP_COPY 3
#define MI_CHP_STOREQ 4
#define MI_CHP_STOREF 5
#define MI_CHP_STOREP 6
typedef struct { uint32_t kind, reserved; uint64_t v[4]; } mi_chelpers_operand;
typedef struct { uint32_t cls, dst_kind; uint64_t dst_slot; mi_chelpers_operand a, b; } mi_chelpers_microop;
int mi_chelpers_compile_micro(mi_ctx *ctx, mi_chelpers_prog **out, int step, const mi_chelpers_microop *ops, uint64_t n_ops,
                              const mi_chelpers_section *sections, uint64_t n_sections, uint64_t n_const, uint64_t nrows_ext);
void mi_chelpers_free(mi_ctx *ctx, mi_chelpers_prog *prog);
/* out[0..8) = opcodes in, field operations decoded, after copy forwarding, after dead-code removal, live 64-bit words per
 * row as generated, after the reschedule, base temps, extension temps (host form); out[8..16) = device instructions per row,
 * base / extension temporaries kept in LDS, base temporaries spilled to HBM, temporary reads per row and how many of them
 * from the spill, LDS bytes per workgroup (without the constant tables), staged columns (0 when compiled without a ctx) */
int mi_chelpers_stats(const mi_chelpers_prog *prog, uint64_t out[16]);
/* Benchmarking knob: allocate at least `words` 64-bit LDS words of temporaries per row whatever the program needs (0 = off),
 * so that a synthetic program runs at the occupancy of a larger one (the zkEVM step42ns program needs 96: 3 workgroups
 * per CU).  Results are unaffected. */
int mi_set_chelpers_min_words(mi_ctx *ctx, uint64_t words);
/* Second backend: compile the translated program to gfx950 code (straight-line kernels generated from the program, built with
 * hiprtc; the reference compiles its generated chelpers into the prover at build time).  Needs no GPU.  cache_dir (or
 * $MI_CHELPERS_CACHE; NULL/unset = no cache) keeps the code objects, keyed by the hash of the generated source.  chunk_cost =
 * estimated VALU instructions per kernel (0 = 25 000).  Afterwards mi_chelpers_run_dev runs the compiled kernels instead of the
 * interpreter; results are the same field elements.  Requires shifts < 64 and power-of-two section row counts. */
int mi_chelpers_build_native(mi_chelpers_prog *prog, const char *cache_dir, uint64_t chunk_cost);
/* A section the caller keeps TILE-MAJOR in HBM -- [tile of 64 rows][column][row in tile], element (row, col) at
 * (row / 64 * ncols + col) * 64 + rev6(row % 64), rev6 = the 6-bit reversal (the rows that are multiples of 2^e are then the first
 * 64 >> e words of a column's run in a tile: what evmap reads of an extension), canonical values (mi_tile_major_dev,
 * mi_lde_merkle_host_keep_tiled / _host_tiled / _dev_tiled write it) -- instead of
 * row-major: the generated kernels read it in place, a lane per row and 512 contiguous bytes per operand and wave, and the per-batch
 * tile-major copy of that section (k_chp_transpose: a read and a write of the whole section per step) is not made.  For a section every
 * base-domain step reads but nothing writes or reads by stride: the witness cm1_n (host/starks.hpp; starks.cpp:66-210 read it three
 * times), and the extended sections cm1_2ns .. cm3_2ns as mi_lde_merkle_*_tiled leave them.  Call after mi_chelpers_compile and before
 * mi_chelpers_build_native / _precompile_shard / _lower_stats, once per tile-major section, named by its offset; the program then runs
 * through the compiled kernels only, over rows from a multiple of 64. */
int mi_chelpers_set_tiled_section(mi_chelpers_prog *prog, uint64_t section_offset);
/* ... and the constant polynomials: const_pols of mi_chelpers_run_dev is then [nrows / 64][n_const][64] (rows bit-reversed inside a tile
 * as above) -- a proving key's constants never change, so they can be kept that way for good.  Same call order; a program that reads no
 * constant polynomial is left as it is. */
int mi_chelpers_set_tiled_consts(mi_chelpers_prog *prog);
/* dst (tile-major as above, nrows x ncols_total) <- src (row-major, src_pitch words per row, ncols columns), placed at column col0 of
 * the tiles; nrows a multiple of 64; values canonicalised. */
int mi_tile_major_dev(mi_ctx *ctx, uint64_t *dst, uint64_t ncols_total, uint64_t col0, const uint64_t *src, uint64_t src_pitch,
                      uint64_t nrows, uint64_t ncols);
/* the way back, for checks: dst (row-major, dst_pitch words per row) <- rows [row0, row0 + nrows) x columns [col0, col0 + ncols) of the
 * tile-major section src_tiled (nrows_total x ncols_total) */
int mi_untile_dev(mi_ctx *ctx, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src_tiled, uint64_t ncols_total, uint64_t nrows_total,
                  uint64_t col0, uint64_t row0, uint64_t nrows, uint64_t ncols);
/* What the native backend makes of the program, without compiling anything: out = kernels, instructions evaluated as
 * Horner-chain accumulator steps, chain pieces, estimated VALU instructions per row, chain coefficients (+- C^e), per-piece
 * constants, folded (polynomial - evaluation) leaves, temporary words moved through the spill per row, polynomial elements
 * loaded per row (distinct per kernel, summed), distinct polynomial elements the generated kernels read, terms handed to the
 * linear kernel, sums it carries */
int mi_chelpers_lower_stats(const mi_chelpers_prog *prog, uint64_t chunk_cost, uint64_t out[12]);
/* Parallel builds: process `shard` of `nshards` compiles every nshards-th kernel into the cache (cache_dir required) and keeps
 * nothing; mi_chelpers_build_native afterwards finds every kernel there. */
int mi_chelpers_precompile_shard(mi_chelpers_prog *prog, const char *cache_dir, uint64_t chunk_cost, uint32_t shard,
                                 uint32_t nshards);
/* out = kernels, code-object bytes, build milliseconds, cache hits, estimated VALU instructions per row, temporary words moved
 * through the chunk-boundary spill per row, instructions evaluated as Horner-chain accumulator steps, words of the constants
 * table */
int mi_chelpers_native_stats(const mi_chelpers_prog *prog, uint64_t out[8]);
/* allocate the device buffers a native run over nrows rows needs now rather than inside the first run */
int mi_chelpers_reserve(mi_ctx *ctx, const mi_chelpers_prog *prog, uint64_t nrows);
/* rows whose tile-major operand copy is made at a time by the native backend (multiple of 64; 0 = about 8 GiB worth) */
int mi_set_chelpers_batch_rows(mi_ctx *ctx, uint64_t rows);
/* rows [row0, row0 + nrows) of the extended domain (the reference runs all NExtended rows: starks.cpp:240) */
int mi_chelpers_run_dev(mi_ctx *ctx, const mi_chelpers_prog *prog, const mi_chelpers_params *params, uint64_t row0,
                        uint64_t nrows);

/* ------------------------------------------------------------------ utilities */
/* Deterministic synthetic trace: out[i] = splitmix64(seed, i+1) reduced mod p. */
int mi_fill_synthetic_dev(mi_ctx *ctx, uint64_t *out, uint64_t count, uint64_t seed);
/* Column shard of the same synthetic matrix: out[r*out_pitch + c] = value of global element
 * (r, col0 + c) of a matrix with global_cols columns, so any column partition reproduces one trace. */
int mi_fill_synthetic_2d_dev(mi_ctx *ctx, uint64_t *out, uint64_t out_pitch, uint64_t nrows, uint64_t ncols,
                             uint64_t global_cols, uint64_t col0, uint64_t seed);
/* dst[r*dst_pitch + c] = canonical(src[r*src_pitch + c]): strided 2-D copy (column windows <-> row-major rows). */
int mi_copy_2d_dev(mi_ctx *ctx, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src, uint64_t src_pitch,
                   uint64_t nrows, uint64_t ncols);
void *mi_dev_alloc(mi_ctx *ctx, uint64_t bytes);
/* Sparse device memory (HIP virtual-memory management).  A row-sharded proof addresses FULL-HEIGHT sections on every device but a device
 * touches only its own rows of them (20 of 157 GB at zkEVM size): mi_vmm_reserve takes the address range (no memory), mi_vmm_back puts
 * physical memory under [offset, offset + bytes) (widened to 2 MiB boundaries; idempotent; fresh memory is not zeroed),
 * mi_vmm_allow_peer lets another device's kernels reach what is backed so far, mi_vmm_free unmaps and releases everything.  An access
 * outside a backed part faults. */
int mi_vmm_reserve(mi_ctx *ctx, uint64_t bytes, void **base);
int mi_vmm_back(mi_ctx *ctx, void *base, uint64_t offset, uint64_t bytes);
int mi_vmm_allow_peer(mi_ctx *ctx, void *base, int peer_device);
int mi_vmm_backed_bytes(mi_ctx *ctx, void *base, uint64_t *bytes);
int mi_vmm_free(mi_ctx *ctx, void *base);
int mi_dev_free(mi_ctx *ctx, void *p);
int mi_copy_h2d(mi_ctx *ctx, void *dst, const void *src, uint64_t bytes);
int mi_copy_d2h(mi_ctx *ctx, void *dst, const void *src, uint64_t bytes);
int mi_dev_zero(mi_ctx *ctx, void *p, uint64_t bytes); /* zeros, enqueued on the context's stream */
/* a column window: host rows of `width` words at row pitch src_pitch -> device rows at row pitch dst_pitch (words) */
int mi_copy_h2d_2d(mi_ctx *ctx, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src, uint64_t src_pitch, uint64_t width, uint64_t rows);
/* Selects the Poseidon code path: 2 (default) = full rounds with the MDS on 32-bit halves (v_mad_u64_u32) and the
 * 22 partial rounds in the grouped optimised form (dot products with 64-bit constants, tools/gen_poseidon_sparse.py);
 * 0 = naive rounds throughout, MDS on 32-bit halves; 1 = naive rounds, MDS on 22-bit limbs (v_mad_u32_u24).
 * All are bit-identical; exposed for benchmarking. */
int mi_set_poseidon_variant(mi_ctx *ctx, int variant);
/* Launches of at most max_states independent permutations (a single hash, the states of a small batch, the nodes of a small tree level,
 * the top of every tree) take the WAVE-COOPERATIVE form: one state across 12 lanes, ~10 us per dependent permutation instead of the
 * 58 us a lane needs for a whole state on its own.  Default 16384; 0 = always one state per lane.  Results are identical; exposed for
 * benchmarking and for testing both paths. */
int mi_set_poseidon_coop_max(mi_ctx *ctx, uint64_t max_states);
/* NTT tile width in elements per row segment: log_b = 4 (128-byte segments, 4 workgroups per CU) or 5
 * (256-byte segments, 2 per CU).  Results are identical; exposed for benchmarking. */
int mi_set_ntt_tile(mi_ctx *ctx, int log_b);
/* extendPol: 1 (default) = the last INTT pass and the first pass of the extended NTT run as one kernel whenever
 * their radices line up (the coefficients never reach HBM), 0 = always separate passes.  Results are identical;
 * exposed for benchmarking and for testing both paths. */
int mi_set_lde_fuse(mi_ctx *ctx, int fuse);
/* Leaf sponge memory access: 1 (default) = every lane fetches whole aligned 128-byte lines straight into a
 * per-lane ring in LDS (global_load_lds), 0 = plain per-block loads.  Results are identical; exposed for
 * benchmarking. */
int mi_set_leaf_mode(mi_ctx *ctx, int line_aligned);

/* Timing hooks used by bench.py: HIP events recorded on the context's stream. */
int mi_timer_start(mi_ctx *ctx, int slot);
int mi_timer_stop(mi_ctx *ctx, int slot);
int mi_timer_elapsed_ms(mi_ctx *ctx, int slot, float *ms); /* synchronises on the stop event */

/* ------------------------------------------------------------------ debug hooks (tests only)
 * Run the same inline arithmetic the kernels use on the HOST CPU so that it can be checked on machines
 * without a GPU.  Not used by any entry point above. */
void mi_dbg_host_poseidon_permute(uint64_t st[12], int variant);
uint64_t mi_dbg_host_mul(uint64_t a, uint64_t b);
void mi_dbg_host_e3_mul(uint64_t out[3], const uint64_t a[3], const uint64_t b[3]);
void mi_dbg_host_e3_inv(uint64_t out[3], const uint64_t a[3]);
void mi_dbg_host_dft16(uint64_t x[16], int log_size, int inverse);
/* The DEVICE build of the field arithmetic (it differs from the host build: inline asm, wave-uniform branches), element
 * by element over device arrays a, b of n entries (any u64 encodings): out[0..n) = a*b, out[n..2n) = a+b,
 * out[2n..3n) = a-b, out[3n..4n) = -a (as a * 2^96 through the 32-bit-shift twiddle form), out[4n..5n) = a * 2^40
 * (the wide-shift twiddle form); all canonical. */
int mi_dbg_field_ops_dev(mi_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n);
/* The translated constraint-evaluator program run on the HOST CPU over host pointers (every pointer of `params` is a
 * host pointer here) for the listed rows: same translator output, same instruction semantics as the kernel. */
/* tests only: the program as LOWERED for the native backend (Horner chains, chain pieces per kernel, spill lists), on the CPU */
int mi_dbg_host_chelpers_run_lowered(const mi_chelpers_prog *prog, const mi_chelpers_params *params, const uint64_t *rows,
                                     uint64_t nrows, uint64_t chunk_cost);
int mi_dbg_host_chelpers_run(const mi_chelpers_prog *prog, const mi_chelpers_params *params, const uint64_t *rows,
                             uint64_t nrows);
/* Verification hook: out[r] = (accumulate ? out[r] : 0) + sum_c coef[c] * src[r*pitch + c] mod p for r < nrows (device
 * pointers; coef: ncols canonical values; accumulate lets a matrix stored as several column windows be summed).  The LDE is linear, so the full-size checks compare the oracle's extension of this one column
 * of the trace with the same combination of the extended trace: a checksum over every column at every row. */
/* EXPERIMENT (round 5): the transform of ncols contiguous COLUMNS (column-major: column c at src + c n) with the passes after the first
 * as radix-256 passes over tiles of 32 consecutive rows of one column, a workgroup walking over the columns (csrc/ntt.hip k_ntt_pass_cm);
 * n = 2^16 or 2^24; *ms_cm = the time of those passes.  tools/ntt_colmajor_probe.py; not used by the product's transforms. */
int mi_dbg_ntt_colmajor_dev(mi_ctx *ctx, uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t ncols, int inverse, float *ms_cm);
int mi_dbg_lincomb_cols_dev(mi_ctx *ctx, uint64_t *out, const uint64_t *src, uint64_t pitch, uint64_t nrows,
                            uint64_t ncols, const uint64_t *coef, int accumulate);

#ifdef __cplusplus
}
#endif
#endif
