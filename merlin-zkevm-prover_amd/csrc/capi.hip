// capi.hip -- extern "C" entry points of libmi_stark.so (see include/mi_stark.h for the contract and the
// reference interfaces each one replaces).  No CPU fallback exists: without a HIP device every compute
// call returns MI_ERR_NO_DEVICE.
#include "common.h"
#include <thread>
#include <emmintrin.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>

static thread_local char g_err[512] = "";

void mi_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *mi_last_error(void) { return g_err; }
extern "C" const char *mi_version(void) { return "mi_stark 0.1 (gfx950)"; }

extern "C" int mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


// ------------------------------------------------------------------ MI_MULTI_CHECK: logical ownership of device address ranges
#include <atomic>
namespace {
struct OwnRange { uintptr_t hi; int shard; std::string what; };
std::mutex g_own_mu;
std::map<uintptr_t, OwnRange> g_own; // by start address
std::atomic<uint64_t> g_own_checks{0}, g_own_unknown{0}, g_own_violations{0};
int g_check_state = -1; // -1 not read yet, 0 off, 1 on
int g_own_lead[64]; // device groups (csrc/multi.hip): shards of one group share buffers legitimately; identity until mi_own_set_leaders
bool g_own_lead_set = false;
}
void mi_own_set_leaders(const uint32_t *lead, uint32_t n)
{
    std::lock_guard<std::mutex> lock(g_own_mu);
    for (uint32_t i = 0; i < 64; i++) g_own_lead[i] = i < n ? (int)lead[i] : (int)i;
    g_own_lead_set = true;
}
bool mi_check_on()
{
    if (g_check_state < 0) {
        const char *e = getenv("MI_MULTI_CHECK");
        g_check_state = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_check_state == 1;
}
void mi_own_add(const void *p, uint64_t bytes, int shard, const char *what)
{
    if (!mi_check_on() || !p || !bytes) return;
    std::lock_guard<std::mutex> lock(g_own_mu);
    g_own[(uintptr_t)p] = OwnRange{(uintptr_t)p + bytes, shard, what ? what : ""};
}
void mi_own_del(const void *p)
{
    if (!mi_check_on() || !p) return;
    std::lock_guard<std::mutex> lock(g_own_mu);
    g_own.erase((uintptr_t)p);
}
int mi_own_check(int shard, const void *p, const char *what)
{
    if (!p) return MI_OK;
    g_own_checks.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lock(g_own_mu);
    auto it = g_own.upper_bound((uintptr_t)p);
    if (it == g_own.begin()) { g_own_unknown.fetch_add(1, std::memory_order_relaxed); return MI_OK; }
    --it;
    if ((uintptr_t)p >= it->second.hi) { g_own_unknown.fetch_add(1, std::memory_order_relaxed); return MI_OK; }
    if (it->second.shard == shard) return MI_OK;
    if (g_own_lead_set && shard >= 0 && shard < 64 && it->second.shard >= 0 && it->second.shard < 64 && g_own_lead[shard] == g_own_lead[it->second.shard]) return MI_OK; // one device group
    g_own_violations.fetch_add(1, std::memory_order_relaxed);
    mi_set_error("MI_MULTI_CHECK: %s was handed %p, which lies in \"%s\" of logical shard %d, while working for shard %d (on a real multi-GPU node: memory of another device)",
                 what, p, it->second.what.c_str(), it->second.shard, shard);
    fprintf(stderr, "mi_stark: %s\n", mi_last_error());
    return MI_ERR_INVALID;
}
// [enabled, checks made, pointers nobody entered (passed), violations]
extern "C" int mi_multi_check_stats(uint64_t out[4])
{
    if (!out) return MI_ERR_INVALID;
    out[0] = mi_check_on() ? 1 : 0;
    out[1] = g_own_checks.load(); out[2] = g_own_unknown.load(); out[3] = g_own_violations.load();
    return MI_OK;
}
// a caller that allocates device memory itself for a shard (host/starks.hpp's arenas come through mi_dev_alloc / mi_vmm_reserve and need
// not call this) enters the range; bytes == 0 withdraws it
extern "C" int mi_multi_own(const void *p, uint64_t bytes, int shard, const char *what)
{
    if (!p) return MI_ERR_INVALID;
    if (bytes) mi_own_add(p, bytes, shard, what); else mi_own_del(p);
    return MI_OK;
}

// every entry point: reject a null context, take the context lock, make its device current
#define CTX_OK(ctx)                                                   \
    if (!(ctx)) {                                                     \
        mi_set_error("%s: null context", __func__);                   \
        return MI_ERR_INVALID;                                        \
    }                                                                 \
    std::lock_guard<std::recursive_mutex> ctx_lock_((ctx)->mu);       \
    MI_HIP_CHECK(hipSetDevice((ctx)->device))

int mi_scratch(mi_ctx *c, uint64_t bytes, void **p)
{
    if (c->pool->scratch_bytes < bytes) {
        MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // earlier users are done with the old buffer
        if (c->pool->scratch) MI_HIP_CHECK(hipFree(c->pool->scratch));
        c->pool->scratch = nullptr;
        c->pool->scratch_bytes = 0;
        const uint64_t want = (bytes + (1ull << 20) - 1) & ~((1ull << 20) - 1);
        hipError_t e = hipMalloc((void **)&c->pool->scratch, want);
        if (e != hipSuccess) {
            mi_set_error("cannot allocate %llu bytes of device scratch: %s", (unsigned long long)want, hipGetErrorString(e));
            return MI_ERR_NOMEM;
        }
        c->pool->scratch_bytes = want;
    }
    *p = c->pool->scratch;
    return MI_OK;
}

extern "C" void mi_ctx_destroy(mi_ctx *c);

static int ctx_init(mi_ctx *c, int device)
{
    if (device >= 0) MI_HIP_CHECK(hipSetDevice(device));
    MI_HIP_CHECK(hipGetDevice(&c->device));
    hipDeviceProp_t prop;
    MI_HIP_CHECK(hipGetDeviceProperties(&prop, c->device));
    c->cu_count = prop.multiProcessorCount;
    c->stream = nullptr; // the device's default stream until the caller hands one over
    c->own_stream = false;
    MI_HIP_CHECK(hipMalloc((void **)&c->small, 4096));
    MI_HIP_CHECK(hipHostMalloc((void **)&c->pinned, 4096, hipHostMallocDefault));
    for (int i = 0; i < mi_ctx::N_TIMERS; i++) {
        MI_HIP_CHECK(hipEventCreate(&c->ev_start[i]));
        MI_HIP_CHECK(hipEventCreate(&c->ev_stop[i]));
    }
    return MI_OK;
}

extern "C" int mi_ctx_create(mi_ctx **out, int device)
{
    if (!out) return MI_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        mi_set_error("no HIP device available: libmi_stark has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    if (device >= n) {
        mi_set_error("device %d out of range (%d devices)", device, n);
        return MI_ERR_INVALID;
    }
    mi_ctx *c = new mi_ctx();
    const int st = ctx_init(c, device);
    if (st != MI_OK) { // nothing half-built survives a failed create
        mi_ctx_destroy(c);
        return st;
    }
    *out = c;
    return MI_OK;
}

extern "C" int mi_ctx_device(const mi_ctx *c) { return c ? c->device : -1; }

extern "C" void mi_ctx_destroy(mi_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (void *p : c->owned) (void)hipFree(p);
    if (c->workspace_lent) { if (c->own_workspace) (void)hipFree(c->own_workspace); } // the lent buffer is the caller's
    else if (c->workspace) (void)hipFree(c->workspace);
    if (c->w256) (void)hipFree(c->w256);
    if (c->small) (void)hipFree(c->small);
    if (c->tr_stream) { (void)hipStreamSynchronize(c->tr_stream); (void)hipStreamDestroy(c->tr_stream); }
    if (c->tr_dev) (void)hipFree(c->tr_dev);
    if (c->own_pool.scratch) (void)hipFree(c->own_pool.scratch);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->chelpers_scratch) (void)hipFree(c->chelpers_scratch);
    if (c->own_pool.chelpers_stage) (void)hipFree(c->own_pool.chelpers_stage);
    if (c->own_pool.chelpers_cst) (void)hipFree(c->own_pool.chelpers_cst);
    if (c->own_pool.chelpers_tiled) (void)hipFree(c->own_pool.chelpers_tiled);
    if (c->own_pool.chelpers_spill) (void)hipFree(c->own_pool.chelpers_spill);
    if (c->own_pool.chelpers_lin) (void)hipFree(c->own_pool.chelpers_lin);
    if (c->stage) (void)hipFree(c->stage);
    for (int i = 0; i < 3; i++) {
        if (c->pack_stage[i]) (void)hipHostFree(c->pack_stage[i]);
        if (c->ev_pack_sent[i]) (void)hipEventDestroy(c->ev_pack_sent[i]);
    }
    for (int s = 0; s < 2; s++)
        if (c->copy_stream[s]) (void)hipStreamSynchronize(c->copy_stream[s]);
    for (int i = 0; i < mi_ctx::N_STAGE; i++) {
        for (int s = 0; s < 2; s++)
            if (c->ev_uploaded[i][s]) (void)hipEventDestroy(c->ev_uploaded[i][s]);
        if (c->ev_consumed[i]) (void)hipEventDestroy(c->ev_consumed[i]);
    }
    for (int s = 0; s < 2; s++)
        if (c->copy_stream[s]) (void)hipStreamDestroy(c->copy_stream[s]);
    for (int i = 0; i < mi_ctx::N_TIMERS; i++) {
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mi_ctx_set_stream(mi_ctx *c, void *s)
{
    CTX_OK(c);
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    // NULL is a valid handle: it names the device's default stream (what torch.cuda.current_stream()
    // returns as 0 unless a side stream is active), which orders with every other stream's work.
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return MI_OK;
}

extern "C" int mi_ctx_sync(mi_ctx *c)
{
    CTX_OK(c);
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

extern "C" int mi_ctx_set_workspace_limit(mi_ctx *c, uint64_t bytes)
{
    CTX_OK(c);
    MI_REQUIRE(bytes >= (1ull << 20), "workspace limit must be at least 1 MiB");
    c->workspace_limit = bytes;
    return MI_OK;
}

// The reference's `buf` argument of extendPol / NTT (starks.cpp:52 lends p_cm2_2ns, :133,214 pBuffer): scratch the CALLER owns.
// While lent, transforms size their column chunks to it and allocate nothing.
extern "C" int mi_ctx_lend_workspace(mi_ctx *c, void *ptr, uint64_t bytes)
{
    CTX_OK(c);
    MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // queued transforms still use the current scratch
    if (c->workspace_lent) {
        c->workspace = c->own_workspace;
        c->workspace_bytes = c->own_workspace_bytes;
        c->workspace_limit = c->own_workspace_limit;
        c->own_workspace = nullptr;
        c->workspace_lent = false;
    }
    if (!ptr) return MI_OK;
    MI_REQUIRE(bytes >= (1ull << 20) && ((uintptr_t)ptr & 15) == 0, "a lent workspace must be 16-byte aligned and at least 1 MiB");
    c->own_workspace = c->workspace;
    c->own_workspace_bytes = c->workspace_bytes;
    c->own_workspace_limit = c->workspace_limit;
    c->workspace = (u64 *)ptr;
    c->workspace_bytes = c->workspace_limit = bytes;
    c->workspace_lent = true;
    return MI_OK;
}

extern "C" int mi_dev_mem_info(mi_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    CTX_OK(c);
    size_t f = 0, t = 0;
    MI_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MI_OK;
}

int mi_ensure_workspace(mi_ctx *c, uint64_t bytes)
{
    if (bytes <= c->workspace_bytes) return MI_OK;
    if (c->workspace_lent) {
        mi_set_error("the lent NTT workspace (%llu bytes) is too small: one column chunk of this transform needs %llu",
                     (unsigned long long)c->workspace_bytes, (unsigned long long)bytes);
        return MI_ERR_NOMEM;
    }
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->workspace) MI_HIP_CHECK(hipFree(c->workspace));
    c->workspace = nullptr;
    c->workspace_bytes = 0;
    hipError_t e = hipMalloc((void **)&c->workspace, bytes);
    if (e != hipSuccess) {
        mi_set_error("cannot allocate %llu bytes of NTT workspace: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return MI_ERR_NOMEM;
    }
    c->workspace_bytes = bytes;
    return MI_OK;
}

extern "C" int mi_set_poseidon_coop_max(mi_ctx *c, uint64_t max_states)
{
    CTX_OK(c);
    c->poseidon_coop_max = max_states;
    return MI_OK;
}

extern "C" int mi_set_leaf_mode(mi_ctx *c, int line_aligned)
{
    CTX_OK(c);
    c->leaf_line_aligned = line_aligned != 0;
    return MI_OK;
}

extern "C" int mi_set_lde_fuse(mi_ctx *c, int fuse)
{
    CTX_OK(c);
    c->lde_fuse_mid = fuse != 0;
    return MI_OK;
}

extern "C" int mi_set_ntt_tile(mi_ctx *c, int log_b)
{
    CTX_OK(c);
    MI_REQUIRE(log_b == 4 || log_b == 5, "log_b must be 4 or 5");
    c->ntt_log_b = (uint32_t)log_b;
    return MI_OK;
}

extern "C" int mi_set_poseidon_variant(mi_ctx *c, int v)
{
    CTX_OK(c);
    MI_REQUIRE(v >= 0 && v <= 2, "variant must be 0, 1 or 2");
    c->poseidon_variant = v;
    return MI_OK;
}

// ------------------------------------------------------------------ NTT / LDE
extern "C" int mi_ntt_dev(mi_ctx *c, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src, uint64_t src_pitch, uint64_t n,
                          uint64_t ncols, int inverse)
{
    CTX_OK(c);
    MI_OWN(c, dst);
    MI_OWN(c, src);
    MI_REQUIRE((dst && src) || n == 0 || ncols == 0, "null buffer");
    return launch_ntt(c, (u64 *)dst, dst_pitch, (const u64 *)src, src_pitch, n, ncols, inverse);
}

extern "C" int mi_lde_dev(mi_ctx *c, uint64_t *out, uint64_t out_pitch, const uint64_t *in, uint64_t in_pitch, uint64_t n_ext,
                          uint64_t n, uint64_t ncols)
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_OWN(c, in);
    MI_REQUIRE((out && in) || n == 0 || ncols == 0, "null buffer");
    return launch_lde(c, (u64 *)out, out_pitch, (const u64 *)in, in_pitch, n_ext, n, ncols);
}

// host-pointer staging helper: device buffers owned for the duration of one call
struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(uint64_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) {
            mi_set_error("hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
            return MI_ERR_NOMEM;
        }
        return MI_OK;
    }
};

extern "C" int mi_ntt(mi_ctx *c, uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t ncols, int inverse)
{
    CTX_OK(c);
    if (n == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(dst && src, "null buffer");
    const uint64_t bytes = n * ncols * 8;
    DevBuf d;
    MI_TRY(d.alloc(bytes));
    MI_HIP_CHECK(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    MI_TRY(launch_ntt(c, (u64 *)d.p, ncols, (const u64 *)d.p, ncols, n, ncols, inverse));
    MI_HIP_CHECK(hipMemcpyAsync(dst, d.p, bytes, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

extern "C" int mi_lde(mi_ctx *c, uint64_t *out, const uint64_t *in, uint64_t n_ext, uint64_t n, uint64_t ncols)
{
    CTX_OK(c);
    if (n == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(out && in, "null buffer");
    DevBuf di, dout;
    MI_TRY(di.alloc(n * ncols * 8));
    MI_TRY(dout.alloc(n_ext * ncols * 8));
    MI_HIP_CHECK(hipMemcpyAsync(di.p, in, n * ncols * 8, hipMemcpyHostToDevice, c->stream));
    MI_TRY(launch_lde(c, (u64 *)dout.p, ncols, (const u64 *)di.p, ncols, n_ext, n, ncols));
    MI_HIP_CHECK(hipMemcpyAsync(out, dout.p, n_ext * ncols * 8, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

// ------------------------------------------------------------------ Poseidon / Merkle
extern "C" int mi_poseidon_permute_dev(mi_ctx *c, uint64_t *out, const uint64_t *in, uint64_t count)
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_OWN(c, in);
    MI_REQUIRE((out && in) || count == 0, "null buffer");
    return launch_permute(c, (u64 *)out, (const u64 *)in, count);
}

extern "C" int mi_poseidon_hash_full_result(mi_ctx *c, uint64_t out[12], const uint64_t in[12])
{
    CTX_OK(c);
    MI_REQUIRE(out && in, "null buffer");
    memcpy(c->pinned, in, 96);
    MI_HIP_CHECK(hipMemcpyAsync(c->small, c->pinned, 96, hipMemcpyHostToDevice, c->stream));
    MI_TRY(launch_permute(c, c->small + 16, c->small, 1));
    MI_HIP_CHECK(hipMemcpyAsync(c->pinned + 16, c->small + 16, 96, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(out, c->pinned + 16, 96);
    return MI_OK;
}

// Transcript::put (transcript.cpp:4-29) over the class's members, all HOST pointers: n elements absorbed in one launch (the per-hash
// form costs a launch, two copies and a synchronisation for every 8 elements: a put of the 118 evaluations of a recursive STARK is 44
// of them).  Puts that complete no block of 8 stay on the host.
extern "C" int mi_transcript_put(mi_ctx *c, uint64_t state[4], uint64_t pending[8], uint64_t out[12], uint32_t *pending_cursor,
                                 uint32_t *out_cursor, const uint64_t *input, uint64_t n)
{
    CTX_OK(c);
    MI_REQUIRE(state && pending && out && pending_cursor && out_cursor && (input || n == 0), "null buffer");
    MI_REQUIRE(*pending_cursor < 8, "pending cursor out of range");
    if (n == 0) return MI_OK;
    if (*pending_cursor + n < 8) { // nothing to hash: the elements wait in `pending`
        for (uint64_t i = 0; i < n; i++) pending[(*pending_cursor)++] = input[i];
        *out_cursor = 0;
        return MI_OK;
    }
    MI_REQUIRE(n <= (1ull << 20), "transcript input too long for one call");
    // no allocation on this path (a proof makes dozens of these calls): the context's 4 KiB words when they suffice, its scratch otherwise
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    const bool fits = (32 + n) * 8 <= 4096;
    if (!c->tr_stream) MI_HIP_CHECK(hipStreamCreateWithFlags(&c->tr_stream, hipStreamNonBlocking));
    if (c->tr_dev_bytes < (32 + n) * 8) { // (its own words: the context's scratch may be in use by kernels queued on the main stream)
        MI_HIP_CHECK(hipStreamSynchronize(c->tr_stream));
        if (c->tr_dev) MI_HIP_CHECK(hipFree(c->tr_dev));
        c->tr_dev = nullptr;
        c->tr_dev_bytes = 0;
        const uint64_t want = std::max<uint64_t>(1 << 16, (32 + n) * 8);
        MI_HIP_CHECK(hipMalloc((void **)&c->tr_dev, want));
        c->tr_dev_bytes = want;
    }
    u64 *dev = c->tr_dev;
    std::vector<uint64_t> big(fits ? 0 : 32 + n);
    uint64_t *h = fits ? (uint64_t *)c->pinned : big.data();
    memcpy(&h[0], state, 32); memcpy(&h[4], pending, 64); memcpy(&h[12], out, 96);
    h[24] = *pending_cursor; h[25] = *out_cursor;
    memcpy(&h[32], input, n * 8);
    MI_HIP_CHECK(hipMemcpyAsync(dev, h, (32 + n) * 8, hipMemcpyHostToDevice, c->tr_stream));
    MI_TRY(launch_transcript_put(c, dev, (const u64 *)dev + 32, n, c->tr_stream));
    MI_HIP_CHECK(hipMemcpyAsync(h, dev, 26 * 8, hipMemcpyDeviceToHost, c->tr_stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->tr_stream));
    memcpy(state, &h[0], 32); memcpy(pending, &h[4], 64); memcpy(out, &h[12], 96);
    *pending_cursor = (uint32_t)h[24]; *out_cursor = (uint32_t)h[25];
    return MI_OK;
}

extern "C" int mi_poseidon_hash(mi_ctx *c, uint64_t out[4], const uint64_t in[12])
{
    uint64_t full[12];
    MI_TRY(mi_poseidon_hash_full_result(c, full, in));
    memcpy(out, full, 32);
    return MI_OK;
}

extern "C" int mi_linear_hash_rows_dev(mi_ctx *c, uint64_t *digests, const uint64_t *src, uint64_t pitch, uint64_t ncols,
                                       uint64_t nrows)
{
    CTX_OK(c);
    MI_OWN(c, digests);
    MI_OWN(c, src);
    MI_REQUIRE((digests && (src || ncols == 0)) || nrows == 0, "null buffer");
    MI_REQUIRE(pitch >= ncols, "pitch smaller than ncols");
    return launch_linear_hash_rows(c, (u64 *)digests, (const u64 *)src, pitch, ncols, nrows);
}

extern "C" int mi_linear_hash_absorb_dev(mi_ctx *c, uint64_t *digests, uint32_t nwindows, const uint64_t *const *bases,
                                         const uint64_t *pitches, const uint64_t *widths, uint64_t nrows, int first, int final)
{
    CTX_OK(c);
    MI_REQUIRE(nrows == 0 || nwindows == 0 || (digests && bases && pitches && widths), "null buffer");
    for (uint32_t i = 0; i < nwindows && nrows; i++) MI_REQUIRE(bases[i] || widths[i] == 0, "null column window");
    MI_OWN(c, digests);
    for (uint32_t i = 0; i < nwindows && nrows; i++) MI_OWN(c, bases[i]);
    return launch_linear_hash_absorb(c, (u64 *)digests, nwindows, (const u64 *const *)bases, pitches, widths, nrows, first != 0,
                                     final != 0);
}

extern "C" int mi_poseidon_linear_hash(mi_ctx *c, uint64_t out[4], const uint64_t *in, uint64_t size)
{
    CTX_OK(c);
    MI_REQUIRE(out && (in || size == 0), "null buffer");
    DevBuf d;
    MI_TRY(d.alloc(size * 8 + 32));
    if (size) MI_HIP_CHECK(hipMemcpyAsync((char *)d.p + 32, in, size * 8, hipMemcpyHostToDevice, c->stream));
    MI_TRY(launch_linear_hash_rows(c, (u64 *)d.p, (const u64 *)d.p + 4, size, size, 1));
    MI_HIP_CHECK(hipMemcpyAsync(c->pinned, d.p, 32, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(out, c->pinned, 32);
    return MI_OK;
}

extern "C" int mi_merkle_levels_dev(mi_ctx *c, uint64_t *nodes, uint64_t nleaves)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    if (nleaves == 0) return MI_OK;
    MI_REQUIRE(nodes, "null buffer");
    return launch_merkle_levels(c, (u64 *)nodes, nleaves);
}

extern "C" int mi_merkle_build_dev(mi_ctx *c, uint64_t *nodes, const uint64_t *src, uint64_t pitch, uint64_t ncols,
                                   uint64_t nrows)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    MI_OWN(c, src);
    if (nrows == 0) return MI_OK; // merkletree() returns immediately on zero rows
    MI_REQUIRE(nodes && (src || ncols == 0), "null buffer");
    MI_REQUIRE(is_pow2(nrows), "number of rows must be a power of two");
    MI_REQUIRE(pitch >= ncols, "pitch smaller than ncols");
    if (ncols == 0) return launch_merkle_zero_width(c, (u64 *)nodes, nrows);
    MI_TRY(launch_linear_hash_rows(c, (u64 *)nodes, (const u64 *)src, pitch, ncols, nrows));
    return launch_merkle_levels(c, (u64 *)nodes, nrows);
}

extern "C" int mi_merkle_build(mi_ctx *c, uint64_t *nodes, const uint64_t *src, uint64_t ncols, uint64_t nrows)
{
    CTX_OK(c);
    if (nrows == 0) return MI_OK;
    MI_REQUIRE(nodes && (src || ncols == 0), "null buffer");
    DevBuf ds, dn;
    MI_TRY(ds.alloc(nrows * ncols * 8));
    MI_TRY(dn.alloc(mi_merkle_num_nodes_elems(nrows) * 8));
    MI_HIP_CHECK(hipMemcpyAsync(ds.p, src, nrows * ncols * 8, hipMemcpyHostToDevice, c->stream));
    MI_TRY(mi_merkle_build_dev(c, (uint64_t *)dn.p, (const uint64_t *)ds.p, ncols, ncols, nrows));
    MI_HIP_CHECK(hipMemcpyAsync(nodes, dn.p, mi_merkle_num_nodes_elems(nrows) * 8, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

static int group_proofs_impl(mi_ctx *c, uint64_t *proofs, const uint64_t *nodes, const uint64_t *src, uint64_t pitch, uint64_t height, uint64_t width,
                             const uint64_t *idx, uint64_t nq, bool tiled);
extern "C" int mi_merkle_group_proofs_dev(mi_ctx *c, uint64_t *proofs, const uint64_t *nodes, const uint64_t *src,
                                          uint64_t pitch, uint64_t height, uint64_t width, const uint64_t *idx, uint64_t nq)
{
    return group_proofs_impl(c, proofs, nodes, src, pitch, height, width, idx, nq, false);
}
extern "C" int mi_merkle_group_proofs_tiled_dev(mi_ctx *c, uint64_t *proofs, const uint64_t *nodes, const uint64_t *src_tiled,
                                                uint64_t ncols_total, uint64_t height, uint64_t width, const uint64_t *idx, uint64_t nq)
{
    if (c && width > ncols_total) { mi_set_error("mi_merkle_group_proofs_tiled_dev: more values asked for than the section has columns"); return MI_ERR_INVALID; }
    return group_proofs_impl(c, proofs, nodes, src_tiled, ncols_total, height, width, idx, nq, true);
}
static int group_proofs_impl(mi_ctx *c, uint64_t *proofs, const uint64_t *nodes, const uint64_t *src, uint64_t pitch, uint64_t height, uint64_t width,
                             const uint64_t *idx, uint64_t nq, bool tiled)
{
    CTX_OK(c);
    MI_OWN(c, proofs);
    MI_OWN(c, nodes);
    MI_OWN(c, src);
    if (nq == 0) return MI_OK;
    MI_REQUIRE(proofs && (nodes || src) && (src || width == 0) && idx, "null buffer"); // width 0: sibling paths only; nodes NULL: row values only
    for (uint64_t q = 0; q < nq; q++) MI_REQUIRE(idx[q] < height, "query index out of range");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    u64 *di = c->small; // the indices: the context's 4 KiB words (512 queries) or its scratch, no allocation per call
    if (nq * 8 > 4096) MI_TRY(mi_scratch(c, nq * 8, (void **)&di));
    MI_HIP_CHECK(hipMemcpyAsync(di, idx, nq * 8, hipMemcpyHostToDevice, c->stream));
    MI_TRY(launch_group_proofs(c, (u64 *)proofs, (const u64 *)nodes, (const u64 *)src, pitch, height, width, (const u64 *)di, nq, tiled));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // idx is the caller's pageable memory: it may go once this returns
    return MI_OK;
}

// ------------------------------------------------------------------ stage driver: host trace in, resident extension + tree out
// Starks::genProof step 1 (starks.cpp:48-59) with the callers unchanged hands over a HOST trace (p_cm1_n, 44.6 GB) and wants
// the extended trace and its tree.  Done call by call (extendPol, then merkelize) that is 44.6 GB up, 89 GB down, 89 GB up
// again.  Here the trace is streamed up in COLUMN CHUNKS on a copy stream while the GPU works on the chunks that have
// arrived: chunk k+1 uploads (packed by host threads into page-locked staging and sent as one contiguous copy -- or, with
// mi_set_host_pack_threads(ctx, 0), as a strided 2-D copy out of the row-major host matrix, which the DMA engines move at
// 39-53 GB/s instead of 57) while
// chunk k is extended into its columns of the resident extended trace and chunk k-1's columns are absorbed into the
// running leaf sponges (the streaming form of linear_hash: the capacity is carried in the digest buffer).  The PCIe
// time (0.78 s at 57 GB/s) and the kernel time (0.77 s) overlap instead of adding; the extended trace and the nodes stay
// in HBM for the later steps and only the root is read back by the caller.
static int effective_pack_threads(const mi_ctx *c)
{
    const unsigned hw = std::thread::hardware_concurrency();
    return c->pack_threads >= 0 ? c->pack_threads : (hw >= 8 ? (int)std::min(16u, hw) : 0);
}

extern "C" int mi_get_host_pack_threads(mi_ctx *c)
{
    if (!c) return -1;
    return effective_pack_threads(c);
}

// base_pitch == 0: the base-domain section is kept tile-major (mi_lde_merkle_host_keep_tiled); ext_tiled: so is the extension -- a
// chunk is extended into a compact buffer and the leaf kernel, absorbing it, writes its words into the section (launch_linear_hash_absorb_emit)
static int lde_merkle_host_impl(mi_ctx *c, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, uint64_t *base, uint64_t base_pitch,
                                const uint64_t *trace_host, uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols, bool ext_tiled = false)
{
    if (n == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(nodes && ext && trace_host, "null buffer");
    MI_REQUIRE(!ext_tiled || (n_ext % 64 == 0 && ncols > 4 && ext_pitch == ncols), "a tile-major extension has a multiple of 64 rows, more than 4 columns and no pitch of its own");
    MI_REQUIRE(!base || base_pitch == 0 || base_pitch >= ncols, "pitch smaller than ncols");
    MI_REQUIRE(!base || base_pitch != 0 || n % 64 == 0, "a tile-major section has a multiple of 64 rows");
    MI_REQUIRE(is_pow2(n) && is_pow2(n_ext) && n_ext >= n, "sizes must be powers of two with n_ext >= n");
    MI_REQUIRE(ext_pitch >= ncols, "pitch smaller than ncols");
    if (chunk_cols == 0) chunk_cols = 64;
    MI_REQUIRE(chunk_cols % 8 == 0, "chunk width must be a multiple of 8 (the sponge absorbs whole blocks per chunk)");
    // Chunk schedule.  A chunk is a 2-D copy whose rows are (8 x width) bytes at the trace's row pitch, and the DMA engines
    // move short rows slower (measured on MI355X, profiles/r02_pcie_chunk_sweep.json: 39.5 GB/s at 32 columns, 49 at 64,
    // 52.7 at 128, 55 for whole rows) -- but nothing can run before the first chunk is up, and a chunk's kernels (0.91 x its
    // upload time) only hide behind the NEXT upload, so wide chunks leave a long tail after the last one.  Measured optimum
    // for the 665-column trace: a 32-column first chunk, then 64 (1.00 s per step against 0.91 s for the bare upload).
    // With pack_threads > 0 (mi_set_host_pack_threads) a chunk is first packed by host threads into page-locked staging and sent
    // as ONE contiguous copy at the full rate whatever its width, so the chunks can be narrow (a short head before the first kernel
    // and a short tail after the last upload): 32 columns each, the remainder split so that the last chunk is the smallest.
    // default: up to 16 packing threads; a host with fewer than 8 hardware threads packs slower than the strided 2-D copies run
    // (measured: 6 threads 1.10 s, 8 threads 0.91 s, 2-D copies 1.00 s per zkEVM step)
    const int pack_threads = effective_pack_threads(c);
    const bool packed = pack_threads > 0;
    std::vector<uint64_t> c0s, cws;
    if (packed) {
        // nothing runs before the first chunk is up and only the last chunk's kernels run after the last upload: a narrow first
        // chunk (8 columns, then 24), 32-column chunks, and a tapering end (... 32, 16, the rest)
        uint64_t pw = std::min<uint64_t>(32, chunk_cols);
        if (const char *e = getenv("MI_PACK_COLS")) { // experiments; anything that is not a multiple of 8 >= 8 is ignored
            const uint64_t v = (uint64_t)std::max(0, atoi(e)) & ~7ull;
            if (v >= 8) pw = std::min<uint64_t>(v, chunk_cols);
        }
        const bool taper = pw >= 32 && ncols > 3 * pw && !getenv("MI_PACK_NO_TAPER");
        for (uint64_t c0 = 0, k = 0; c0 < ncols; k++) {
            const uint64_t rem = ncols - c0;
            uint64_t w = std::min(pw, rem);
            if (taper) {
                if (k == 0) w = 8;
                else if (k == 1) w = 24;
                else if (rem <= 32 && rem > 16) w = 16;
            }
            c0s.push_back(c0);
            cws.push_back(w);
            c0 += w;
        }
    } else {
        for (uint64_t c0 = 0, k = 0; c0 < ncols; k++) {
            uint64_t w = k == 0 ? 32 : k == 1 ? 96 : chunk_cols;
            w = std::min(std::min(w, chunk_cols), ncols - c0);
            c0s.push_back(c0);
            cws.push_back(w);
            c0 += w;
        }
    }
    const uint64_t n_chunks = c0s.size();
    if (!c->copy_stream[0]) {
        for (int s = 0; s < 2; s++) MI_HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream[s], hipStreamNonBlocking));
        for (int i = 0; i < mi_ctx::N_STAGE; i++) {
            for (int s = 0; s < 2; s++) MI_HIP_CHECK(hipEventCreateWithFlags(&c->ev_uploaded[i][s], hipEventDisableTiming));
            MI_HIP_CHECK(hipEventCreateWithFlags(&c->ev_consumed[i], hipEventDisableTiming));
        }
    }
    // compact staging buffers [n x chunk]
    constexpr int NS = mi_ctx::N_STAGE;
    const uint64_t max_cw = *std::max_element(cws.begin(), cws.end()); // widest chunk of the schedule
    const uint64_t z_pitch = (max_cw + 15) & ~15ull, z_bytes = ext_tiled ? n_ext * z_pitch * 8 + 256 : 0; // the compact extension of one chunk
    const uint64_t stage_bytes = NS * n * max_cw * 8 + z_bytes;
    // With a lent workspace (a caller that plans its HBM: host/starks.hpp) the staging comes out of the lent buffer's tail and nothing
    // is allocated; the transforms of this call see the rest.  Later work on the stream is ordered behind this call's kernels, which
    // wait for the last upload, so the tail is free again when the call's work is done.
    struct WorkspaceCarve {
        mi_ctx *c; uint64_t saved = 0;
        ~WorkspaceCarve() { if (saved) c->workspace_bytes = c->workspace_limit = saved; }
    } carve{c};
    u64 *stage_base = nullptr;
    if (c->workspace_lent && c->workspace_bytes >= stage_bytes + (2ull << 30)) {
        carve.saved = c->workspace_bytes;
        c->workspace_bytes = c->workspace_limit = (carve.saved - stage_bytes) & ~(uint64_t)255;
        stage_base = c->workspace + c->workspace_bytes / 8;
    } else {
        if (c->stage_bytes < stage_bytes) {
            MI_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->stage) MI_HIP_CHECK(hipFree(c->stage));
            c->stage = nullptr;
            c->stage_bytes = 0;
            hipError_t e = hipMalloc((void **)&c->stage, stage_bytes);
            if (e != hipSuccess) {
                mi_set_error("cannot allocate %llu bytes of upload staging: %s", (unsigned long long)stage_bytes, hipGetErrorString(e));
                return MI_ERR_NOMEM;
            }
            c->stage_bytes = stage_bytes;
        }
        stage_base = c->stage;
    }
    u64 *const st[NS] = {stage_base, stage_base + n * max_cw, stage_base + 2 * n * max_cw};
    u64 *const zbuf = ext_tiled ? (u64 *)(((uintptr_t)(stage_base + NS * n * max_cw) + 127) & ~(uintptr_t)127) : nullptr;
    // the copy streams must not overtake work already queued on the compute stream that still reads the staging buffers
    for (int i = 0; i < NS; i++) MI_HIP_CHECK(hipEventRecord(c->ev_consumed[i], c->stream));
    if (packed) {
        const uint64_t need = n * max_cw * 8;
        if (c->pack_stage_bytes < need) {
            MI_HIP_CHECK(hipStreamSynchronize(c->copy_stream[0]));
            for (int i = 0; i < 3; i++) {
                if (c->pack_stage[i]) MI_HIP_CHECK(hipHostFree(c->pack_stage[i]));
                c->pack_stage[i] = nullptr;
                MI_HIP_CHECK(hipHostMalloc((void **)&c->pack_stage[i], need, hipHostMallocDefault));
                if (!c->ev_pack_sent[i]) MI_HIP_CHECK(hipEventCreateWithFlags(&c->ev_pack_sent[i], hipEventDisableTiming));
            }
            c->pack_stage_bytes = need;
        }
    }
    static const bool timing = getenv("MI_UPLOAD_TIMING") != nullptr; // stderr: where the host thread's time goes, per call
    double t_wait = 0, t_pack = 0, t_enq = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    struct Report { const bool on; double &w, &p, &e; uint64_t n; ~Report() { if (on) fprintf(stderr, "mi_lde_merkle_host: %llu chunks; host thread: waiting for a staging buffer %.1f ms, packing %.1f ms, enqueueing copies %.1f ms\n", (unsigned long long)n, w * 1e3, p * 1e3, e * 1e3); } } report{timing, t_wait, t_pack, t_enq, n_chunks};
    auto upload_packed = [&](uint64_t k) -> int {
        const uint64_t cw = cws[k], slot = k % 3;
        u64 *hs = c->pack_stage[slot];
        const double t0 = now();
        MI_HIP_CHECK(hipEventSynchronize(c->ev_pack_sent[slot])); // the copy that last read this staging buffer (three chunks ago, or in
                                                                   // an earlier call) is done; a never-recorded event is complete
        const double t1 = now();
        t_wait += t1 - t0;
        const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pack_threads, (n * cw * 8) >> 22)); // at least 4 MiB per thread
        std::vector<std::thread> th;
        const uint64_t rows_per = (n + T - 1) / T;
        for (int t = 0; t < T; t++) {
            const uint64_t r0 = (uint64_t)t * rows_per, r1 = std::min(n, r0 + rows_per);
            if (r0 >= r1) break;
            auto work = [=]() {
                const uint64_t *src = trace_host + r0 * ncols + c0s[k];
                u64 *dst = hs + r0 * cw;
                if (cw % 8 == 0) { // 64-byte groups: unaligned loads, streaming stores (the staging is read next by the DMA engine, not by
                                   // this core: no write-allocate, no cache pollution)
                    for (uint64_t r = r0; r < r1; r++, src += ncols, dst += cw)
                        for (uint64_t j = 0; j < cw; j += 8) {
                            const __m128i v0 = _mm_loadu_si128((const __m128i *)(src + j)), v1 = _mm_loadu_si128((const __m128i *)(src + j + 2));
                            const __m128i v2 = _mm_loadu_si128((const __m128i *)(src + j + 4)), v3 = _mm_loadu_si128((const __m128i *)(src + j + 6));
                            _mm_stream_si128((__m128i *)(dst + j), v0); _mm_stream_si128((__m128i *)(dst + j + 2), v1);
                            _mm_stream_si128((__m128i *)(dst + j + 4), v2); _mm_stream_si128((__m128i *)(dst + j + 6), v3);
                        }
                    _mm_sfence();
                } else {
                    for (uint64_t r = r0; r < r1; r++, src += ncols, dst += cw) memcpy(dst, src, cw * 8);
                }
            };
            if (T == 1) work(); // a small chunk: not worth a thread
            else th.emplace_back(work);
        }
        for (auto &t : th) t.join();
        const double t2 = now();
        t_pack += t2 - t1;
        MI_HIP_CHECK(hipStreamWaitEvent(c->copy_stream[0], c->ev_consumed[k % NS], 0)); // the LDE that read this device buffer is done
        MI_HIP_CHECK(hipMemcpyAsync(st[k % NS], hs, n * cw * 8, hipMemcpyHostToDevice, c->copy_stream[0]));
        MI_HIP_CHECK(hipEventRecord(c->ev_pack_sent[slot], c->copy_stream[0]));
        for (int s = 0; s < 2; s++) MI_HIP_CHECK(hipEventRecord(c->ev_uploaded[k % NS][s], c->copy_stream[0]));
        t_enq += now() - t2;
        return MI_OK;
    };
    // A PAGE-LOCKED trace (hipHostMalloc / mi_host_register) could let some chunks skip the host's packers -- the packed form moves at the
    // rate 16 threads gather rows (51-53 GB/s end to end against the link's 57.3) --: with MI_UPLOAD_STRIDED_EVERY = k > 0 every k-th chunk
    // (never the first two: the kernels wait for them) is read IN PLACE by the DMA engines as a strided 2-D copy on the second copy stream
    // while the packers work on the next one.  MEASURED SLOWER and therefore OFF by default (profiles/r05_upload_mix_ab.txt, full size,
    // page-locked trace: all packed 868-873 ms per step; k = 4: 969-981; k = 3: 1 008; k = 2: 1 068): a 32-column strided read crosses the
    // link at 38 GB/s (256-byte requests) and takes from the packed copies beside it more than the packers gain.  The knob stays for wider
    // chunks on another host.  A pageable trace takes the packed form throughout (the DMA engines cannot read it).
    bool src_locked = false;
    int strided_every = 0;
    if (packed) {
        if (const char *e = getenv("MI_UPLOAD_STRIDED_EVERY")) strided_every = atoi(e);
        if (strided_every > 0) {
            hipPointerAttribute_t a0, a1;
            src_locked = hipPointerGetAttributes(&a0, trace_host) == hipSuccess && a0.type == hipMemoryTypeHost &&
                         hipPointerGetAttributes(&a1, trace_host + (n * ncols - 1)) == hipSuccess && a1.type == hipMemoryTypeHost;
            (void)hipGetLastError(); // (a pageable pointer is "invalid value" to the query, not an error of this call)
        }
    }
    auto strided_chunk = [&](uint64_t k) { return packed && src_locked && k >= 2 && strided_every > 0 && (k % (uint64_t)strided_every) == (uint64_t)strided_every - 1; };
    auto upload = [&](uint64_t k) -> int { // the chunk's upper and lower rows on two copy streams (two DMA engines)
        if (strided_chunk(k)) {
            const uint64_t cw = cws[k];
            MI_HIP_CHECK(hipStreamWaitEvent(c->copy_stream[1], c->ev_consumed[k % NS], 0)); // the LDE that read this device buffer is done
            MI_HIP_CHECK(hipMemcpy2DAsync(st[k % NS], cw * 8, trace_host + c0s[k], ncols * 8, cw * 8, n, hipMemcpyHostToDevice, c->copy_stream[1]));
            for (int s2 = 0; s2 < 2; s2++) MI_HIP_CHECK(hipEventRecord(c->ev_uploaded[k % NS][s2], c->copy_stream[1]));
            return MI_OK;
        }
        if (packed) return upload_packed(k);
        const uint64_t cw = cws[k], half = n / 2 ? n / 2 : n;
        for (int s = 0; s < 2; s++) {
            const uint64_t r0 = s * half, nr = s == 0 ? half : n - half;
            MI_HIP_CHECK(hipStreamWaitEvent(c->copy_stream[s], c->ev_consumed[k % NS], 0)); // the LDE that read this buffer is done
            if (nr)
                MI_HIP_CHECK(hipMemcpy2DAsync(st[k % NS] + r0 * cw, cw * 8, trace_host + r0 * ncols + c0s[k], ncols * 8, cw * 8, nr,
                                              hipMemcpyHostToDevice, c->copy_stream[s]));
            MI_HIP_CHECK(hipEventRecord(c->ev_uploaded[k % NS][s], c->copy_stream[s]));
        }
        return MI_OK;
    };
    auto absorb = [&](uint64_t k) -> int {
        if (ext_tiled) return launch_linear_hash_absorb_emit(c, (u64 *)nodes, zbuf, z_pitch, cws[k], n_ext, k == 0, k + 1 == n_chunks, (u64 *)ext, ncols, c0s[k]);
        if (ncols <= 4) // linear_hash copies rows of at most 4 elements instead of hashing them: one chunk, plain leaf kernel
            return launch_linear_hash_rows(c, (u64 *)nodes, (const u64 *)ext, ext_pitch, ncols, n_ext);
        const u64 *base = (const u64 *)ext + c0s[k];
        const uint64_t pitch = ext_pitch, width = cws[k];
        return launch_linear_hash_absorb(c, (u64 *)nodes, 1, &base, &pitch, &width, n_ext, k == 0, k + 1 == n_chunks);
    };
    // Per chunk on the compute stream: wait for its upload, extend it, absorb its columns.  The uploads run back to back
    // on the copy streams (chunk k + 1 only needs the staging buffer the LDE of chunk k - 1 has finished reading), so what
    // is left after the last upload is the extension and absorption of the last -- short -- chunk.
    MI_TRY(upload(0));
    for (uint64_t k = 0; k < n_chunks; k++) {
        for (int s = 0; s < 2; s++) MI_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_uploaded[k % NS][s], 0));
        if (ext_tiled) MI_TRY(launch_lde(c, zbuf, z_pitch, st[k % NS], cws[k], n_ext, n, cws[k]));
        else MI_TRY(launch_lde(c, (u64 *)ext + c0s[k], ext_pitch, st[k % NS], cws[k], n_ext, n, cws[k]));
        if (base && base_pitch) MI_TRY(launch_copy_2d(c, (u64 *)base + c0s[k], base_pitch, st[k % NS], cws[k], n, cws[k])); // the base-domain section stays too
        else if (base) MI_TRY(launch_tile_major(c, (u64 *)base, ncols, c0s[k], st[k % NS], cws[k], n, cws[k]));
        MI_HIP_CHECK(hipEventRecord(c->ev_consumed[k % NS], c->stream));
        MI_TRY(absorb(k));
        // enqueued after this chunk's kernels (the copy call may block the host), but its only dependency -- the LDE of chunk
        // k - 1 -- is long done: it starts as soon as chunk k's upload ends and runs beside the kernels above
        if (k + 1 < n_chunks) MI_TRY(upload(k + 1));
    }
    // the strided form reads trace_host from the copy engines: nothing of it may still be queued when the caller gets its trace back
    if (!packed || src_locked)
        for (int s = 0; s < 2; s++) MI_HIP_CHECK(hipStreamSynchronize(c->copy_stream[s]));
    return launch_merkle_levels(c, (u64 *)nodes, n_ext);
}

extern "C" int mi_lde_merkle_host(mi_ctx *c, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, const uint64_t *trace_host,
                                  uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    MI_OWN(c, ext);
    return lde_merkle_host_impl(c, nodes, ext, ext_pitch, nullptr, 0, trace_host, n, n_ext, ncols, chunk_cols);
}

extern "C" int mi_lde_merkle_host_keep(mi_ctx *c, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, uint64_t *base, uint64_t base_pitch,
                                       const uint64_t *trace_host, uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    MI_OWN(c, ext);
    MI_OWN(c, base);
    MI_REQUIRE(base, "null buffer");
    return lde_merkle_host_impl(c, nodes, ext, ext_pitch, base, base_pitch, trace_host, n, n_ext, ncols, chunk_cols);
}

extern "C" int mi_lde_merkle_host_keep_tiled(mi_ctx *c, uint64_t *nodes, uint64_t *ext, uint64_t ext_pitch, uint64_t *base_tiled,
                                             const uint64_t *trace_host, uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    MI_OWN(c, ext);
    MI_OWN(c, base_tiled);
    MI_REQUIRE(base_tiled, "null buffer");
    return lde_merkle_host_impl(c, nodes, ext, ext_pitch, base_tiled, 0, trace_host, n, n_ext, ncols, chunk_cols);
}

extern "C" int mi_lde_merkle_host_tiled(mi_ctx *c, uint64_t *nodes, uint64_t *ext_tiled, uint64_t *base, uint64_t base_pitch, const uint64_t *trace_host,
                                        uint64_t n, uint64_t n_ext, uint64_t ncols, uint64_t chunk_cols)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    MI_OWN(c, ext_tiled);
    MI_OWN(c, base);
    return lde_merkle_host_impl(c, nodes, ext_tiled, ncols, base, base_pitch, trace_host, n, n_ext, ncols, chunk_cols, true);
}

// Device source: the section is extended in column chunks into a compact buffer (out of the tail of the workspace the caller lent, or
// the context's staging allocation) and absorbed from there by the emitting leaf kernel.
extern "C" int mi_lde_merkle_dev_tiled(mi_ctx *c, uint64_t *nodes, uint64_t *ext_tiled, const uint64_t *src, uint64_t src_pitch, uint64_t n, uint64_t n_ext,
                                       uint64_t ncols)
{
    CTX_OK(c);
    MI_OWN(c, nodes);
    MI_OWN(c, ext_tiled);
    MI_OWN(c, src);
    if (n == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(nodes && ext_tiled && src, "null buffer");
    MI_REQUIRE(is_pow2(n) && is_pow2(n_ext) && n_ext >= n, "sizes must be powers of two with n_ext >= n");
    MI_REQUIRE(n_ext % 64 == 0 && ncols > 4 && src_pitch >= ncols, "a tile-major extension has a multiple of 64 rows and more than 4 columns");
    // chunk width: the leaf kernel likes long launches, the compact chunk and the transform's scratch have to fit the region the caller
    // lent (MI_TILED_CHUNK_COLS: experiments; multiples of 32)
    static const uint64_t chunk_cols = [] { const char *e = getenv("MI_TILED_CHUNK_COLS"); const long v = e ? atol(e) : 0; return v >= 32 && v <= 1024 ? (uint64_t)v & ~31ull : 128; }();
    uint64_t cw_max = std::min<uint64_t>(ncols, chunk_cols);
    auto z_need = [&](uint64_t cw) { return n_ext * ((cw + 15) & ~15ull) * 8 + 256; };
    auto fits = [&](uint64_t cw) { return c->workspace_bytes >= z_need(cw) + cw * (n + n_ext) * 8 + (1ull << 20); };
    if (c->workspace_lent)
        while (cw_max > 32 && !fits(cw_max)) cw_max = (cw_max - 1) & ~31ull; // the widest chunk the lent region holds beside the transform's scratch
    const uint64_t z_pitch = (cw_max + 15) & ~15ull, z_bytes = z_need(cw_max);
    struct WorkspaceCarve {
        mi_ctx *c; uint64_t saved = 0;
        ~WorkspaceCarve() { if (saved) c->workspace_bytes = c->workspace_limit = saved; }
    } carve{c};
    u64 *zraw = nullptr;
    if (c->workspace_lent && fits(cw_max)) {
        carve.saved = c->workspace_bytes;
        c->workspace_bytes = c->workspace_limit = (carve.saved - z_bytes) & ~(uint64_t)255;
        zraw = c->workspace + c->workspace_bytes / 8;
    } else {
        if (c->stage_bytes < z_bytes) {
            MI_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->stage) MI_HIP_CHECK(hipFree(c->stage));
            c->stage = nullptr;
            c->stage_bytes = 0;
            hipError_t e = hipMalloc((void **)&c->stage, z_bytes);
            if (e != hipSuccess) {
                mi_set_error("cannot allocate %llu bytes for a chunk's compact extension: %s", (unsigned long long)z_bytes, hipGetErrorString(e));
                return MI_ERR_NOMEM;
            }
            c->stage_bytes = z_bytes;
        }
        zraw = c->stage;
    }
    u64 *const zbuf = (u64 *)(((uintptr_t)zraw + 127) & ~(uintptr_t)127);
    for (uint64_t c0 = 0; c0 < ncols; c0 += cw_max) {
        const uint64_t cw = std::min(cw_max, ncols - c0);
        MI_TRY(launch_lde(c, zbuf, z_pitch, (const u64 *)src + c0, src_pitch, n_ext, n, cw));
        MI_TRY(launch_linear_hash_absorb_emit(c, (u64 *)nodes, zbuf, z_pitch, cw, n_ext, c0 == 0, c0 + cw == ncols, (u64 *)ext_tiled, ncols, c0));
    }
    return launch_merkle_levels(c, (u64 *)nodes, n_ext);
}

extern "C" int mi_tile_major_dev(mi_ctx *c, uint64_t *dst, uint64_t ncols_total, uint64_t col0, const uint64_t *src, uint64_t src_pitch,
                                 uint64_t nrows, uint64_t ncols)
{
    CTX_OK(c);
    MI_OWN(c, dst);
    MI_OWN(c, src);
    if (nrows == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(dst && src, "null buffer");
    MI_REQUIRE(nrows % 64 == 0, "a tile-major section has a multiple of 64 rows");
    MI_REQUIRE(src_pitch >= ncols && col0 + ncols <= ncols_total, "columns outside the section");
    return launch_tile_major(c, (u64 *)dst, ncols_total, col0, (const u64 *)src, src_pitch, nrows, ncols);
}

extern "C" int mi_untile_dev(mi_ctx *c, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src_tiled, uint64_t ncols_total, uint64_t nrows_total, uint64_t col0,
                             uint64_t row0, uint64_t nrows, uint64_t ncols)
{
    CTX_OK(c);
    MI_OWN(c, dst);
    MI_OWN(c, src_tiled);
    if (nrows == 0 || ncols == 0) return MI_OK;
    MI_REQUIRE(dst && src_tiled, "null buffer");
    MI_REQUIRE(nrows_total % 64 == 0 && row0 + nrows <= nrows_total && col0 + ncols <= ncols_total && dst_pitch >= ncols, "rows or columns outside the section");
    return launch_untile(c, (u64 *)dst, dst_pitch, (const u64 *)src_tiled, ncols_total, col0, row0, nrows, ncols);
}

extern "C" int mi_set_host_pack_threads(mi_ctx *c, int threads)
{
    CTX_OK(c);
    MI_REQUIRE(threads >= -1 && threads <= 256, "thread count out of range");
    c->pack_threads = threads;
    return MI_OK;
}

// page-locking works on whole pages: the range is widened to the pages that cover it (all of them hold the caller's bytes), so that a
// section in the middle of a malloc'ed area (pAddress + an offset) can be named as it is
extern "C" int mi_host_register(mi_ctx *c, void *p, uint64_t bytes)
{
    CTX_OK(c);
    MI_REQUIRE(p && bytes, "null range");
    const uintptr_t a = (uintptr_t)p & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes + 4095) & ~(uintptr_t)4095;
    MI_HIP_CHECK(hipHostRegister((void *)a, e - a, hipHostRegisterPortable)); // every device of a multi-device commit reads it over its own link
    return MI_OK;
}

extern "C" int mi_host_unregister(mi_ctx *c, void *p)
{
    CTX_OK(c);
    MI_REQUIRE(p, "null pointer");
    MI_HIP_CHECK(hipHostUnregister((void *)((uintptr_t)p & ~(uintptr_t)4095)));
    return MI_OK;
}

// ------------------------------------------------------------------ FRI and the rest
extern "C" int mi_fri_fold_dev(mi_ctx *c, uint64_t *out, const uint64_t *pol, unsigned prev_bits, unsigned cur_bits,
                               unsigned nbits_ext, const uint64_t x[3])
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_OWN(c, pol);
    MI_REQUIRE(out && pol && x, "null buffer");
    const u64 xe[3] = {x[0], x[1], x[2]};
    MI_REQUIRE(cur_bits <= 40, "bad FRI step sizes");
    return launch_fri_fold(c, (u64 *)out, (const u64 *)pol, prev_bits, cur_bits, nbits_ext, xe, 0, 1ull << cur_bits);
}

extern "C" int mi_fri_fold_range_dev(mi_ctx *c, uint64_t *out, const uint64_t *pol, unsigned prev_bits, unsigned cur_bits,
                                     unsigned nbits_ext, const uint64_t x[3], uint64_t g0, uint64_t g_count)
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_OWN(c, pol);
    MI_REQUIRE(out && pol && x, "null buffer");
    const u64 xe[3] = {x[0], x[1], x[2]};
    return launch_fri_fold(c, (u64 *)out, (const u64 *)pol, prev_bits, cur_bits, nbits_ext, xe, g0, g_count);
}

extern "C" int mi_fri_transpose_dev(mi_ctx *c, uint64_t *aux, const uint64_t *pol, uint64_t degree, unsigned tbits)
{
    CTX_OK(c);
    MI_OWN(c, aux);
    MI_OWN(c, pol);
    MI_REQUIRE((aux && pol) || degree == 0, "null buffer");
    return launch_fri_transpose(c, (u64 *)aux, (const u64 *)pol, degree, tbits);
}

extern "C" int mi_q_split_dev(mi_ctx *c, uint64_t *qq2, const uint64_t *qq1, uint64_t n, uint64_t n_ext, unsigned qdeg)
{
    CTX_OK(c);
    MI_OWN(c, qq2);
    MI_OWN(c, qq1);
    MI_REQUIRE(qq2 && qq1, "null buffer");
    return launch_q_split(c, (u64 *)qq2, (const u64 *)qq1, n, n_ext, qdeg);
}

extern "C" int mi_evmap_dev(mi_ctx *c, uint64_t *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits,
                            const uint64_t *const *pol_ptr, const uint32_t *pol_dim, const uint64_t *pol_stride,
                            const uint8_t *prime, const uint64_t *lev, const uint64_t *lpev)
{
    CTX_OK(c);
    if (n_evals == 0) return MI_OK;
    MI_REQUIRE(evals && pol_ptr && pol_dim && pol_stride && prime && lev && lpev, "null buffer");
    MI_OWN(c, evals); MI_OWN(c, lev); MI_OWN(c, lpev);
    for (uint64_t i = 0; i < n_evals; i++) MI_OWN(c, pol_ptr[i]);
    return launch_evmap(c, (u64 *)evals, n_evals, n, ext_bits, (const u64 *const *)pol_ptr, pol_dim, (const u64 *)pol_stride,
                        prime, (const u64 *)lev, (const u64 *)lpev, 0, n);
}

// The same with some of the polynomials in TILE-MAJOR sections ([n_ext / 64][width][64], mi_lde_merkle_dev_tiled): tile_cols[i] != 0 is the
// width of polynomial i's section and pol_ptr[i] its element of row 0 (section + 64 * column); pol_stride[i] is ignored for it.
extern "C" int mi_evmap_tiled_dev(mi_ctx *c, uint64_t *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits,
                                  const uint64_t *const *pol_ptr, const uint32_t *pol_dim, const uint64_t *pol_stride,
                                  const uint8_t *prime, const uint64_t *tile_cols, const uint64_t *lev, const uint64_t *lpev)
{
    CTX_OK(c);
    if (n_evals == 0) return MI_OK;
    MI_REQUIRE(evals && pol_ptr && pol_dim && pol_stride && prime && tile_cols && lev && lpev, "null buffer");
    MI_OWN(c, evals); MI_OWN(c, lev); MI_OWN(c, lpev);
    for (uint64_t i = 0; i < n_evals; i++) MI_OWN(c, pol_ptr[i]);
    return launch_evmap(c, (u64 *)evals, n_evals, n, ext_bits, (const u64 *const *)pol_ptr, pol_dim, (const u64 *)pol_stride,
                        prime, (const u64 *)lev, (const u64 *)lpev, 0, n, tile_cols);
}

// the partial sums over rows [row0, row0 + nrows) of the base domain (a row shard's share; the shares add up to mi_evmap_dev's result)
extern "C" int mi_evmap_range_dev(mi_ctx *c, uint64_t *evals, uint64_t n_evals, uint64_t n, unsigned ext_bits,
                                  const uint64_t *const *pol_ptr, const uint32_t *pol_dim, const uint64_t *pol_stride,
                                  const uint8_t *prime, const uint64_t *lev, const uint64_t *lpev, uint64_t row0, uint64_t nrows)
{
    CTX_OK(c);
    if (n_evals == 0 || nrows == 0) return MI_OK;
    MI_REQUIRE(evals && pol_ptr && pol_dim && pol_stride && prime && lev && lpev, "null buffer");
    MI_OWN(c, evals); MI_OWN(c, lev); MI_OWN(c, lpev);
    for (uint64_t i = 0; i < n_evals; i++) MI_OWN(c, pol_ptr[i]);
    return launch_evmap(c, (u64 *)evals, n_evals, n, ext_bits, (const u64 *const *)pol_ptr, pol_dim, (const u64 *)pol_stride,
                        prime, (const u64 *)lev, (const u64 *)lpev, row0, nrows);
}

extern "C" int mi_batch_inverse3_dev(mi_ctx *c, uint64_t *res, const uint64_t *src, uint64_t n)
{
    CTX_OK(c);
    MI_OWN(c, res);
    MI_OWN(c, src);
    MI_REQUIRE((res && src) || n == 0, "null buffer");
    return launch_batch_inverse3(c, (u64 *)res, (const u64 *)src, n);
}

extern "C" int mi_calculate_h1h2_dev(mi_ctx *c, uint64_t *h1, uint64_t h1_stride, uint64_t *h2, uint64_t h2_stride, const uint64_t *f,
                                     uint64_t f_stride, const uint64_t *t, uint64_t t_stride, unsigned dim, uint64_t n)
{
    CTX_OK(c);
    MI_OWN(c, h1);
    MI_OWN(c, h2);
    MI_OWN(c, f);
    MI_OWN(c, t);
    MI_REQUIRE((h1 && h2 && f && t) || n == 0, "null buffer");
    return launch_calculate_h1h2(c, (u64 *)h1, h1_stride, (u64 *)h2, h2_stride, (const u64 *)f, f_stride, (const u64 *)t, t_stride, dim, n);
}

extern "C" int mi_calculate_z_dev(mi_ctx *c, uint64_t *z, uint64_t z_stride, const uint64_t *num, uint64_t num_stride, const uint64_t *den,
                                  uint64_t den_stride, uint64_t n, int *closes)
{
    CTX_OK(c);
    MI_OWN(c, z);
    MI_OWN(c, num);
    MI_OWN(c, den);
    MI_REQUIRE((z && num && den) || n == 0, "null buffer");
    return launch_calculate_z(c, (u64 *)z, z_stride, (const u64 *)num, num_stride, (const u64 *)den, den_stride, n, closes);
}

extern "C" int mi_calculate_z_batch_dev(mi_ctx *c, unsigned nprod, uint64_t *const *z, const uint64_t *z_stride, const uint64_t *const *num,
                                        const uint64_t *num_stride, const uint64_t *const *den, const uint64_t *den_stride, uint64_t n, int *closes)
{
    CTX_OK(c);
    MI_REQUIRE(nprod == 0 || (z && z_stride && num && num_stride && den && den_stride), "null argument array");
    for (unsigned i = 0; i < nprod; i++) { MI_OWN(c, z[i]); MI_OWN(c, num[i]); MI_OWN(c, den[i]); }
    return launch_calculate_z_batch(c, nprod, (u64 *const *)z, z_stride, (const u64 *const *)num, num_stride, (const u64 *const *)den, den_stride, n, closes);
}

extern "C" int mi_geom_seq_dev(mi_ctx *c, uint64_t *out, uint64_t n, uint64_t start, uint64_t ratio)
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_REQUIRE(out || n == 0, "null buffer");
    return launch_geom_seq(c, (u64 *)out, n, start, ratio);
}

extern "C" int mi_geom_seq3_dev(mi_ctx *c, uint64_t *out, uint64_t n, const uint64_t ratio[3])
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_REQUIRE((out || n == 0) && ratio, "null buffer");
    const u64 r[3] = {ratio[0], ratio[1], ratio[2]};
    return launch_geom_seq3(c, (u64 *)out, n, r);
}

extern "C" int mi_x_div_x_sub_dev(mi_ctx *c, uint64_t *out, const uint64_t *x, uint64_t n, const uint64_t xi[3])
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_OWN(c, x);
    MI_REQUIRE(((out && x) || n == 0) && xi, "null buffer");
    const u64 e[3] = {xi[0], xi[1], xi[2]};
    return launch_x_div_x_sub(c, (u64 *)out, (const u64 *)x, n, e);
}

extern "C" int mi_zhinv(mi_ctx *c, uint64_t *out, unsigned nbits, unsigned nbits_ext)
{
    CTX_OK(c);
    MI_REQUIRE(out && nbits > 0 && nbits < nbits_ext && nbits_ext - nbits <= 8, "need 0 < nbits < nbits_ext, blow-up <= 256");
    // ZHInv[i] = 1 / (shift^(2^nbits) * w(extendBits)^i - 1)   (zhInv.cpp:7-31)
    const unsigned ext = nbits_ext - nbits;
    const uint64_t cnt = 1ull << ext;
    u64 sn = 49;
    for (unsigned i = 0; i < nbits; i++) sn = gl::mul(sn, sn);
    u64 w = 7277203076849721926ULL;
    for (unsigned i = ext; i < 32; i++) w = gl::mul(w, w);
    MI_TRY(launch_zhinv(c, c->small, cnt, sn, w)); // cnt <= 256 values fit the 4 KiB scratch
    MI_HIP_CHECK(hipMemcpyAsync(out, c->small, cnt * 8, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

// ------------------------------------------------------------------ utilities
extern "C" int mi_fill_synthetic_dev(mi_ctx *c, uint64_t *out, uint64_t count, uint64_t seed)
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_REQUIRE(out || count == 0, "null buffer");
    return launch_fill_synthetic(c, (u64 *)out, count, seed);
}

extern "C" int mi_fill_synthetic_2d_dev(mi_ctx *c, uint64_t *out, uint64_t out_pitch, uint64_t nrows, uint64_t ncols,
                                        uint64_t global_cols, uint64_t col0, uint64_t seed)
{
    CTX_OK(c);
    MI_OWN(c, out);
    MI_REQUIRE(out || nrows * ncols == 0, "null buffer");
    MI_REQUIRE(out_pitch >= ncols && col0 + ncols <= global_cols, "bad column window");
    return launch_fill_synthetic_2d(c, (u64 *)out, out_pitch, nrows, ncols, global_cols, col0, seed);
}

extern "C" int mi_copy_2d_dev(mi_ctx *c, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src, uint64_t src_pitch,
                              uint64_t nrows, uint64_t ncols)
{
    CTX_OK(c);
    MI_OWN(c, dst);
    MI_OWN(c, src);
    MI_REQUIRE((dst && src) || nrows * ncols == 0, "null buffer");
    MI_REQUIRE(dst_pitch >= ncols && src_pitch >= ncols, "pitch smaller than ncols");
    return launch_copy_2d(c, (u64 *)dst, dst_pitch, (const u64 *)src, src_pitch, nrows, ncols);
}

extern "C" void *mi_dev_alloc(mi_ctx *c, uint64_t bytes)
{
    if (!c) return nullptr;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
    if (e != hipSuccess) {
        mi_set_error("hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return nullptr;
    }
    mi_own_add(p, bytes ? bytes : 8, c->logical, "mi_dev_alloc");
    return p;
}

extern "C" int mi_dev_free(mi_ctx *c, void *p)
{
    CTX_OK(c);
    if (p) {
        MI_OWN(c, p);
        MI_HIP_CHECK(hipStreamSynchronize(c->stream));
        MI_HIP_CHECK(hipFree(p));
        mi_own_del(p);
    }
    return MI_OK;
}


// ------------------------------------------------------------------ sparse device memory (HIP virtual-memory management)
// A row-sharded proof (host/starks.hpp) addresses FULL-HEIGHT sections on every device -- the evaluator's programs, the code-object cache
// and the openings keep the one-device image's offsets -- but a device only ever touches its own rows of them: 20 GB of the 157 GB an
// extended part spans at zkEVM size.  mi_vmm_reserve takes the ADDRESS RANGE, mi_vmm_back puts physical memory under the parts that are
// used, in 64 MiB pieces (tools/vmm_probe.hip: this stack maps under a terabyte of addresses).  Nothing outside a backed part may be
// touched: such an access faults like any wild pointer.
namespace {
// Pieces of ONE size at multiples of it.  The driver of this stack (ROCm 7.2) refuses hipMemSetAccess on a second mapping of a
// reservation when the two mappings differ in size and lie far apart (tools/vmm_probe3.hip, profiles/r05_vmm_probe3.txt: 74 MiB at
// 2 MiB, then 4 MiB at 400 GiB -> "invalid argument"; equal sizes at any distance are taken), so every piece is VMM_PIECE long.
constexpr uint64_t VMM_PIECE = 64ull << 20;
struct VmmRange { int device = 0; uint64_t bytes = 0, backed = 0; std::map<uint64_t, hipMemGenericAllocationHandle_t> pieces; }; // by piece index
std::mutex g_vmm_mu;
std::map<uintptr_t, VmmRange> g_vmm;
}
extern "C" int mi_vmm_reserve(mi_ctx *c, uint64_t bytes, void **base)
{
    CTX_OK(c);
    MI_REQUIRE(base && bytes, "null argument");
    *base = nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0;
    MI_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    MI_REQUIRE(gran && VMM_PIECE % gran == 0, "the driver's mapping granularity does not divide 64 MiB");
    const uint64_t size = (bytes + VMM_PIECE - 1) / VMM_PIECE * VMM_PIECE;
    void *p = nullptr;
    MI_HIP_CHECK(hipMemAddressReserve(&p, size, VMM_PIECE, nullptr, 0));
    {
        std::lock_guard<std::mutex> lock(g_vmm_mu);
        VmmRange &R = g_vmm[(uintptr_t)p];
        R.device = c->device; R.bytes = size;
    }
    mi_own_add(p, size, c->logical, "mi_vmm_reserve");
    *base = p;
    return MI_OK;
}
// physical memory under [offset, offset + bytes) of the range (widened to 64 MiB boundaries); what is backed already stays as it is.
// Fresh pieces are NOT zeroed.
extern "C" int mi_vmm_back(mi_ctx *c, void *base, uint64_t offset, uint64_t bytes)
{
    CTX_OK(c);
    if (!bytes) return MI_OK;
    std::lock_guard<std::mutex> lock(g_vmm_mu);
    auto it = g_vmm.find((uintptr_t)base);
    MI_REQUIRE(it != g_vmm.end(), "not a range of mi_vmm_reserve");
    VmmRange &R = it->second;
    MI_REQUIRE(R.device == c->device, "the range belongs to another device");
    MI_REQUIRE(offset + bytes <= R.bytes && offset + bytes >= offset, "beyond the reserved range");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = R.device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (uint64_t i = offset / VMM_PIECE; i <= (offset + bytes - 1) / VMM_PIECE; i++) {
        if (R.pieces.count(i)) continue;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, VMM_PIECE, &prop, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            mi_set_error("mi_vmm_back: cannot create a 64 MiB piece of device memory (%s); %.1f GB backed in this range, %.1f of %.1f GB free on device %d",
                         hipGetErrorString(e), R.backed / 1e9, fr / 1e9, tot / 1e9, R.device);
            return MI_ERR_NOMEM;
        }
        char *at = (char *)base + i * VMM_PIECE;
        e = hipMemMap(at, VMM_PIECE, 0, h, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipMemRelease(h);
            mi_set_error("mi_vmm_back: hipMemMap of the piece at offset %.2f GB failed: %s", i * VMM_PIECE / 1e9, hipGetErrorString(e));
            return MI_ERR_HIP;
        }
        e = hipMemSetAccess(at, VMM_PIECE, &acc, 1);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipMemUnmap(at, VMM_PIECE);
            (void)hipMemRelease(h);
            mi_set_error("mi_vmm_back: hipMemSetAccess of the piece at offset %.2f GB failed: %s", i * VMM_PIECE / 1e9, hipGetErrorString(e));
            return MI_ERR_HIP;
        }
        R.pieces[i] = h;
        R.backed += VMM_PIECE;
    }
    return MI_OK;
}
// one more device may read and write the backed parts of the range directly (a peer's kernels write a shard's row image); parts backed
// LATER are not covered: call it again after mi_vmm_back
extern "C" int mi_vmm_allow_peer(mi_ctx *c, void *base, int peer_device)
{
    CTX_OK(c);
    std::lock_guard<std::mutex> lock(g_vmm_mu);
    auto it = g_vmm.find((uintptr_t)base);
    MI_REQUIRE(it != g_vmm.end(), "not a range of mi_vmm_reserve");
    if (peer_device == it->second.device) return MI_OK;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = peer_device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (auto &pc : it->second.pieces) MI_HIP_CHECK(hipMemSetAccess((char *)base + pc.first * VMM_PIECE, VMM_PIECE, &acc, 1));
    return MI_OK;
}
extern "C" int mi_vmm_backed_bytes(mi_ctx *c, void *base, uint64_t *bytes)
{
    CTX_OK(c);
    std::lock_guard<std::mutex> lock(g_vmm_mu);
    auto it = g_vmm.find((uintptr_t)base);
    MI_REQUIRE(it != g_vmm.end() && bytes, "not a range of mi_vmm_reserve");
    *bytes = it->second.backed;
    return MI_OK;
}
extern "C" int mi_vmm_free(mi_ctx *c, void *base)
{
    CTX_OK(c);
    if (!base) return MI_OK;
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::lock_guard<std::mutex> lock(g_vmm_mu);
    auto it = g_vmm.find((uintptr_t)base);
    MI_REQUIRE(it != g_vmm.end(), "not a range of mi_vmm_reserve");
    for (auto &pc : it->second.pieces) {
        (void)hipMemUnmap((char *)base + pc.first * VMM_PIECE, VMM_PIECE);
        (void)hipMemRelease(pc.second);
    }
    (void)hipMemAddressFree(base, it->second.bytes);
    g_vmm.erase(it);
    mi_own_del(base);
    return MI_OK;
}

extern "C" int mi_copy_h2d(mi_ctx *c, void *dst, const void *src, uint64_t bytes)
{
    CTX_OK(c);
    MI_OWN(c, dst);
    MI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

// zeros, enqueued on the context's stream
extern "C" int mi_dev_zero(mi_ctx *c, void *p, uint64_t bytes)
{
    CTX_OK(c);
    MI_OWN(c, p);
    if (!bytes) return MI_OK;
    MI_REQUIRE(p, "null buffer");
    MI_HIP_CHECK(hipMemsetAsync(p, 0, bytes, c->stream));
    return MI_OK;
}

// host rows of `width` words at pitch src_pitch -> device rows at pitch dst_pitch (a column window of a row-major section); returns when done
extern "C" int mi_copy_h2d_2d(mi_ctx *c, uint64_t *dst, uint64_t dst_pitch, const uint64_t *src, uint64_t src_pitch, uint64_t width, uint64_t rows)
{
    CTX_OK(c);
    MI_OWN(c, dst);
    if (!width || !rows) return MI_OK;
    MI_REQUIRE(dst && src && dst_pitch >= width && src_pitch >= width, "bad 2-D copy");
    MI_HIP_CHECK(hipMemcpy2DAsync(dst, dst_pitch * 8, src, src_pitch * 8, width * 8, rows, hipMemcpyHostToDevice, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

extern "C" int mi_copy_d2h(mi_ctx *c, void *dst, const void *src, uint64_t bytes)
{
    CTX_OK(c);
    MI_OWN(c, src);
    MI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

extern "C" int mi_timer_start(mi_ctx *c, int slot)
{
    CTX_OK(c);
    MI_REQUIRE(slot >= 0 && slot < mi_ctx::N_TIMERS, "timer slot out of range");
    MI_HIP_CHECK(hipEventRecord(c->ev_start[slot], c->stream));
    return MI_OK;
}

extern "C" int mi_timer_stop(mi_ctx *c, int slot)
{
    CTX_OK(c);
    MI_REQUIRE(slot >= 0 && slot < mi_ctx::N_TIMERS, "timer slot out of range");
    MI_HIP_CHECK(hipEventRecord(c->ev_stop[slot], c->stream));
    return MI_OK;
}

extern "C" int mi_timer_elapsed_ms(mi_ctx *c, int slot, float *ms)
{
    CTX_OK(c);
    MI_REQUIRE(slot >= 0 && slot < mi_ctx::N_TIMERS && ms, "bad timer arguments");
    MI_HIP_CHECK(hipEventSynchronize(c->ev_stop[slot]));
    MI_HIP_CHECK(hipEventElapsedTime(ms, c->ev_start[slot], c->ev_stop[slot]));
    return MI_OK;
}
