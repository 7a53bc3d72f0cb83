// multi.hip -- ONE process, several devices: the stage commit (extendPol + merkelize, starks.cpp:48-59,133-141,214-221,284-292) sharded
// over the GPUs of a node behind the C ABI, for a caller like the reference's Prover -- one process, one proof in flight
// (prover.cpp:187-260) -- that cannot be split into torch.distributed ranks (shard.py is that other form; same plan, same result).
//
// the LDE is independent per COLUMN, a Merkle leaf needs a whole ROW, so there is exactly one exchange.  Shard g (a device;
// several shards may share one, which is how a single-GPU box rehearses the path) is dealt column tiles in rounds -- round k: shard g
// extends the g-th tile of the round, at most 32 columns, so that the G tiles of a round are a contiguous piece of every row -- and owns
// the rows [g * n_ext / G, (g + 1) * n_ext / G) of the tree.  Per round and shard, each on its own stream:
//     upload stream    the tile's columns of the base-domain trace come up over THAT device's PCIe link (host threads pack them into
//                      page-locked staging, one contiguous copy each), or across from the device that holds the section (stages 2-4)
//     compute stream   LDE of the tile  ->  [later]  absorb the previous round's columns of my rows into the leaf sponges
//     exchange stream  my tile's rows go to the shard that owns them: one hipMemcpyPeerAsync per peer and round, every xGMI link of the
//                      device busy at once (point-to-point, no ring); optionally the whole tile also goes into the ROW-MAJOR image of the
//                      extension on one device (what Starks::genProof's constraint evaluation reads)
// then each shard builds the subtree over its rows, the G subtree roots meet on shard 0 and the top log2 G levels are hashed there:
// the same root, node for node, as the single-device tree.  Openings: the row's values from the owning shard's column windows, its
// lower siblings from that shard's subtree, the top log2 G from the roots' tree.
//
// Round 5.  (1) TRANSIENT commits (mi_multi_set_transient: what a row-sharded Starks::genProof asks for): every shard names a row image,
// nobody will open values from the tree, so a tile's rows are written ONCE -- by a kernel of the extending shard straight into each
// owner's row image, where the leaf sponge then absorbs them at the image's pitch -- instead of twice (contiguous windows + image);
// the extended tiles live in a ring of two, the windows are gone: 21.5 GB of buffers per device at zkEVM size instead of 36.5, half the
// bytes on the links.  (2) DEVICE GROUPS (mi_multi_create2(..., group_same_device)): shards that share a physical device share that
// device's streams, staging, tile ring, NTT workspace and per-context pools -- ordered by the one stream -- so that eight logical shards
// at the full 2^23 rows fit one GPU.  Off (the default), every shard is its own group, as on a real node.  (3) Cross-device 2-D copies
// (row images, the image, the base section, device sources) are KERNELS of the device that issues them (k_rows_2d), not
// hipMemcpy2DAsync(hipMemcpyDefault): 256-byte rows at a 5 320-byte pitch across xGMI behave the same on every runtime, and a
// one-GPU box runs the very code an eight-GPU node runs.  (4) MI_MULTI_CHECK=1: every buffer, stream and event belongs to a logical
// shard; an operation issued for shard g that touches another shard's is refused (common.h mi_own_check), peer operands are declared.
#include "common.h"
#include <algorithm>
#include <chrono>
#include <emmintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

namespace {

struct Plan { // merlin-zkevm-prover_amd/shard.py ShardPlan, word for word in meaning
    uint64_t n = 0, n_ext = 0, ncols = 0;
    uint32_t G = 1, tile = 32;
    uint64_t per_rank = 0, rows_per_rank = 0;
    std::vector<uint64_t> round_w, round_c0, local_c0;
    void init(uint64_t n_, uint64_t n_ext_, uint64_t ncols_, uint32_t G_)
    {
        n = n_; n_ext = n_ext_; ncols = ncols_; G = G_;
        per_rank = (ncols + 8 * G - 1) / (8 * G) * 8;
        round_w.clear(); round_c0.clear(); local_c0.clear();
        for (uint64_t c = 0; c < per_rank; c += tile) round_w.push_back(std::min<uint64_t>(tile, per_rank - c));
        uint64_t acc = 0;
        for (uint64_t w : round_w) { round_c0.push_back(G * acc); local_c0.push_back(acc); acc += w; }
        rows_per_rank = n_ext / G;
    }
    size_t rounds() const { return round_w.size(); }
    uint64_t c0(size_t k, uint32_t g) const { return round_c0[k] + g * round_w[k]; }
    uint64_t width(size_t k, uint32_t g) const { const uint64_t c = c0(k, g); return c >= ncols ? 0 : std::min(round_w[k], ncols - c); }
    uint64_t ext_base(size_t k) const { return n_ext * local_c0[k]; }                       // in a shard's `ext`: my round-k tile, [n_ext x w] at pitch w
    uint64_t recv_off(size_t k, uint32_t peer) const { return rows_per_rank * c0(k, peer); } // in a shard's `recv`: peer's round-k tile restricted to my rows
};

struct Stats { double lde_ms = 0, absorb_ms = 0, wait_ms = 0, upload_ms = 0; std::vector<uint64_t> bytes_to; };

// rows x w words from src (row pitch spitch) to dst (row pitch dpitch): the 2-D copies of a commit as a kernel of the issuing device.
// One element per lane and step: a row segment of 32 columns is one 256-byte run on both sides, whatever the pitches' alignment
// (a 665-column section starts its rows on 8-byte boundaries only).
__global__ __launch_bounds__(256) void k_rows_2d(u64 *__restrict__ dst, uint64_t dpitch, const u64 *__restrict__ src, uint64_t spitch, uint32_t w, uint64_t rows)
{
    const uint64_t total = rows * w, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const uint64_t r = e / w;
        const uint32_t c = (uint32_t)(e - r * w);
        dst[r * dpitch + c] = src[r * spitch + c];
    }
}

} // namespace

struct mi_multi {
    uint32_t G = 0;
    std::vector<int> dev;
    std::vector<mi_ctx *> ctx;
    // device groups: lead[g] = the first shard of g's group (g itself unless mi_multi_create2 grouped the shards of one physical device).
    // A group has ONE set of streams (cs[g] == cs[lead[g]] ...), one staging ring, one tile ring, one NTT workspace; its contexts share
    // the leader's pools (CtxPool).  Everything a group does is ordered by its streams.
    std::vector<uint32_t> lead;
    bool grouped = false;
    std::vector<hipStream_t> cs, xs, us, us2; // compute, exchange, upload (us2: the lower rows of a strided upload -- a second DMA engine)
    std::vector<hipStream_t> own_streams;     // what mi_multi_destroy destroys (a grouped shard's entries above are its leader's)
    bool transient_next = false;              // mi_multi_set_transient: the NEXT commit keeps no rows (row images for every shard)
    // MI_MULTI_CHECK: which group a stream or an event was created for; an event is recorded on a stream of ITS group only
    std::map<void *, uint32_t> tag;
    std::vector<hipEvent_t> ev_us2;           // per shard: us2's half of the current tile is up
    int upload_mode = -1;                     // -1 auto (page-locked source -> strided DMA, pageable -> host-packed), 0 packed, 1 strided
    int last_upload = -1;                     // what the last commit did: -1 device source, 0 host-packed staging, 1 strided DMA from the page-locked source
    int pack_threads = 16;
    static constexpr int HS = 3;         // page-locked staging ring of the host uploads (shared by the shards: the host packs one tile at a time)
    u64 *hstage[HS] = {};
    uint64_t hstage_bytes = 0;
    std::vector<hipEvent_t> hstage_sent[HS]; // [ring slot][shard]: an event is recorded on streams of the device it was created on
    int hstage_user[HS] = {-1, -1, -1};      // the shard whose upload last read the slot
    std::vector<Stats> stats;
    double last_wall_ms = 0;
    // what hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess answered per ordered pair of shards (row a, column b: a's device reaching
    // b's memory): 2 same device, 1 peer access enabled, 0 the devices cannot reach each other directly, -1 enabling it failed.  A pair
    // that is not 1 or 2 moves its exchange through host memory (the runtime stages the copies; the kernels that write a peer's row image
    // are replaced by the runtime's 2-D copy across such a pair: rows_2d): `warnings` says so in words, mi_multi_create prints it.
    std::vector<int> peer;
    std::string warnings;
    // mi_multi_lend: a caller that plans a device's HBM (host/starks.hpp: the proof's image fills the device that also is shard 0) hands the
    // next commit a region that is not live; the shard's row buffers, staging and NTT workspace are carved from it instead of allocated
    std::vector<u64 *> lent;
    std::vector<uint64_t> lent_elems;
    // mi_multi_set_row_images: the NEXT commit also leaves, on every shard that names one, the shard's OWN rows of the extension (and the
    // `halo` rows after them, wrapping at n_ext) in row-major form at their place in a full-height section -- what a row-sharded
    // constraint evaluation reads on that device (host/chelpers_steps.hpp)
    std::vector<u64 *> row_img;
    uint64_t row_img_pitch = 0, row_img_halo = 0;
    // device buffers of freed trees, per shard: the next commit takes them back instead of allocating (tens of GB per shard and proof: a fresh
    // allocation is wiped by the driver on the GPU's own bandwidth, DESIGN_HISTORY.md section 6)
    struct Cached { u64 *p; uint64_t elems; };
    std::vector<std::vector<Cached>> pool;
    u64 *take(uint32_t g, uint64_t elems)
    {
        auto &v = pool[g];
        if (getenv("MI_MULTI_NO_POOL")) return nullptr; // (debugging: every buffer a fresh allocation)
        size_t best = v.size();
        for (size_t i = 0; i < v.size(); i++)
            if (v[i].elems >= elems && (best == v.size() || v[i].elems < v[best].elems)) best = i;
        if (best == v.size() || v[best].elems > 2 * elems + (1u << 20)) return nullptr; // nothing fits (or only something wastefully large)
        u64 *p = v[best].p;
        v.erase(v.begin() + best);
        return p;
    }
};

struct mi_multi_tree {
    mi_multi *m = nullptr;
    Plan p;
    std::vector<u64 *> ext, recv, nodes, stage; // per shard, on its device
    u64 *roots = nullptr;                       // shard 0's device: (2G - 1) * 4
    std::vector<u64> roots_host;
    bool keep_rows = true;                      // ext / recv still hold the rows (openings read them)
    std::vector<char> rows_lent;                // per shard: ext / recv / stage live in a lent region (not freed here)
    uint64_t e_ext = 0, e_recv = 0, e_stage = 0, e_nodes = 0; // elements of the buffers (for the pool)
    std::vector<u64 *> gbuf;                    // transient commits: per group leader, [tile ring | staging | NTT workspace] while the commit runs
    uint64_t e_gbuf = 0;
};

#define MM_DEV(m, g) MI_HIP_CHECK(hipSetDevice((m)->dev[g]))

extern "C" void mi_multi_destroy(mi_multi *m);
static int copy_dd(const mi_multi *m, void *dst, int gd, const void *src, int gs, uint64_t bytes, hipStream_t s);

// ---- MI_MULTI_CHECK helpers: events and streams carry the group they were created for
static int mm_own(const mi_multi *m, uint32_t shard, const void *p, const char *what)
{
    (void)m;
    return mi_check_on() ? mi_own_check((int)shard, p, what) : MI_OK;
}
#define MM_OWN(m, g, p, what) MI_TRY(mm_own(m, g, p, what))
static int mm_new_event(mi_multi *m, uint32_t g, hipEvent_t *e, unsigned flags)
{
    MI_HIP_CHECK(hipEventCreateWithFlags(e, flags));
    if (mi_check_on()) m->tag[(void *)*e] = m->lead[g];
    return MI_OK;
}
static void mm_del_event(mi_multi *m, hipEvent_t e)
{
    if (!e) return;
    if (mi_check_on()) m->tag.erase((void *)e);
    (void)hipEventDestroy(e);
}
// hipEventRecord, refused when the event was created for another group than the stream's (on a real node: for another device)
static int mm_record(mi_multi *m, hipEvent_t e, hipStream_t s)
{
    if (mi_check_on()) {
        auto a = m->tag.find((void *)e), b = m->tag.find((void *)s);
        if (a != m->tag.end() && b != m->tag.end() && a->second != b->second) {
            mi_set_error("MI_MULTI_CHECK: an event of shard group %u is recorded on a stream of group %u (on a real multi-GPU node: an event of another device)", a->second, b->second);
            fprintf(stderr, "mi_stark: %s\n", mi_last_error());
            return MI_ERR_INVALID;
        }
    }
    MI_HIP_CHECK(hipEventRecord(e, s));
    return MI_OK;
}
#define MM_RECORD(m, e, s) MI_TRY(mm_record(m, e, s))

extern "C" int mi_multi_create2(mi_multi **out, const int *devices, int n_shards, int group_same_device);
extern "C" int mi_multi_create(mi_multi **out, const int *devices, int n_shards)
{
    const char *e = getenv("MI_MULTI_GROUP_SAME_DEVICE");
    return mi_multi_create2(out, devices, n_shards, e && e[0] == '1');
}

// group_same_device != 0: shards that name the same physical device form one GROUP (see the header of this file): one set of streams
// and buffers for all of them.  What a one-GPU box needs to rehearse many shards at full size; on a node with one shard per device it
// changes nothing.
extern "C" int mi_multi_create2(mi_multi **out, const int *devices, int n_shards, int group_same_device)
{
    MI_REQUIRE(out && devices && n_shards >= 1 && is_pow2((uint64_t)n_shards), "the number of shards must be a power of two");
    MI_REQUIRE(n_shards <= MI_MAX_SLABS, "at most 16 shards (one leaf-hash launch absorbs one column window per shard)");
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have == 0) {
        mi_set_error("no HIP device available: libmi_stark has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    for (int g = 0; g < n_shards; g++)
        if (devices[g] < 0 || devices[g] >= have) {
            mi_set_error("mi_multi_create: device %d out of range (%d devices)", devices[g], have);
            return MI_ERR_INVALID;
        }
    mi_multi *m = new mi_multi();
    m->G = (uint32_t)n_shards;
    m->dev.assign(devices, devices + n_shards);
    m->stats.resize(n_shards);
    m->lent.assign(n_shards, nullptr);
    m->lent_elems.assign(n_shards, 0);
    m->pool.resize(n_shards);
    m->grouped = group_same_device != 0;
    m->lead.resize(n_shards);
    for (int g = 0; g < n_shards; g++) {
        m->lead[g] = (uint32_t)g;
        for (int h = 0; h < g && m->grouped; h++)
            if (devices[h] == devices[g]) { m->lead[g] = (uint32_t)h; break; }
    }
    mi_own_set_leaders(m->lead.data(), (uint32_t)n_shards);
    for (int g = 0; g < n_shards; g++) {
        mi_ctx *c = nullptr;
        int st = mi_ctx_create(&c, devices[g]);
        if (st != MI_OK) { mi_multi_destroy(m); return st; }
        c->logical = g;
        m->ctx.push_back(c);
        const uint32_t L = m->lead[g];
        if (L != (uint32_t)g) { // a grouped shard: its leader's streams and pools
            c->pool = &m->ctx[L]->own_pool;
            m->cs.push_back(m->cs[L]); m->xs.push_back(m->xs[L]); m->us.push_back(m->us[L]); m->us2.push_back(m->us2[L]);
            m->ev_us2.push_back(nullptr);
            for (int i = 0; i < mi_multi::HS; i++) m->hstage_sent[i].push_back(nullptr);
            mi_ctx_set_stream(c, m->cs[L]);
            continue;
        }
        hipStream_t s[4];
        for (int i = 0; i < 4; i++) {
            if (hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) { mi_set_error("mi_multi_create: cannot create a stream"); mi_multi_destroy(m); return MI_ERR_HIP; }
            m->own_streams.push_back(s[i]);
            if (mi_check_on()) m->tag[(void *)s[i]] = (uint32_t)g;
        }
        m->cs.push_back(s[0]); m->xs.push_back(s[1]); m->us.push_back(s[2]); m->us2.push_back(s[3]);
        {
            hipEvent_t e = nullptr;
            if (mm_new_event(m, (uint32_t)g, &e, hipEventDisableTiming) != MI_OK) { mi_set_error("mi_multi_create: cannot create an event"); mi_multi_destroy(m); return MI_ERR_HIP; }
            m->ev_us2.push_back(e);
        }
        mi_ctx_set_stream(c, s[0]);
        for (int i = 0; i < mi_multi::HS; i++) {
            hipEvent_t e = nullptr;
            if (mm_new_event(m, (uint32_t)g, &e, hipEventDisableTiming) != MI_OK) { mi_set_error("mi_multi_create: cannot create an event"); mi_multi_destroy(m); return MI_ERR_HIP; }
            m->hstage_sent[i].push_back(e);
        }
    }
    // every pair of distinct devices talks directly (xGMI): without peer access the runtime would stage the copies through the host.
    // What the driver answers is RECORDED (mi_multi_peer_access) and a missing link is said aloud: a node whose devices cannot reach each
    // other still computes the right root, several times slower, and nobody should have to find that out from a profile.
    m->peer.assign((size_t)n_shards * n_shards, 2);
    for (int a = 0; a < n_shards; a++)
        for (int b = 0; b < n_shards; b++) {
            if (devices[a] == devices[b]) continue;
            int &cell = m->peer[(size_t)a * n_shards + b];
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
            if (e != hipSuccess || !can) {
                (void)hipGetLastError();
                cell = 0;
                char w[160];
                snprintf(w, sizeof w, "device %d cannot access device %d directly (hipDeviceCanAccessPeer: %s); ", devices[a], devices[b], e != hipSuccess ? hipGetErrorString(e) : "no");
                if (m->warnings.find(w) == std::string::npos) m->warnings += w;
                continue;
            }
            (void)hipSetDevice(devices[a]);
            e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); cell = 1; continue; }
            (void)hipGetLastError();
            cell = -1;
            char w[200];
            snprintf(w, sizeof w, "hipDeviceEnablePeerAccess(%d -> %d) failed: %s; ", devices[a], devices[b], hipGetErrorString(e));
            if (m->warnings.find(w) == std::string::npos) m->warnings += w;
        }
    (void)hipGetLastError();
    if (!m->warnings.empty()) {
        m->warnings += "exchanges over such a pair are staged through host memory by the runtime";
        fprintf(stderr, "mi_stark: WARNING (mi_multi_create): %s\n", m->warnings.c_str());
    }
    (void)hipSetDevice(devices[0]);
    // host threads that pack tiles: the cores this process may actually use (a container's CPU quota is not in hardware_concurrency()), at most 64
    unsigned hw = std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = "";
        unsigned long per = 0;
        if (fscanf(f, "%31s %lu", q, &per) == 2 && strcmp(q, "max") != 0 && per) {
            const unsigned long quota = (strtoul(q, nullptr, 10) + per - 1) / per;
            if (quota >= 1 && (!hw || quota < hw)) hw = (unsigned)quota;
        }
        fclose(f);
    }
    m->pack_threads = (int)std::max(1u, std::min(64u, hw ? hw : 16u));
    *out = m;
    return MI_OK;
}

extern "C" void mi_multi_destroy(mi_multi *m)
{
    if (!m) return;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        (void)hipSetDevice(m->dev[g]);
        if (g < m->cs.size()) { (void)hipStreamSynchronize(m->cs[g]); (void)hipStreamSynchronize(m->xs[g]); (void)hipStreamSynchronize(m->us[g]); }
        if (g < m->us2.size()) (void)hipStreamSynchronize(m->us2[g]);
    }
    for (size_t g = 0; g < m->ctx.size(); g++) {
        (void)hipSetDevice(m->dev[g]);
        if (g < m->pool.size()) for (auto &c : m->pool[g]) { mi_own_del(c.p); (void)hipFree(c.p); }
        if (m->ctx[g]) { m->ctx[g]->pool = &m->ctx[g]->own_pool; mi_ctx_set_stream(m->ctx[g], nullptr); mi_ctx_destroy(m->ctx[g]); }
        if (g < m->ev_us2.size() && m->ev_us2[g]) (void)hipEventDestroy(m->ev_us2[g]);
    }
    for (hipStream_t s : m->own_streams) (void)hipStreamDestroy(s); // (a stream is destroyed on any current device)
    for (int i = 0; i < mi_multi::HS; i++) {
        if (m->hstage[i]) (void)hipHostFree(m->hstage[i]);
        for (hipEvent_t e : m->hstage_sent[i]) if (e) (void)hipEventDestroy(e);
    }
    delete m;
}

extern "C" int mi_multi_shards(const mi_multi *m) { return m ? (int)m->G : 0; }
// the first shard of `shard`'s device group (the shard itself unless mi_multi_create2 grouped the shards of one physical device)
extern "C" int mi_multi_lead(const mi_multi *m, int shard) { return (m && shard >= 0 && (uint32_t)shard < m->G) ? (int)m->lead[shard] : -1; }
// The peer-access matrix recorded by mi_multi_create: out[a * G + b] for shard a's device reaching shard b's memory -- 2 same device,
// 1 enabled, 0 no direct access, -1 enabling failed.  Returns the number of pairs that are neither 1 nor 2; warning (optional) receives
// the sentence mi_multi_create printed about them ("" when every pair is direct).
extern "C" int mi_multi_peer_access(const mi_multi *m, int *out, char *warning, uint64_t warning_cap)
{
    if (!m) return -1;
    int bad = 0;
    for (size_t i = 0; i < m->peer.size(); i++) {
        if (out) out[i] = m->peer[i];
        bad += m->peer[i] != 1 && m->peer[i] != 2;
    }
    if (warning && warning_cap) snprintf(warning, warning_cap, "%s", m->warnings.c_str());
    return bad;
}
extern "C" mi_ctx *mi_multi_ctx(mi_multi *m, int shard) { return (m && shard >= 0 && (uint32_t)shard < m->G) ? m->ctx[shard] : nullptr; }
extern "C" int mi_multi_set_pack_threads(mi_multi *m, int threads)
{
    MI_REQUIRE(m && threads >= 1, "bad argument");
    m->pack_threads = threads;
    return MI_OK;
}
// How a HOST source reaches the shards.  -1 (default): a page-locked source (hipHostMalloc / hipHostRegister: mi_host_register) is read by
// each device's own DMA engines as strided 2-D copies straight out of the caller's trace -- no host thread touches the data, G links run
// at once -- when the shards sit on at least two devices; a pageable source (the DMA engines cannot read it) and shards that share one
// device (one link: contiguous copies are faster) are packed by host threads into page-locked staging.  0 / 1 force a form (1 on a
// pageable source is refused by the commit).
extern "C" int mi_multi_set_upload_mode(mi_multi *m, int mode)
{
    MI_REQUIRE(m && mode >= -1 && mode <= 1, "bad argument");
    m->upload_mode = mode;
    return MI_OK;
}
// what the last commit did with its source: -1 device source, 0 host-packed staging, 1 strided DMA from the page-locked source
extern "C" int mi_multi_last_upload_mode(const mi_multi *m) { return m ? m->last_upload : -1; }

// The NEXT commit carves shard `shard`'s row buffers, staging and NTT workspace from [ptr, ptr + bytes) (memory of that shard's device)
// instead of allocating them; the region is the caller's again when the commit returns and mi_multi_tree_release_rows has been called
// (openings with values need the rows: a caller that lends opens the values from its own image).
extern "C" int mi_multi_lend(mi_multi *m, int shard, void *ptr, uint64_t bytes)
{
    MI_REQUIRE(m && shard >= 0 && (uint32_t)shard < m->G, "bad shard");
    m->lent[shard] = (u64 *)ptr;
    m->lent_elems[shard] = ptr ? bytes / 8 : 0;
    return MI_OK;
}

// The NEXT commit also writes shard q's rows [q R, (q + 1) R) of the extension, and the halo_rows rows that follow them (wrapping at
// n_ext: the shifted reads of a constraint program, starks.cpp:240 "(i + next) % NExtended"), row-major at row pitch `pitch` into
// imgs[q] -- the START of a FULL-HEIGHT section (n_ext x pitch) in device memory of shard q's device, of which only those rows are
// written -- for every q with a non-null imgs[q].  With 288 GB a device the full-height layout is affordable and keeps the evaluator's
// addressing that of the one-device image.  One-shot, like mi_multi_lend.
extern "C" int mi_multi_set_row_images(mi_multi *m, uint64_t *const *imgs, uint64_t pitch, uint64_t halo_rows)
{
    MI_REQUIRE(m, "null argument");
    m->row_img.clear();
    if (!imgs) { m->row_img_pitch = m->row_img_halo = 0; return MI_OK; } // (takes the request back)
    m->row_img.assign(m->G, nullptr);
    for (uint32_t g = 0; g < m->G; g++) m->row_img[g] = (u64 *)imgs[g];
    m->row_img_pitch = pitch;
    m->row_img_halo = halo_rows;
    return MI_OK;
}
// the calling thread's current device := shard's device (for a caller that drives mi_multi_ctx(m, shard) itself: kernels are launched on
// the current device).  The caller switches back to its own device (shard 0's, by host/mi_runtime.hpp's convention) afterwards.
extern "C" int mi_multi_set_device(mi_multi *m, int shard)
{
    MI_REQUIRE(m && shard >= 0 && (uint32_t)shard < m->G, "bad shard");
    MM_DEV(m, shard);
    return MI_OK;
}
// bytes from src (device memory of src_shard's device) to dst (of dst_shard's), behind the work queued on src_shard's context
extern "C" int mi_multi_copy(mi_multi *m, void *dst, int dst_shard, const void *src, int src_shard, uint64_t bytes)
{
    MI_REQUIRE(m && dst && src && dst_shard >= 0 && (uint32_t)dst_shard < m->G && src_shard >= 0 && (uint32_t)src_shard < m->G, "bad argument");
    MM_OWN(m, (uint32_t)dst_shard, dst, "mi_multi_copy (dst: memory of dst_shard)");
    MM_OWN(m, (uint32_t)src_shard, src, "mi_multi_copy (src: memory of src_shard)");
    MM_DEV(m, src_shard);
    return copy_dd(m, dst, dst_shard, src, src_shard, bytes, m->cs[src_shard]);
}
extern "C" int mi_multi_sync(mi_multi *m, int shard)
{
    MI_REQUIRE(m && shard >= 0 && (uint32_t)shard < m->G, "bad shard");
    MM_DEV(m, shard);
    MI_HIP_CHECK(hipStreamSynchronize(m->cs[shard]));
    return MI_OK;
}

extern "C" void mi_multi_tree_free(mi_multi_tree *t)
{
    if (!t) return;
    mi_multi *m = t->m;
    for (uint32_t g = 0; g < m->G; g++) {
        (void)hipSetDevice(m->dev[g]);
        (void)hipStreamSynchronize(m->cs[g]); (void)hipStreamSynchronize(m->xs[g]); (void)hipStreamSynchronize(m->us[g]);
        const bool lent = g < t->rows_lent.size() && t->rows_lent[g];
        auto back = [&](u64 *q, uint64_t elems) { if (q) m->pool[g].push_back({q, elems}); };
        if (!lent && g < t->ext.size()) back(t->ext[g], t->e_ext);
        if (!lent && g < t->recv.size()) back(t->recv[g], t->e_recv);
        if (g < t->nodes.size()) back(t->nodes[g], t->e_nodes);
        if (!lent && g < t->stage.size()) back(t->stage[g], t->e_stage);
        if (g < t->gbuf.size() && t->gbuf[g]) { // a transient commit that did not finish: its group's contexts still work in the buffer
            for (uint32_t h = 0; h < m->G; h++) if (m->lead[h] == g) (void)mi_ctx_lend_workspace(m->ctx[h], nullptr, 0);
            if (!lent) back(t->gbuf[g], t->e_gbuf);
        }
    }
    if (t->roots) { (void)hipSetDevice(m->dev[0]); mi_own_del(t->roots); (void)hipFree(t->roots); }
    delete t;
}

// the rows' values are no longer needed from the shards (a caller that keeps the row-major image of the extension opens them there):
// ext / recv / staging go back, the subtrees stay
extern "C" int mi_multi_tree_release_rows(mi_multi_tree *t)
{
    MI_REQUIRE(t, "null tree");
    mi_multi *m = t->m;
    for (uint32_t g = 0; g < m->G; g++) {
        MM_DEV(m, g);
        MI_HIP_CHECK(hipStreamSynchronize(m->cs[g])); MI_HIP_CHECK(hipStreamSynchronize(m->xs[g])); MI_HIP_CHECK(hipStreamSynchronize(m->us[g]));
        if (!t->rows_lent[g]) {
            if (t->ext[g]) m->pool[g].push_back({t->ext[g], t->e_ext});
            if (t->recv[g]) m->pool[g].push_back({t->recv[g], t->e_recv});
            if (t->stage[g]) m->pool[g].push_back({t->stage[g], t->e_stage});
        }
        t->ext[g] = t->recv[g] = t->stage[g] = nullptr;
    }
    t->keep_rows = false;
    return MI_OK;
}

static int copy_dd(const mi_multi *m, void *dst, int gd, const void *src, int gs, uint64_t bytes, hipStream_t s)
{
    if (!bytes) return MI_OK;
    if (m->dev[gd] == m->dev[gs]) MI_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    else MI_HIP_CHECK(hipMemcpyPeerAsync(dst, m->dev[gd], src, m->dev[gs], bytes, s));
    return MI_OK;
}

// rows x w words, 2-D, issued by shard g on stream s: a kernel of g's device when g's device reaches both sides directly (always, inside
// one device; across devices when peer access is on: the recorded matrix), the runtime's 2-D copy -- staged through the host -- otherwise
static int rows_2d(const mi_multi *m, uint32_t g, int dst_dev, int src_dev, u64 *dst, uint64_t dpitch, const u64 *src, uint64_t spitch, uint64_t w, uint64_t rows, hipStream_t s)
{
    if (!w || !rows) return MI_OK;
    auto direct = [&](int d) {
        if (d == m->dev[g]) return true;
        for (uint32_t q = 0; q < m->G; q++)
            if (m->dev[q] == d) return m->peer[(size_t)g * m->G + q] == 1;
        return false; // (a device no shard sits on: nobody asked the driver)
    };
    if (direct(dst_dev) && direct(src_dev)) {
        const uint64_t total = rows * w;
        const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(k_rows_2d, dim3(blocks), dim3(256), 0, s, dst, dpitch, src, spitch, (uint32_t)w, rows);
        MI_HIP_CHECK(hipGetLastError());
    } else {
        MI_HIP_CHECK(hipMemcpy2DAsync(dst, dpitch * 8, src, spitch * 8, w * 8, rows, hipMemcpyDefault, s));
    }
    return MI_OK;
}

// The NEXT commit is TRANSIENT: nobody will open row values from its tree (mi_multi_group_proofs with_values, mi_multi_gather_rows) and
// every shard has a row image (mi_multi_set_row_images with G non-null pointers): the tiles' rows go straight into the images and are
// absorbed there; the tree keeps the subtrees only.  One-shot, like mi_multi_lend.  Without row images for every shard the commit refuses.
extern "C" int mi_multi_set_transient(mi_multi *m, int on)
{
    MI_REQUIRE(m, "null argument");
    m->transient_next = on != 0;
    return MI_OK;
}
// elements of device memory a transient commit of this shape takes per device GROUP (tile ring + staging + NTT workspace): what
// a caller that lends a region to a group's leader must offer (mi_multi_lend) for the commit to allocate nothing
extern "C" uint64_t mi_multi_transient_need(uint64_t n, uint64_t n_ext, uint64_t ncols, uint32_t shards)
{
    Plan p;
    if (!shards || !is_pow2(shards)) return 0;
    p.init(n, n_ext, ncols, shards);
    const uint64_t maxw = *std::max_element(p.round_w.begin(), p.round_w.end());
    const uint64_t al = 31;
    return ((2 * n_ext * maxw + al) & ~al) + ((2 * n * maxw + al) & ~al) + std::max<uint64_t>((2 * n + n_ext) * maxw + 4096, 1ull << 17) + 64;
}

// elements of device memory the WINDOWED (non-transient) commit takes on one shard: extended tiles, row windows, staging, NTT workspace --
// what a region lent to that shard must offer (host/starks.hpp checks it before it shards a stage: 103 GB at 2^23 x 665 and two shards)
extern "C" uint64_t mi_multi_windowed_need(uint64_t n, uint64_t n_ext, uint64_t ncols, uint32_t shards)
{
    Plan p;
    if (!shards || !is_pow2(shards)) return 0;
    p.init(n, n_ext, ncols, shards);
    const uint64_t maxw = *std::max_element(p.round_w.begin(), p.round_w.end());
    const uint64_t e_ext = (n_ext * p.per_rank + 31) & ~31ull, e_recv = shards > 1 ? (p.rows_per_rank * shards * p.per_rank + 31) & ~31ull : 0, e_stage = (2 * n * maxw + 31) & ~31ull;
    return e_ext + e_recv + e_stage + (2 * n + n_ext) * maxw + 4096;
}

namespace {
// what both forms of the commit share
struct CommitEnv {
    mi_multi *m; mi_multi_tree *t;
    const uint64_t *src; uint64_t src_pitch; int src_device; uint32_t src_owner; // src_owner: the shard the device source belongs to
    uint64_t n, n_ext, ncols;
    bool from_host, strided;
    int hs_next = 0;
    std::vector<hipEvent_t> tm; // timing events, destroyed at the end
    struct Timed { uint32_t g; int kind; hipEvent_t a, b; }; // kind 0 lde, 1 absorb, 2 wait
    std::vector<Timed> timed;
    ~CommitEnv() { for (hipEvent_t x : tm) if (x) (void)hipEventDestroy(x); }
    int stamp(hipStream_t s, hipEvent_t *e)
    {
        MI_HIP_CHECK(hipEventCreate(e));
        tm.push_back(*e);
        MI_HIP_CHECK(hipEventRecord(*e, s));
        return MI_OK;
    }
};

// the base-domain columns [c0, c0 + w) of the section onto shard g's device, into `st` ([n x w] at pitch w): enqueued on g's upload
// stream(s) behind `consumed` and `based` (the staging buffer's last readers), `up` is recorded when the tile is there
int upload_tile(CommitEnv &E, uint32_t g, uint64_t w, uint64_t c0, u64 *st, hipEvent_t consumed, hipEvent_t based, hipEvent_t up)
{
    mi_multi *m = E.m;
    const uint64_t n = E.n;
    const uint32_t L = m->lead[g];
    MI_HIP_CHECK(hipStreamWaitEvent(m->us[g], consumed, 0));
    MI_HIP_CHECK(hipStreamWaitEvent(m->us[g], based, 0));
    if (E.from_host && E.strided) {
        // the tile's columns straight out of the caller's page-locked trace: rows of 8 w bytes at the trace's pitch, upper and lower
        // rows on two streams (two DMA engines of THIS device; profiles/r02_pcie_chunk_sweep.json: 39.5 GB/s per link at 32 columns)
        const uint64_t half = n / 2;
        MI_HIP_CHECK(hipStreamWaitEvent(m->us2[g], consumed, 0));
        MI_HIP_CHECK(hipStreamWaitEvent(m->us2[g], based, 0));
        MI_HIP_CHECK(hipMemcpy2DAsync(st, w * 8, E.src + c0, E.src_pitch * 8, w * 8, n - half, hipMemcpyHostToDevice, m->us[g]));
        if (half) MI_HIP_CHECK(hipMemcpy2DAsync(st + (n - half) * w, w * 8, E.src + (n - half) * E.src_pitch + c0, E.src_pitch * 8, w * 8, half, hipMemcpyHostToDevice, m->us2[g]));
        MM_RECORD(m, m->ev_us2[L], m->us2[g]);
        MI_HIP_CHECK(hipStreamWaitEvent(m->us[g], m->ev_us2[L], 0));
    } else if (E.from_host) {
        const int hs = E.hs_next++ % mi_multi::HS;
        const auto t0 = std::chrono::steady_clock::now();
        if (m->hstage_user[hs] >= 0) MI_HIP_CHECK(hipEventSynchronize(m->hstage_sent[hs][m->hstage_user[hs]])); // the copy that last read this slot is done
        u64 *hbuf = m->hstage[hs];
        const uint64_t *src = E.src;
        const uint64_t src_pitch = E.src_pitch;
        const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)m->pack_threads, (n * w * 8) >> 22));
        const uint64_t rows_per = (n + T - 1) / T;
        std::vector<std::thread> th;
        for (int ti = 0; ti < T; ti++) {
            const uint64_t r0 = (uint64_t)ti * rows_per, r1 = std::min(n, r0 + rows_per);
            if (r0 >= r1) break;
            auto work = [=]() {
                const uint64_t *s_ = src + r0 * src_pitch + c0;
                u64 *d_ = hbuf + r0 * w;
                if (w % 8 == 0) { // 64-byte groups: unaligned loads, streaming stores (the staging is read next by the DMA engine, not by this core)
                    for (uint64_t r = r0; r < r1; r++, s_ += src_pitch, d_ += w)
                        for (uint64_t j = 0; j < w; j += 8) {
                            const __m128i v0 = _mm_loadu_si128((const __m128i *)(s_ + j)), v1 = _mm_loadu_si128((const __m128i *)(s_ + j + 2));
                            const __m128i v2 = _mm_loadu_si128((const __m128i *)(s_ + j + 4)), v3 = _mm_loadu_si128((const __m128i *)(s_ + j + 6));
                            _mm_stream_si128((__m128i *)(d_ + j), v0); _mm_stream_si128((__m128i *)(d_ + j + 2), v1);
                            _mm_stream_si128((__m128i *)(d_ + j + 4), v2); _mm_stream_si128((__m128i *)(d_ + j + 6), v3);
                        }
                    _mm_sfence();
                } else {
                    for (uint64_t r = r0; r < r1; r++, s_ += src_pitch, d_ += w) memcpy(d_, s_, w * 8);
                }
            };
            if (T == 1) work(); else th.emplace_back(work);
        }
        for (auto &x : th) x.join();
        m->stats[g].upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        MI_HIP_CHECK(hipMemcpyAsync(st, hbuf, n * w * 8, hipMemcpyHostToDevice, m->us[g]));
        MM_RECORD(m, m->hstage_sent[hs][L], m->us[g]);
        m->hstage_user[hs] = (int)L;
    } else {
        // a device source (stages 2-4: a section of the proof's image): read by a kernel of THIS shard's device, across xGMI when the
        // section lives on another one -- a declared peer operand
        MM_OWN(m, E.src_owner, E.src + c0, "mi_multi_commit (the device source: memory of the shard that holds the section)");
        MI_TRY(rows_2d(m, g, m->dev[g], E.src_device, st, w, (const u64 *)E.src + c0, E.src_pitch, w, n, m->us[g]));
    }
    MM_RECORD(m, up, m->us[g]);
    return MI_OK;
}
} // namespace

static int commit_transient(CommitEnv &E, uint64_t *base, uint64_t base_pitch, uint32_t base_owner);

// Stage commit.  src: the n x ncols row-major base-domain section at row pitch src_pitch (elements) -- in HOST memory (src_device < 0;
// pageable is fine, host threads read it) or on device `src_device` (the image of a proof: stages 2-4).  image / base (optional, on
// device image_device): receive the whole extension (n_ext x ncols at image_pitch) / the section itself (n x ncols at base_pitch) in
// row-major form.  root: 4 words, host.
extern "C" int mi_multi_commit(mi_multi *m, mi_multi_tree **out, const uint64_t *src, uint64_t src_pitch, int src_device, uint64_t n, uint64_t n_ext,
                               uint64_t ncols, uint64_t *image, uint64_t image_pitch, uint64_t *base, uint64_t base_pitch, int image_device, uint64_t root[4])
{
    MI_REQUIRE(m, "null argument");
    // mi_multi_lend, mi_multi_set_row_images and mi_multi_set_transient arm ONE commit: they are disarmed however this call ends, also by
    // a refused argument (a later commit, possibly of another stage, must not carve its buffers out of a region its caller considers
    // live again)
    struct RowImgGuard { mi_multi *m; ~RowImgGuard() { m->row_img.clear(); m->row_img_pitch = m->row_img_halo = 0; m->transient_next = false; } } rowimgguard{m};
    struct LendGuard { mi_multi *m; ~LendGuard() { for (uint32_t g = 0; g < m->G; g++) if (m->lent[g]) { (void)hipSetDevice(m->dev[g]); for (uint32_t h = 0; h < m->G; h++) if (m->lead[h] == m->lead[g]) (void)mi_ctx_lend_workspace(m->ctx[h], nullptr, 0); m->lent[g] = nullptr; m->lent_elems[g] = 0; } } } lendguard{m};
    MI_REQUIRE(out && src && root, "null argument");
    MI_REQUIRE(is_pow2(n) && is_pow2(n_ext) && n_ext >= n && ncols > 4, "sizes: powers of two, more than 4 columns (linear_hash copies shorter rows)");
    MI_REQUIRE(n_ext % m->G == 0 && n_ext / m->G >= 2, "too few rows for this many shards");
    const auto t_begin = std::chrono::steady_clock::now();
    const uint32_t G = m->G;
    int image_owner = -1; // the shard image / base belong to: the first one on image_device
    for (uint32_t g = 0; g < G && image_owner < 0; g++) if (m->dev[g] == image_device) image_owner = (int)g;
    MI_REQUIRE(!(image || base) || image_owner >= 0, "image_device is not the device of any shard");
    int src_owner = 0;
    if (src_device >= 0) {
        src_owner = -1;
        for (uint32_t g = 0; g < G && src_owner < 0; g++) if (m->dev[g] == src_device) src_owner = (int)g;
        MI_REQUIRE(src_owner >= 0, "src_device is not the device of any shard");
    }
    mi_multi_tree *t = new mi_multi_tree();
    struct Guard { mi_multi_tree *t; ~Guard() { if (t) mi_multi_tree_free(t); } } guard{t};
    t->m = m;
    Plan &p = t->p;
    p.init(n, n_ext, ncols, G);
    const size_t R = p.rounds();
    const uint64_t maxw = *std::max_element(p.round_w.begin(), p.round_w.end());
    constexpr int NS = 2; // device staging buffers per shard
    t->ext.assign(G, nullptr); t->recv.assign(G, nullptr); t->nodes.assign(G, nullptr); t->stage.assign(G, nullptr); t->gbuf.assign(G, nullptr);
    t->rows_lent.assign(G, 0);
    MI_REQUIRE(m->row_img.empty() || (m->row_img_pitch >= ncols && m->row_img_halo <= n_ext / m->G), "row images: pitch smaller than ncols or halo larger than a shard");
    if (mi_check_on())
        for (uint32_t q = 0; q < G && !m->row_img.empty(); q++)
            if (m->row_img[q]) MM_OWN(m, q, m->row_img[q] + (uint64_t)q * p.rows_per_rank * m->row_img_pitch, "mi_multi_set_row_images (imgs[q] is memory of shard q)");
    const bool transient = m->transient_next;
    if (transient) {
        MI_REQUIRE(!m->row_img.empty() && !image, "a transient commit writes row images only (mi_multi_set_row_images for every shard, no whole-extension image)");
        for (uint32_t q = 0; q < G; q++) MI_REQUIRE(m->row_img[q], "a transient commit needs a row image for EVERY shard");
    }
    CommitEnv E{m, t, src, src_pitch, src_device, (uint32_t)src_owner, n, n_ext, ncols, src_device < 0, false};
    for (uint32_t g = 0; g < G; g++) {
        m->stats[g] = Stats();
        m->stats[g].bytes_to.assign(G, 0);
    }
    if (E.from_host) {
        hipPointerAttribute_t at;
        const uint64_t last = (n - 1) * src_pitch + ncols - 1;
        bool locked = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
        if (locked) { hipPointerAttribute_t a2; locked = hipPointerGetAttributes(&a2, src + last) == hipSuccess && a2.type == hipMemoryTypeHost; }
        (void)hipGetLastError(); // a pageable pointer is "invalid value" to the query, not an error of this call
        int mode = m->upload_mode;
        if (const char *e = getenv("MI_MULTI_UPLOAD")) mode = !strcmp(e, "packed") ? 0 : !strcmp(e, "strided") ? 1 : mode;
        MI_REQUIRE(mode != 1 || locked, "strided upload asked for a source that is not page-locked (mi_host_register it, or leave the mode at -1)");
        // auto: in place when the shards sit on at least two devices -- G links at the strided rate (39.5 GB/s each at 32 columns) beat one
        // host's packing threads (55 GB/s on 16 cores out of a page-locked trace, measured); logical shards on ONE device share one link,
        // where the packed form's contiguous copies are the faster (1.07 s against 1.40 s per zkEVM step: profiles/r04_sp4_upload_ab.json)
        bool two_devices = false;
        for (uint32_t g = 1; g < G; g++) two_devices = two_devices || m->dev[g] != m->dev[0];
        E.strided = mode == 1 || (mode == -1 && locked && two_devices);
    }
    m->last_upload = E.from_host ? (E.strided ? 1 : 0) : -1;
    if (E.from_host && !E.strided) {
        const uint64_t need = n * maxw * 8;
        if (m->hstage_bytes < need) {
            for (int i = 0; i < mi_multi::HS; i++) {
                if (m->hstage_user[i] >= 0) MI_HIP_CHECK(hipEventSynchronize(m->hstage_sent[i][m->hstage_user[i]]));
                m->hstage_user[i] = -1;
                if (m->hstage[i]) MI_HIP_CHECK(hipHostFree(m->hstage[i]));
                m->hstage[i] = nullptr;
                MI_HIP_CHECK(hipHostMalloc((void **)&m->hstage[i], need, hipHostMallocPortable));
            }
            m->hstage_bytes = need;
        }
    }
    auto alloc = [&](uint32_t g, u64 **q, uint64_t elems, const char *what) -> int {
        elems = std::max<uint64_t>(elems, 1);
        if ((*q = m->take(g, elems))) return MI_OK;
        hipError_t e = hipMalloc((void **)q, elems * 8);
        if (e != hipSuccess && !m->pool[g].empty()) { // make room: give the cached buffers back and try once more
            (void)hipGetLastError();
            for (auto &c : m->pool[g]) { mi_own_del(c.p); (void)hipFree(c.p); }
            m->pool[g].clear();
            e = hipMalloc((void **)q, elems * 8);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); mi_set_error("mi_multi_commit: shard %u cannot allocate %.2f GB (%s): %s", g, elems * 8 / 1e9, what, hipGetErrorString(e)); return MI_ERR_NOMEM; }
        mi_own_add(*q, elems * 8, (int)g, what);
        return MI_OK;
    };
    for (uint32_t g = 0; g < G; g++) {
        MM_DEV(m, g);
        t->e_nodes = std::max<uint64_t>((2 * p.rows_per_rank - 1) * 4, 1);
        MI_TRY(alloc(g, &t->nodes[g], (2 * p.rows_per_rank - 1) * 4, "subtree nodes"));
    }
    MM_DEV(m, 0);
    MI_HIP_CHECK(hipMalloc((void **)&t->roots, (2 * G - 1) * 4 * 8));
    mi_own_add(t->roots, (2 * G - 1) * 4 * 8, 0, "roots of the subtrees");

    if (transient) {
        // ---- buffers per GROUP: [tile ring | staging ring | NTT workspace], lent to the leader or from its pool
        const uint64_t e_ring = (2 * n_ext * maxw + 31) & ~31ull, e_stage = (NS * n * maxw + 31) & ~31ull;
        const uint64_t e_ws = std::max<uint64_t>((2 * n + n_ext) * maxw + 4096, 1ull << 17); // (a lent workspace is at least 1 MiB)
        t->e_gbuf = e_ring + e_stage + e_ws;
        t->keep_rows = false;
        for (uint32_t L = 0; L < G; L++) {
            if (m->lead[L] != L) continue;
            MM_DEV(m, L);
            if (m->lent[L] && m->lent_elems[L] >= t->e_gbuf) { t->gbuf[L] = m->lent[L]; t->rows_lent[L] = 1; MM_OWN(m, L, m->lent[L], "mi_multi_lend (a region of the leader shard's device)"); }
            else MI_TRY(alloc(L, &t->gbuf[L], t->e_gbuf, "tile ring, staging and NTT workspace of a transient commit"));
            for (uint32_t h = 0; h < G; h++)
                if (m->lead[h] == L) MI_TRY(mi_ctx_lend_workspace(m->ctx[h], t->gbuf[L] + e_ring + e_stage, e_ws * 8));
        }
        MI_TRY(commit_transient(E, base, base_pitch, (uint32_t)std::max(image_owner, 0)));
    } else {
    for (uint32_t g = 0; g < G; g++) {
        MM_DEV(m, g);
        const uint64_t e_ext = (n_ext * p.per_rank + 31) & ~31ull, e_recv = G > 1 ? (p.rows_per_rank * G * p.per_rank + 31) & ~31ull : 0, e_stage = (NS * n * maxw + 31) & ~31ull;
        t->e_ext = std::max<uint64_t>(n_ext * p.per_rank, 1); t->e_recv = std::max<uint64_t>(p.rows_per_rank * G * p.per_rank, 1); t->e_stage = std::max<uint64_t>(NS * n * maxw, 1);
        const uint64_t e_ws = (2 * n + n_ext) * maxw + 4096; // the tile's transforms (launch_lde into a compact window: INTT intermediate + one n_ext ping-pong buffer [+ coefficients when the middle pass is not fused])
        if (m->lent[g] && m->lent_elems[g] >= e_ext + e_recv + e_stage + e_ws) { // the row buffers, the staging and the transforms' workspace out of the lent region
            u64 *q = m->lent[g];
            MM_OWN(m, g, q, "mi_multi_lend (a region of that shard's device)");
            t->ext[g] = q; q += e_ext;
            if (G > 1) { t->recv[g] = q; q += e_recv; }
            t->stage[g] = q; q += e_stage;
            MI_TRY(mi_ctx_lend_workspace(m->ctx[g], q, (m->lent_elems[g] - (uint64_t)(q - m->lent[g])) * 8));
            t->rows_lent[g] = 1;
        } else {
            if (m->lent[g])
                fprintf(stderr, "mi_stark: mi_multi_commit: the region lent to shard %u (%.1f GB) is smaller than the %.1f GB its row buffers, staging and workspace take at %u shards: allocating them instead\n",
                        g, m->lent_elems[g] * 8 / 1e9, (e_ext + e_recv + e_stage + e_ws) * 8 / 1e9, G);
            MI_TRY(alloc(g, &t->ext[g], n_ext * p.per_rank, "extended column tiles"));
            if (G > 1) MI_TRY(alloc(g, &t->recv[g], p.rows_per_rank * G * p.per_rank, "row windows"));
            MI_TRY(alloc(g, &t->stage[g], NS * n * maxw, "tile staging"));
        }
    }
    // events: per shard and round
    std::vector<std::vector<hipEvent_t>> ev_lde(G), ev_sent(G);
    std::vector<hipEvent_t> ev_up(G * NS), ev_consumed(G * NS), ev_based(G * NS);
    struct EvGuard { mi_multi *m; std::vector<std::vector<hipEvent_t>> *a, *b; std::vector<hipEvent_t> *c, *d, *e;
                     ~EvGuard() { for (auto *vv : {a, b}) for (auto &v : *vv) for (hipEvent_t x : v) mm_del_event(m, x);
                                  for (auto *v : {c, d, e}) for (hipEvent_t x : *v) mm_del_event(m, x); } } evguard{m, &ev_lde, &ev_sent, &ev_up, &ev_consumed, &ev_based};
    for (uint32_t g = 0; g < G; g++) {
        MM_DEV(m, g);
        ev_lde[g].assign(R, nullptr); ev_sent[g].assign(R, nullptr);
        for (size_t k = 0; k < R; k++) {
            MI_TRY(mm_new_event(m, g, &ev_lde[g][k], hipEventDisableTiming));
            MI_TRY(mm_new_event(m, g, &ev_sent[g][k], hipEventDisableTiming));
        }
        for (int s = 0; s < NS; s++) {
            MI_TRY(mm_new_event(m, g, &ev_up[g * NS + s], hipEventDisableTiming));
            MI_TRY(mm_new_event(m, g, &ev_consumed[g * NS + s], hipEventDisableTiming));
            MI_TRY(mm_new_event(m, g, &ev_based[g * NS + s], hipEventDisableTiming));
            MM_RECORD(m, ev_consumed[g * NS + s], m->cs[g]);
            MM_RECORD(m, ev_based[g * NS + s], m->xs[g]);
        }
    }
    std::vector<int> slot_of(G, 0);
    auto absorb_round = [&](size_t k) -> int {
        for (uint32_t q = 0; q < G; q++) {
            MM_DEV(m, q);
            hipEvent_t w0, w1, a0, a1;
            MI_TRY(E.stamp(m->cs[q], &w0));
            for (uint32_t g = 0; g < G; g++)
                if (g != q && p.width(k, g)) MI_HIP_CHECK(hipStreamWaitEvent(m->cs[q], ev_sent[g][k], 0));
            MI_TRY(E.stamp(m->cs[q], &w1));
            E.timed.push_back({q, 2, w0, w1});
            const u64 *bases[MI_MAX_SLABS];
            uint64_t pitches[MI_MAX_SLABS], widths[MI_MAX_SLABS];
            uint32_t nw = 0;
            for (uint32_t g = 0; g < G; g++) {
                const uint64_t w = p.width(k, g);
                if (!w) continue;
                MI_REQUIRE(nw < MI_MAX_SLABS, "more shards than one absorb launch takes windows");
                bases[nw] = g == q ? t->ext[q] + p.ext_base(k) + (uint64_t)q * p.rows_per_rank * w : t->recv[q] + p.recv_off(k, g);
                MM_OWN(m, q, bases[nw], "mi_multi_commit (absorb: a column window of this shard's rows)");
                pitches[nw] = w; widths[nw] = w;
                nw++;
            }
            if (!nw) continue;
            MM_OWN(m, q, t->nodes[q], "mi_multi_commit (absorb: this shard's leaf digests)");
            MI_TRY(E.stamp(m->cs[q], &a0));
            MI_TRY(launch_linear_hash_absorb(m->ctx[q], t->nodes[q], nw, bases, pitches, widths, p.rows_per_rank, k == 0, k + 1 == R));
            MI_TRY(E.stamp(m->cs[q], &a1));
            E.timed.push_back({q, 1, a0, a1});
        }
        return MI_OK;
    };
    for (size_t k = 0; k < R; k++) {
        for (uint32_t g = 0; g < G; g++) {
            const uint64_t w = p.width(k, g), c0 = p.c0(k, g);
            if (!w) continue;
            MM_DEV(m, g);
            const int slot = slot_of[g]++ % NS;
            u64 *st = t->stage[g] + (uint64_t)slot * n * maxw;
            MM_OWN(m, g, st, "mi_multi_commit (this shard's tile staging)");
            // ---- the tile's base-domain columns onto shard g's device
            MI_TRY(upload_tile(E, g, w, c0, st, ev_consumed[g * NS + slot], ev_based[g * NS + slot], ev_up[g * NS + slot]));
            // ---- LDE of the tile
            MI_HIP_CHECK(hipStreamWaitEvent(m->cs[g], ev_up[g * NS + slot], 0));
            hipEvent_t l0, l1;
            MI_TRY(E.stamp(m->cs[g], &l0));
            MM_OWN(m, g, t->ext[g] + p.ext_base(k), "mi_multi_commit (this shard's extended tiles)");
            MI_TRY(launch_lde(m->ctx[g], t->ext[g] + p.ext_base(k), w, st, w, n_ext, n, w));
            MI_TRY(E.stamp(m->cs[g], &l1));
            E.timed.push_back({g, 0, l0, l1});
            MM_RECORD(m, ev_consumed[g * NS + slot], m->cs[g]);
            MM_RECORD(m, ev_lde[g][k], m->cs[g]);
            // ---- exchange stream: the section itself and the whole extended tile into the row-major image (if asked), my tile's rows to their owners
            if (base) {
                MI_HIP_CHECK(hipStreamWaitEvent(m->xs[g], ev_up[g * NS + slot], 0));
                MM_OWN(m, (uint32_t)image_owner, base + c0, "mi_multi_commit (base: memory of the shard on image_device)");
                MI_TRY(rows_2d(m, g, image_device, m->dev[g], (u64 *)base + c0, base_pitch, st, w, w, n, m->xs[g]));
            }
            MM_RECORD(m, ev_based[g * NS + slot], m->xs[g]);
            MI_HIP_CHECK(hipStreamWaitEvent(m->xs[g], ev_lde[g][k], 0));
            for (uint32_t q = 0; q < G; q++) {
                if (q == g) continue;
                const uint64_t cnt = p.rows_per_rank * w;
                MM_OWN(m, q, t->recv[q] + p.recv_off(k, g), "mi_multi_commit (exchange: the receiving shard's row windows)");
                MI_TRY(copy_dd(m, t->recv[q] + p.recv_off(k, g), (int)q, t->ext[g] + p.ext_base(k) + (uint64_t)q * cnt, (int)g, cnt * 8, m->xs[g]));
                m->stats[g].bytes_to[q] += cnt * 8;
            }
            if (image) {
                MM_OWN(m, (uint32_t)image_owner, image + c0, "mi_multi_commit (image: memory of the shard on image_device)");
                MI_TRY(rows_2d(m, g, image_device, m->dev[g], (u64 *)image + c0, image_pitch, t->ext[g] + p.ext_base(k), w, w, n_ext, m->xs[g]));
            }
            for (uint32_t q = 0; q < G && !m->row_img.empty(); q++) { // shard q's own rows (+ halo) of the tile, row-major, into its full-height section
                u64 *ri = m->row_img[q];
                if (!ri) continue;
                const u64 *tile = t->ext[g] + p.ext_base(k);
                const uint64_t r0 = (uint64_t)q * p.rows_per_rank, r1 = (r0 + p.rows_per_rank) % n_ext, pt = m->row_img_pitch;
                MI_TRY(rows_2d(m, g, m->dev[q], m->dev[g], ri + r0 * pt + c0, pt, tile + r0 * w, w, w, p.rows_per_rank, m->xs[g]));
                if (m->row_img_halo) MI_TRY(rows_2d(m, g, m->dev[q], m->dev[g], ri + r1 * pt + c0, pt, tile + r1 * w, w, w, m->row_img_halo, m->xs[g]));
                m->stats[g].bytes_to[q] += q == g ? 0 : (p.rows_per_rank + m->row_img_halo) * w * 8;
            }
            MM_RECORD(m, ev_sent[g][k], m->xs[g]);
        }
        if (k > 0) MI_TRY(absorb_round(k - 1));
    }
    MI_TRY(absorb_round(R - 1));
    } // (the windowed form)
    // ---- subtrees, then the G roots meet on shard 0
    std::vector<hipEvent_t> ev_root(G, nullptr);
    struct RootEvGuard { mi_multi *m; std::vector<hipEvent_t> *v; ~RootEvGuard() { for (hipEvent_t x : *v) mm_del_event(m, x); } } rootguard{m, &ev_root};
    for (uint32_t g = 0; g < G; g++) {
        MM_DEV(m, g);
        MI_TRY(launch_merkle_levels(m->ctx[g], t->nodes[g], p.rows_per_rank));
        MI_TRY(mm_new_event(m, g, &ev_root[g], hipEventDisableTiming));
        MM_RECORD(m, ev_root[g], m->cs[g]);
        MI_HIP_CHECK(hipStreamWaitEvent(m->xs[g], ev_root[g], 0));
        MI_TRY(copy_dd(m, t->roots + 4 * g, 0, t->nodes[g] + (2 * p.rows_per_rank - 2) * 4, (int)g, 32, m->xs[g]));
        MM_RECORD(m, ev_root[g], m->xs[g]);
    }
    MM_DEV(m, 0);
    for (uint32_t g = 0; g < G; g++) MI_HIP_CHECK(hipStreamWaitEvent(m->cs[0], ev_root[g], 0));
    if (G > 1) MI_TRY(launch_merkle_levels(m->ctx[0], t->roots, G));
    t->roots_host.assign((2 * G - 1) * 4, 0);
    MI_HIP_CHECK(hipMemcpyAsync(t->roots_host.data(), t->roots, (2 * G - 1) * 32, hipMemcpyDeviceToHost, m->cs[0]));
    for (uint32_t g = 0; g < G; g++) {
        MM_DEV(m, g);
        MI_HIP_CHECK(hipStreamSynchronize(m->us[g])); MI_HIP_CHECK(hipStreamSynchronize(m->us2[g])); MI_HIP_CHECK(hipStreamSynchronize(m->xs[g])); MI_HIP_CHECK(hipStreamSynchronize(m->cs[g]));
    }
    for (int i = 0; i < 4; i++) root[i] = t->roots_host[(2 * G - 2) * 4 + i];
    for (const CommitEnv::Timed &x : E.timed) {
        float ms = 0;
        (void)hipSetDevice(m->dev[x.g]);
        if (hipEventElapsedTime(&ms, x.a, x.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        (x.kind == 0 ? m->stats[x.g].lde_ms : x.kind == 1 ? m->stats[x.g].absorb_ms : m->stats[x.g].wait_ms) += ms;
    }
    if (transient) { // the group buffers served the commit only: back to their pool (a lent region is its owner's again when this returns)
        for (uint32_t L = 0; L < G; L++) {
            if (m->lead[L] != L || !t->gbuf[L]) continue;
            MM_DEV(m, L);
            for (uint32_t h = 0; h < G; h++) if (m->lead[h] == L) MI_TRY(mi_ctx_lend_workspace(m->ctx[h], nullptr, 0));
            if (!t->rows_lent[L]) m->pool[L].push_back({t->gbuf[L], t->e_gbuf});
            t->gbuf[L] = nullptr;
        }
    }
    MM_DEV(m, 0);
    m->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    guard.t = nullptr;
    *out = t;
    return MI_OK;
}

// The transient form (see the header of this file).  Per round and shard g of group L, on the group's streams:
//     upload stream    the tile's base-domain columns into the group's staging ring
//     compute stream   LDE into a slot of the group's TILE RING (two slots: the slot's previous tile has left for the row images)
//     exchange stream  the section itself into `base` (optional); for EVERY shard q -- g included -- a kernel writes the tile's rows
//                      [q R, (q + 1) R) and the halo after them (wrapping at n_ext) into q's row image at the tile's columns: one
//                      write per element, 256-byte runs, across xGMI for q on another device
// then shard q absorbs the round's columns of its rows WHERE THEY LANDED: windows of its row image, at the image's pitch.
static int commit_transient(CommitEnv &E, uint64_t *base, uint64_t base_pitch, uint32_t base_owner)
{
    mi_multi *m = E.m;
    mi_multi_tree *t = E.t;
    const Plan &p = t->p;
    const uint32_t G = m->G;
    const uint64_t n = E.n, n_ext = E.n_ext;
    const size_t R = p.rounds();
    const uint64_t maxw = *std::max_element(p.round_w.begin(), p.round_w.end());
    constexpr int NS = 2, NE = 2;
    const uint64_t e_ring = (2 * n_ext * maxw + 31) & ~31ull;
    const uint64_t pt = m->row_img_pitch, halo = m->row_img_halo, Rr = p.rows_per_rank;
    // events per GROUP (staging slots, ring slots) and per shard and round (tile sent)
    std::vector<hipEvent_t> ev_up(G * NS, nullptr), ev_consumed(G * NS, nullptr), ev_based(G * NS, nullptr), ev_free(G * NE, nullptr), ev_lde(G, nullptr);
    std::vector<std::vector<hipEvent_t>> ev_sent(G);
    struct EvGuard { mi_multi *m; std::vector<std::vector<hipEvent_t>> *a; std::vector<hipEvent_t> *v[5];
                     ~EvGuard() { for (auto &x : *a) for (hipEvent_t e : x) mm_del_event(m, e); for (auto *w : v) for (hipEvent_t e : *w) mm_del_event(m, e); } }
        evguard{m, &ev_sent, {&ev_up, &ev_consumed, &ev_based, &ev_free, &ev_lde}};
    for (uint32_t g = 0; g < G; g++) {
        MM_DEV(m, g);
        ev_sent[g].assign(R, nullptr);
        for (size_t k = 0; k < R; k++) MI_TRY(mm_new_event(m, g, &ev_sent[g][k], hipEventDisableTiming));
        if (m->lead[g] != g) continue;
        MI_TRY(mm_new_event(m, g, &ev_lde[g], hipEventDisableTiming));
        for (int s = 0; s < NS; s++) {
            MI_TRY(mm_new_event(m, g, &ev_up[g * NS + s], hipEventDisableTiming));
            MI_TRY(mm_new_event(m, g, &ev_consumed[g * NS + s], hipEventDisableTiming));
            MI_TRY(mm_new_event(m, g, &ev_based[g * NS + s], hipEventDisableTiming));
            MM_RECORD(m, ev_consumed[g * NS + s], m->cs[g]);
            MM_RECORD(m, ev_based[g * NS + s], m->xs[g]);
        }
        for (int s = 0; s < NE; s++) {
            MI_TRY(mm_new_event(m, g, &ev_free[g * NE + s], hipEventDisableTiming));
            MM_RECORD(m, ev_free[g * NE + s], m->xs[g]);
        }
    }
    std::vector<int> slot_of(G, 0), ring_of(G, 0);
    auto absorb_round = [&](size_t k) -> int {
        for (uint32_t q = 0; q < G; q++) {
            MM_DEV(m, q);
            hipEvent_t w0, w1, a0, a1;
            MI_TRY(E.stamp(m->cs[q], &w0));
            for (uint32_t g = 0; g < G; g++)
                if (p.width(k, g)) MI_HIP_CHECK(hipStreamWaitEvent(m->cs[q], ev_sent[g][k], 0)); // (g == q too: my own rows went through the exchange stream)
            MI_TRY(E.stamp(m->cs[q], &w1));
            E.timed.push_back({q, 2, w0, w1});
            const u64 *bases[MI_MAX_SLABS];
            uint64_t pitches[MI_MAX_SLABS], widths[MI_MAX_SLABS];
            uint32_t nw = 0;
            for (uint32_t g = 0; g < G; g++) {
                const uint64_t w = p.width(k, g);
                if (!w) continue;
                MI_REQUIRE(nw < MI_MAX_SLABS, "more shards than one absorb launch takes windows");
                bases[nw] = m->row_img[q] + (uint64_t)q * Rr * pt + p.c0(k, g);
                MM_OWN(m, q, bases[nw], "mi_multi_commit (absorb: this shard's row image)");
                pitches[nw] = pt; widths[nw] = w;
                nw++;
            }
            if (!nw) continue;
            MM_OWN(m, q, t->nodes[q], "mi_multi_commit (absorb: this shard's leaf digests)");
            MI_TRY(E.stamp(m->cs[q], &a0));
            MI_TRY(launch_linear_hash_absorb(m->ctx[q], t->nodes[q], nw, bases, pitches, widths, Rr, k == 0, k + 1 == R));
            MI_TRY(E.stamp(m->cs[q], &a1));
            E.timed.push_back({q, 1, a0, a1});
        }
        return MI_OK;
    };
    for (size_t k = 0; k < R; k++) {
        for (uint32_t g = 0; g < G; g++) {
            const uint64_t w = p.width(k, g), c0 = p.c0(k, g);
            if (!w) continue;
            MM_DEV(m, g);
            const uint32_t L = m->lead[g];
            const int slot = slot_of[L]++ % NS, rs = ring_of[L]++ % NE;
            u64 *st = t->gbuf[L] + e_ring + (uint64_t)slot * n * maxw;
            u64 *tile = t->gbuf[L] + (uint64_t)rs * n_ext * maxw;
            MM_OWN(m, L, st, "mi_multi_commit (the group's tile staging)");
            MM_OWN(m, L, tile, "mi_multi_commit (the group's tile ring)");
            MI_TRY(upload_tile(E, g, w, c0, st, ev_consumed[L * NS + slot], ev_based[L * NS + slot], ev_up[L * NS + slot]));
            // ---- LDE of the tile into the ring slot, once the slot's previous tile has left
            MI_HIP_CHECK(hipStreamWaitEvent(m->cs[g], ev_up[L * NS + slot], 0));
            MI_HIP_CHECK(hipStreamWaitEvent(m->cs[g], ev_free[L * NE + rs], 0));
            hipEvent_t l0, l1;
            MI_TRY(E.stamp(m->cs[g], &l0));
            MI_TRY(launch_lde(m->ctx[g], tile, w, st, w, n_ext, n, w));
            MI_TRY(E.stamp(m->cs[g], &l1));
            E.timed.push_back({g, 0, l0, l1});
            MM_RECORD(m, ev_consumed[L * NS + slot], m->cs[g]);
            MM_RECORD(m, ev_lde[L], m->cs[g]);
            // ---- exchange stream
            if (base) {
                MI_HIP_CHECK(hipStreamWaitEvent(m->xs[g], ev_up[L * NS + slot], 0));
                MM_OWN(m, base_owner, base + c0, "mi_multi_commit (base: memory of the shard on image_device)");
                MI_TRY(rows_2d(m, g, m->dev[base_owner], m->dev[g], (u64 *)base + c0, base_pitch, st, w, w, n, m->xs[g]));
            }
            MM_RECORD(m, ev_based[L * NS + slot], m->xs[g]);
            MI_HIP_CHECK(hipStreamWaitEvent(m->xs[g], ev_lde[L], 0));
            for (uint32_t q = 0; q < G; q++) { // a declared peer operand: q's row image, written by g's device
                u64 *ri = m->row_img[q];
                const uint64_t r0 = (uint64_t)q * Rr, r1 = (r0 + Rr) % n_ext;
                MM_OWN(m, q, ri + r0 * pt + c0, "mi_multi_commit (a shard's row image)");
                if (r1 != 0 || !halo) MI_TRY(rows_2d(m, g, m->dev[q], m->dev[g], ri + r0 * pt + c0, pt, tile + r0 * w, w, w, Rr + (r1 ? halo : 0), m->xs[g])); // rows and halo in one piece
                else { // the last shard: its halo is the extension's first rows
                    MI_TRY(rows_2d(m, g, m->dev[q], m->dev[g], ri + r0 * pt + c0, pt, tile + r0 * w, w, w, Rr, m->xs[g]));
                    MI_TRY(rows_2d(m, g, m->dev[q], m->dev[g], ri + c0, pt, tile, w, w, halo, m->xs[g]));
                }
                m->stats[g].bytes_to[q] += m->dev[q] == m->dev[g] ? 0 : (Rr + halo) * w * 8;
            }
            MM_RECORD(m, ev_sent[g][k], m->xs[g]);
            MM_RECORD(m, ev_free[L * NE + rs], m->xs[g]);
        }
        if (k > 0) MI_TRY(absorb_round(k - 1));
    }
    MI_TRY(absorb_round(R - 1));
    return MI_OK;
}

// per shard of the last commit: [lde_ms, absorb_ms, exchange_wait_ms, host_pack_ms, bytes sent to shard 0 .. G-1]  (4 + G doubles each)
extern "C" int mi_multi_last_stats(const mi_multi *m, double *out, double *wall_ms)
{
    MI_REQUIRE(m && out, "null argument");
    for (uint32_t g = 0; g < m->G; g++) {
        double *o = out + (size_t)g * (4 + m->G);
        o[0] = m->stats[g].lde_ms; o[1] = m->stats[g].absorb_ms; o[2] = m->stats[g].wait_ms; o[3] = m->stats[g].upload_ms;
        for (uint32_t q = 0; q < m->G; q++) o[4 + q] = q < m->stats[g].bytes_to.size() ? (double)m->stats[g].bytes_to[q] : 0.0;
    }
    if (wall_ms) *wall_ms = m->last_wall_ms;
    return MI_OK;
}

// the dealing of columns to shards, without a device (tests compare it with shard.py's ShardPlan: the two forms run ONE plan): out receives
// [rounds, per_rank, rows_per_rank] then, per round k and shard g, (first global column, width); returns the number of words written
extern "C" int64_t mi_multi_plan_debug(uint64_t n, uint64_t n_ext, uint64_t ncols, uint32_t shards, uint64_t *out, uint64_t cap)
{
    if (!out || !shards || !is_pow2(shards)) return -1;
    Plan p;
    p.init(n, n_ext, ncols, shards);
    const uint64_t need = 3 + 2 * p.rounds() * shards;
    if (cap < need) return -(int64_t)need;
    out[0] = p.rounds(); out[1] = p.per_rank; out[2] = p.rows_per_rank;
    uint64_t k2 = 3;
    for (size_t k = 0; k < p.rounds(); k++)
        for (uint32_t g = 0; g < shards; g++) { out[k2++] = p.c0(k, g); out[k2++] = p.width(k, g); }
    return (int64_t)need;
}

extern "C" int mi_multi_tree_info(const mi_multi_tree *t, uint64_t out[6])
{
    MI_REQUIRE(t && out, "null argument");
    out[0] = t->p.G; out[1] = t->p.rows_per_rank; out[2] = t->p.per_rank; out[3] = t->p.rounds(); out[4] = t->p.ncols; out[5] = t->p.n_ext;
    return MI_OK;
}

// level-0 digests of shard g's rows, device pointer on that shard's device (rows_per_rank * 4 words; the subtree's other levels follow)
extern "C" const uint64_t *mi_multi_tree_nodes(const mi_multi_tree *t, int shard) { return (t && shard >= 0 && (uint32_t)shard < t->p.G) ? (const uint64_t *)t->nodes[shard] : nullptr; }

// ---- openings (MerkleTreeGL::getGroupProof, merkleTreeGL.cpp:12-35) over the row-sharded tree
namespace {
struct OpenWin { const u64 *base; uint64_t pitch; uint32_t width, col0; };
#define MM_MAX_WIN 512
__global__ void k_multi_open(u64 *out, uint64_t stride, const uint64_t *rows, const OpenWin *wins, uint32_t nwin, const u64 *nodes, uint64_t height, uint32_t levels,
                             uint32_t ncols, int with_values)
{
    const uint64_t q = blockIdx.x;
    uint64_t idx = rows[q];
    u64 *o = out + q * stride;
    if (with_values)
        for (uint32_t w = 0; w < nwin; w++) {
            const OpenWin W = wins[w];
            for (uint32_t c = threadIdx.x; c < W.width; c += blockDim.x) o[W.col0 + c] = W.base[idx * W.pitch + c];
        }
    // siblings: level l has height >> l nodes, the levels are appended (merkleTreeGL.hpp:61-68)
    uint64_t off = 0, hcur = height;
    for (uint32_t l = 0; l < levels; l++) {
        if (threadIdx.x < 4) o[ncols + 4 * l + threadIdx.x] = nodes[(off + (idx ^ 1)) * 4 + threadIdx.x];
        off += hcur;
        hcur >>= 1;
        idx >>= 1;
    }
}
} // namespace

// proofs (HOST): nq x (ncols + 4 * log2(n_ext)) words: row idx[q]'s ncols values (zeros when with_values == 0 or the rows were released:
// the caller opens them from its own image), then the siblings, leaves upward.
extern "C" int mi_multi_group_proofs(mi_multi_tree *t, uint64_t *proofs, const uint64_t *idx, uint64_t nq, int with_values)
{
    MI_REQUIRE(t && proofs && (idx || !nq), "null argument");
    mi_multi *m = t->m;
    const Plan &p = t->p;
    const uint32_t G = p.G, lv_sub = ilog2_u64(p.rows_per_rank), lv_top = ilog2_u64(G);
    const uint64_t stride = p.ncols + 4ull * (lv_sub + lv_top);
    if (with_values) MI_REQUIRE(t->keep_rows, "the rows of this tree were released (mi_multi_tree_release_rows)");
    memset(proofs, 0, nq * stride * 8);
    for (uint32_t g = 0; g < G; g++) {
        std::vector<uint64_t> local, which;
        for (uint64_t q = 0; q < nq; q++) {
            MI_REQUIRE(idx[q] < p.n_ext, "query index out of range");
            if (idx[q] / p.rows_per_rank == g) { local.push_back(idx[q] % p.rows_per_rank); which.push_back(q); }
        }
        if (local.empty()) continue;
        MM_DEV(m, g);
        std::vector<OpenWin> wins;
        if (with_values)
            for (size_t k = 0; k < p.rounds(); k++)
                for (uint32_t o = 0; o < G; o++) {
                    const uint64_t w = p.width(k, o);
                    if (!w) continue;
                    const u64 *b = o == g ? t->ext[g] + p.ext_base(k) + (uint64_t)g * p.rows_per_rank * w : t->recv[g] + p.recv_off(k, o);
                    wins.push_back({b, w, (uint32_t)w, (uint32_t)p.c0(k, o)});
                }
        MI_REQUIRE(wins.size() <= MM_MAX_WIN, "too many column windows");
        const uint64_t lstride = p.ncols + 4ull * lv_sub;
        void *scr = nullptr;
        const uint64_t need = local.size() * lstride * 8 + local.size() * 8 + wins.size() * sizeof(OpenWin) + 64;
        MI_TRY(mi_scratch(m->ctx[g], need, &scr));
        u64 *d_out = (u64 *)scr;
        uint64_t *d_rows = (uint64_t *)(d_out + local.size() * lstride);
        OpenWin *d_wins = (OpenWin *)(d_rows + local.size());
        MI_HIP_CHECK(hipMemcpyAsync(d_rows, local.data(), local.size() * 8, hipMemcpyHostToDevice, m->cs[g]));
        if (!wins.empty()) MI_HIP_CHECK(hipMemcpyAsync(d_wins, wins.data(), wins.size() * sizeof(OpenWin), hipMemcpyHostToDevice, m->cs[g]));
        MI_HIP_CHECK(hipStreamSynchronize(m->cs[g])); // (the host vectors go out of scope)
        hipLaunchKernelGGL(k_multi_open, dim3((unsigned)local.size()), dim3(64), 0, m->cs[g], d_out, lstride, d_rows, d_wins, (uint32_t)wins.size(), t->nodes[g],
                           p.rows_per_rank, lv_sub, (uint32_t)p.ncols, with_values ? 1 : 0);
        MI_HIP_CHECK(hipGetLastError());
        std::vector<u64> h(local.size() * lstride);
        MI_HIP_CHECK(hipMemcpyAsync(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost, m->cs[g]));
        MI_HIP_CHECK(hipStreamSynchronize(m->cs[g]));
        for (size_t j = 0; j < local.size(); j++) {
            uint64_t *o = proofs + which[j] * stride;
            if (with_values) memcpy(o, &h[j * lstride], p.ncols * 8);
            memcpy(o + p.ncols, &h[j * lstride + p.ncols], 4ull * lv_sub * 8);
            // the top log2 G levels: the roots' tree (on the host since the commit), leaf = shard g
            uint64_t i2 = g, off = 0, hcur = G;
            for (uint32_t l = 0; l < lv_top; l++) {
                memcpy(o + p.ncols + 4ull * (lv_sub + l), &t->roots_host[(off + (i2 ^ 1)) * 4], 32);
                off += hcur; hcur >>= 1; i2 >>= 1;
            }
        }
    }
    return MI_OK;
}

// rows [row0, row0 + nrows) x all columns of the sharded extension, gathered into HOST memory row-major (verification at sizes a host holds)
extern "C" int mi_multi_gather_rows(mi_multi_tree *t, uint64_t *out, uint64_t row0, uint64_t nrows)
{
    MI_REQUIRE(t && out && t->keep_rows, "null argument or released rows");
    mi_multi *m = t->m;
    const Plan &p = t->p;
    MI_REQUIRE(row0 + nrows <= p.n_ext, "rows out of range");
    for (uint64_t r = row0; r < row0 + nrows;) {
        const uint32_t g = (uint32_t)(r / p.rows_per_rank);
        const uint64_t lr = r % p.rows_per_rank, cnt = std::min(row0 + nrows - r, p.rows_per_rank - lr);
        MM_DEV(m, g);
        for (size_t k = 0; k < p.rounds(); k++)
            for (uint32_t o = 0; o < p.G; o++) {
                const uint64_t w = p.width(k, o);
                if (!w) continue;
                const u64 *b = o == g ? t->ext[g] + p.ext_base(k) + (uint64_t)g * p.rows_per_rank * w : t->recv[g] + p.recv_off(k, o);
                MI_HIP_CHECK(hipMemcpy2DAsync(out + (r - row0) * p.ncols + p.c0(k, o), p.ncols * 8, b + lr * w, w * 8, w * 8, cnt, hipMemcpyDeviceToHost, m->cs[g]));
            }
        MI_HIP_CHECK(hipStreamSynchronize(m->cs[g]));
        r += cnt;
    }
    return MI_OK;
}
