// mi_runtime.hpp -- process-wide libmi_stark context used by the header shims, and the reference's error
// convention: log + exitProcess() (src/utils/exit_process.cpp:7-22; e.g. starks.hpp:99-100).
#ifndef MI_RUNTIME_HPP
#define MI_RUNTIME_HPP
#include <cstdio>
#include <cstdlib>
#include "../../include/mi_stark.h"

namespace mi {
inline mi_ctx *&ctx_slot()
{
    static mi_ctx *c = nullptr;
    return c;
}
[[noreturn]] inline void fail(const char *where)
{
    std::fprintf(stderr, "mi_stark: %s failed: %s\n", where, mi_last_error());
    std::exit(-1); // exitProcess() without the 5 s grace sleep
}
// MI_STARK_DEVICES ("0,1,2,3": shard g on that device) parsed strictly: digits and commas only -- atoi would read "a,b" as devices 0,0.
inline int parseDevices(int *devs, int cap)
{
    const char *e = std::getenv("MI_STARK_DEVICES");
    if (!e || !*e) return 0;
    int n = 0;
    for (const char *p = e; *p;) {
        if (*p < '0' || *p > '9' || n >= cap) {
            std::fprintf(stderr, "mi_stark: MI_STARK_DEVICES=\"%s\" is not a comma-separated list of at most %d device numbers\n", e, cap);
            std::exit(-1);
        }
        long v = 0;
        while (*p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > 1023) v = 1023; p++; }
        devs[n++] = (int)v;
        if (*p == ',') { p++; if (!*p) { std::fprintf(stderr, "mi_stark: MI_STARK_DEVICES=\"%s\" ends in a comma\n", e); std::exit(-1); } }
        else if (*p) { std::fprintf(stderr, "mi_stark: MI_STARK_DEVICES=\"%s\" is not a comma-separated list of device numbers\n", e); std::exit(-1); }
    }
    return n;
}
// One context per process (one process per GPU); device taken from MI_STARK_DEVICE, else the first entry of MI_STARK_DEVICES (shard 0
// shares the device of the proof's image: host/starks.hpp lends it regions of that image), else the current device.
inline mi_ctx *ctx()
{
    mi_ctx *&c = ctx_slot();
    if (!c) {
        const char *d = std::getenv("MI_STARK_DEVICE");
        int devs[64];
        const int n = parseDevices(devs, 64);
        if (mi_ctx_create(&c, d ? std::atoi(d) : n > 1 ? devs[0] : -1) != MI_OK) fail("mi_ctx_create");
    }
    return c;
}
inline void check(int status, const char *where)
{
    if (status != MI_OK) fail(where);
}
// More than one device for the stage commits (csrc/multi.hip): MI_STARK_DEVICES = "0,1,2,3" -- shard g on that device, a power of two of
// them (at most 16), the first one the device of ctx() (where the proof's image lives: checked); a device may be named twice (logical
// shards: how a one-GPU box rehearses the path).  Unset or one entry: nullptr, everything runs on ctx()'s device.
inline mi_multi *multi()
{
    static mi_multi *m = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        int devs[64];
        const int n = parseDevices(devs, 64);
        if (n > 1) {
            // shard 0 works in regions of the image (mi_multi_lend) and is "home" for every device switch: it must be the image's device
            if (devs[0] != mi_ctx_device(ctx())) {
                std::fprintf(stderr, "mi_stark: MI_STARK_DEVICES starts with device %d but the proof's image lives on device %d (MI_STARK_DEVICE): "
                                     "the first shard must share the image's device\n", devs[0], mi_ctx_device(ctx()));
                std::exit(-1);
            }
            if (mi_multi_create(&m, devs, n) != MI_OK) fail("mi_multi_create (MI_STARK_DEVICES)");
        }
    }
    return m;
}
// Device scratch of the host classes (FRI polynomials, step-tree leaves and nodes, opening buffers).  A caller that owns a plan of
// the HBM (host/starks.hpp) lends a region and the classes carve it in order; without one every request is a device allocation.
// Memory handed back to the driver is wiped in the background on the GPU's own bandwidth, so a proof that
// allocates nothing while it runs is also a faster proof.
struct Bump { uint64_t *base = nullptr; uint64_t cap = 0, used = 0; };
inline Bump &bump()
{
    static Bump b;
    return b;
}
inline void lendScratch(uint64_t *base, uint64_t elems) { bump() = Bump{base, elems, 0}; }
inline uint64_t *devAlloc(uint64_t elems, const char *what)
{
    Bump &b = bump();
    const uint64_t e = (elems + 15) & ~15ULL; // 128-byte granules
    if (b.base && b.used + e <= b.cap) {
        uint64_t *p = b.base + b.used;
        b.used += e;
        return p;
    }
    uint64_t *p = (uint64_t *)mi_dev_alloc(ctx(), (elems ? elems : 1) * 8);
    if (!p) fail(what);
    return p;
}
inline void devFree(uint64_t *p)
{
    Bump &b = bump();
    if (!p || (b.base && p >= b.base && p < b.base + b.cap)) return; // part of a lent region: its owner reuses it
    mi_dev_free(ctx(), p);
}
} // namespace mi
#endif
