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
// One context per process (one process per GPU); device taken from MI_STARK_DEVICE or the current device.
inline mi_ctx *ctx()
{
    mi_ctx *&c = ctx_slot();
    if (!c) {
        const char *d = std::getenv("MI_STARK_DEVICE");
        if (mi_ctx_create(&c, d ? std::atoi(d) : -1) != MI_OK) fail("mi_ctx_create");
    }
    return c;
}
inline void check(int status, const char *where)
{
    if (status != MI_OK) fail(where);
}
// More than one device for the stage commits (csrc/multi.hip): MI_STARK_DEVICES = "0,1,2,3" -- shard g on that device, a power of two of
// them, the first one the device of ctx() (where the proof's image lives); a device may be named twice (logical shards: how a one-GPU box
// rehearses the path).  Unset or one entry: nullptr, everything runs on ctx()'s device.
inline mi_multi *multi()
{
    static mi_multi *m = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *e = std::getenv("MI_STARK_DEVICES");
        int devs[64], n = 0;
        for (const char *p = e; p && *p && n < 64;) {
            devs[n++] = std::atoi(p);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        if (n > 1 && mi_multi_create(&m, devs, n) != MI_OK) fail("mi_multi_create (MI_STARK_DEVICES)");
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
