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
} // namespace mi
#endif
