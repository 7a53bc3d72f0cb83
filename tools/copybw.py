#!/usr/bin/env python3
"""Reference bandwidths on the GPU box: device-to-device copy (what a pure streaming kernel can reach) and
host-to-device / device-to-host copies from pinned and pageable memory (what the host-pointer entry points
mi_lde / mi_merkle_build pay on top of the kernels).  Prints one JSON object."""
import json
import time

import torch


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
x = torch.empty(1 << 31, dtype=torch.int64, device="cuda")  # 16 GiB
y = torch.empty_like(x)
dt = timed(lambda: y.copy_(x), 5)
out["d2d_copy_16GiB_TBps_read_plus_write"] = round(2 * x.numel() * 8 / dt / 1e12, 3)
del x, y
n = 1 << 28  # 2 GiB
d = torch.empty(n, dtype=torch.int64, device="cuda")
hp = torch.empty(n, dtype=torch.int64).pin_memory()
hu = torch.empty(n, dtype=torch.int64)
hu.zero_()
for name, h in (("pinned", hp), ("pageable", hu)):
    out["h2d_%s_2GiB_GBps" % name] = round(n * 8 / timed(lambda: d.copy_(h, non_blocking=True), 3) / 1e9, 2)
    out["d2h_%s_2GiB_GBps" % name] = round(n * 8 / timed(lambda: h.copy_(d, non_blocking=True), 3) / 1e9, 2)
print(json.dumps(out))
