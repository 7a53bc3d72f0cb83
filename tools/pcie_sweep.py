#!/usr/bin/env python3
"""PCIe-inclusive step (mi_lde_merkle_host) against the upload chunk width, plus the bare strided-upload rate: the host
trace is row-major, a column chunk is a 2-D copy whose rows are chunk * 8 bytes long at a pitch of 5 320 bytes, and the DMA
engines move short rows slower than long ones.  Prints one JSON object (profiles/r02_pcie_chunk_sweep.json)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import torch
import mi_stark

log_n, ncols = int(os.environ.get("LOG_N", 23)), 665
n, n_ext = 1 << log_n, 2 << log_n
ctx = mi_stark.Context(0)
trace = ctx.empty(n * ncols)
ctx.fill_synthetic_2d(trace, n, ncols, ncols, 0, 0x5EED0003)
host = torch.empty(n * ncols, dtype=torch.int64, pin_memory=True)
host.copy_(trace)
torch.cuda.synchronize()
del trace
ext, nodes = ctx.empty(n_ext * ncols), ctx.empty((2 * n_ext - 1) * 4)
out = {"rows": n, "cols": ncols, "steps": []}
hip = ctypes.CDLL("libamdhip64.so")
for chunk in (64, 128, 256, 320):
    # bare upload of every chunk, same 2-D copies, no kernels
    stage = ctx.empty(n * chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c0 in range(0, ncols, chunk):
        cw = min(chunk, ncols - c0)
        # hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind=1 H2D, stream 0)
        rc = hip.hipMemcpy2DAsync(ctypes.c_void_p(stage.data_ptr()), ctypes.c_size_t(cw * 8), ctypes.c_void_p(host.data_ptr() + 8 * c0),
                                  ctypes.c_size_t(ncols * 8), ctypes.c_size_t(cw * 8), ctypes.c_size_t(n), ctypes.c_int(1), ctypes.c_void_p(0))
        assert rc == 0, rc
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    del stage
    ctx.lde_merkle_host(nodes, ext, host.data_ptr(), n, n_ext, ncols, chunk_cols=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ctx.lde_merkle_host(nodes, ext, host.data_ptr(), n, n_ext, ncols, chunk_cols=chunk)
        root = ctx.to_host(nodes[(2 * n_ext - 2) * 4:(2 * n_ext - 1) * 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    out["steps"].append({"chunk_cols": chunk, "row_bytes": chunk * 8, "upload_only_ms": 1e3 * t_up, "upload_GBps": n * ncols * 8 / t_up / 1e9,
                         "step_ms": 1e3 * dt, "root": [int(v) for v in root]})
    print(out["steps"][-1], file=sys.stderr, flush=True)
print(json.dumps(out))
