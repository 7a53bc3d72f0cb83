#!/usr/bin/env python3
"""Does the width of a tile-major section matter to the kernels that write and read it?  A tile is width x 512 bytes: with a power-of-two
width every wave's column c lies at the same address modulo the tile size, and waves that run in step (the leaf kernel's do) send their
512-byte stores of a moment to the same few memory channels.  Times mi_lde_merkle_dev_tiled (one 2^23-row extension per width, torch
events around the call) and, under `rocprofv3 --kernel-trace`, gives the leaf kernel's launches in order.
    gpurun -- 'python tools/tiled_width_probe.py > gpurun_out/tiled_width_probe.txt'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import torch
import mi_stark

ctx = mi_stark.Context(0)
log_n = int(os.environ.get("PROBE_LOG_N", "22"))
n, n_ext = 1 << log_n, 2 << log_n
widths = [int(w) for w in os.environ.get("PROBE_WIDTHS", "96,120,128,136,192,256,264").split(",")]
wmax = max(widths)
src = ctx.empty(n * wmax)
ctx.fill_synthetic(src, n * wmax, 0x5EED)
nodes, ext = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * wmax)
loan = torch.empty((40 << 30) // 8, dtype=torch.int64, device="cuda")
ctx.lend_workspace(loan)
for w in widths:
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.lde_merkle_dev_tiled(nodes, ext, src, n, n_ext, w, src_pitch=w)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("width %4d: %.2f ms (%.4f ms per column) %s" % (w, min(ts), min(ts) / w, ["%.2f" % t for t in ts]), flush=True)
ctx.lend_workspace(None)
