#!/usr/bin/env python3
"""What does the leaf kernel's EMIT form cost?  The same 2^24 x W compact matrix absorbed by the plain form (mi_merkle_build_dev: a lane's
rows 16 apart) and by the emitting form (inside mi_lde_merkle_dev_tiled: 64 consecutive rows per wave, every word also stored tile-major).
Run under `rocprofv3 --kernel-trace --stats`: the two k_linear_hash_rows_lines instantiations' average times are the answer.
    gpurun -- 'cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/leaf_emit -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/leaf_emit_probe.py'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import torch
import mi_stark

ctx = mi_stark.Context(0)
log_n = int(os.environ.get("PROBE_LOG_N", "23"))
n, n_ext = 1 << log_n, 2 << log_n
loan = torch.empty((60 << 30) // 8, dtype=torch.int64, device="cuda")
for w in [int(x) for x in os.environ.get("PROBE_WIDTHS", "128,96,32").split(",")]:
    src = ctx.empty(n * w)
    ctx.fill_synthetic(src, n * w, 0x5EED)
    nodes, ext_t, ext_r = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * w), ctx.empty(n_ext * w)
    ctx.lend_workspace(loan)
    for rep in range(3):
        ctx.lde_merkle_dev_tiled(nodes, ext_t, src, n, n_ext, w)      # emitting form over the compact chunk
    ctx.lde(ext_r, src, n_ext, n, w)
    for rep in range(3):
        ctx.merkle_build(nodes, ext_r, w, n_ext)                       # plain form over the same words, compact row-major
    ctx.lend_workspace(None)
    torch.cuda.synchronize()
    print("width", w, "done", flush=True)
    del src, nodes, ext_t, ext_r
