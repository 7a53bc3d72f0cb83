#!/usr/bin/env python3
"""Times the constraint-evaluator interpreter on a synthetic step42ns-sized program (tests/chelpers_programs.py) for a few
LDS footprints (mi_set_chelpers_min_words = occupancy of larger programs).  Usage: chelpers_probe.py [log_rows] [words ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mi_stark
import chelpers_programs as cpg
import glo

log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20
words = [int(w) for w in sys.argv[2:]] or [96]
NE = 1 << log_rows
w1, w2, w3, n_const = 665, 128, 371, 218
ctx = mi_stark.Context(0)
off = [0, NE * w1, NE * (w1 + w2)]
pols = ctx.empty(NE * (w1 + w2 + w3))
ctx.fill_synthetic(pols, pols.numel(), 1)
cpols = ctx.empty(NE * n_const)
ctx.fill_synthetic(cpols, cpols.numel(), 2)
x = ctx.empty(NE)
ctx.fill_synthetic(x, NE, 3)
secs = [(off[0], w1), (off[1], w2), (off[2], w3)]
per_pass = len(cpg.decode(*cpg.synthetic_program(np.random.default_rng(42), NE, secs, n_const, 5, 8, passes=4))[0]) / 4.0
ops, args = cpg.synthetic_program(np.random.default_rng(42), NE, secs, n_const, 5, 8, passes=max(1, int(round(17986 / per_pass))))
prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=[(o, w, NE) for (o, w) in secs], n_const=n_const, nrows_ext=NE)
rng = np.random.default_rng(1)
chal, pub, zh = glo.rand_fe(rng, 15), glo.rand_fe(rng, 8), glo.rand_fe(rng, 2)
q = ctx.empty(NE * 3)
out = {"rows": NE, "instructions_per_row": prog.stats["instructions_per_row"], "runs": []}
for w in words:
    ctx.set_chelpers_min_words(w)
    prog.run(pols, cpols, n_const, chal, pub, x, 1, zh, q, 0, NE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prog.run(pols, cpols, n_const, chal, pub, x, 1, zh, q, 0, NE)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["runs"].append({"lds_words_per_row": w, "ms": 1e3 * dt, "ns_per_row_instruction_per_wave": dt / (NE / 64) / prog.stats["instructions_per_row"] * 1e9,
                        "extrapolated_s_at_2^24_rows": dt * (1 << 24) / NE})
print(json.dumps(out))
