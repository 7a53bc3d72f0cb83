#!/usr/bin/env python3
"""Time a base-domain constraint step (step2prev / step3prev / step3 numbering) of zkEVM size on the GPU: a synthetic program in the
shape of the zkEVM step3 program (the real tables cannot travel: as many field operations per row -- 12 729 --, its operation mix and
challenge-weighted accumulation, ~430 stored elements, ~40 live words), over 2^23 rows of sections as wide as cm1_n, cm3_n and
tmpExp_n (665 / 371 / 265 columns), its results stored into a fourth section; sampled rows against the oracle.  Prints one JSON line.
--precompile J: fill the in-tree code-object cache with J parallel processes (no GPU needed) and exit."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=23)
    ap.add_argument("--field-ops", type=int, default=12729)
    ap.add_argument("--widths", type=int, nargs=3, default=[665, 371, 265])
    ap.add_argument("--out-cols", type=int, default=430, help="columns of the section the results go to (the zkEVM step3 stores 430)")
    ap.add_argument("--n-const", type=int, default=218)
    ap.add_argument("--precompile", type=int, default=0)
    ap.add_argument("--shard", type=int, default=-1, help="internal")
    a = ap.parse_args()
    import mi_stark
    import glo
    import chelpers_programs as cp
    N = 1 << a.log_n
    w1, w2, w3 = a.widths
    offs = [0, N * w1, N * (w1 + w2), N * (w1 + w2 + w3)]
    secs = [(offs[0], w1), (offs[1], w2), (offs[2], w3)]
    ops, args = cp.synthetic_program_zkevm_shape(np.random.default_rng(3), N, secs, a.n_const, 8, field_ops=a.field_ops, next_shift=1, long_lived=30,
                                                 base_out=(offs[3], a.out_cols))
    if a.precompile or a.shard >= 0:
        import subprocess
        if a.shard >= 0:
            prog = mi_stark.ChelpersProgram(None, ops, args, sections=[(o, w, N) for o, w in secs], n_const=a.n_const, nrows_ext=N, step=mi_stark.MI_CHELPERS_STEP3)
            prog.precompile_shard(a.shard, a.precompile)
            return
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--precompile", str(a.precompile), "--shard", str(k), "--log-n", str(a.log_n),
                                   "--field-ops", str(a.field_ops), "--out-cols", str(a.out_cols), "--n-const", str(a.n_const), "--widths"] + [str(w) for w in a.widths])
                 for k in range(a.precompile)]
        if any(p.wait() for p in procs):
            raise SystemExit("a precompile shard failed")
        print("precompiled")
        return
    import torch
    ctx = mi_stark.Context(0)
    t0 = time.perf_counter()
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=[(o, w, N) for o, w in secs], n_const=a.n_const, nrows_ext=N, step=mi_stark.MI_CHELPERS_STEP3)
    native = prog.build_native()
    t_build = time.perf_counter() - t0
    pols = ctx.empty(N * (w1 + w2 + w3 + a.out_cols))
    ctx.fill_synthetic(pols, N * (w1 + w2 + w3), 0x5EED0200)
    pols[offs[3]:].zero_()
    cpols, x = ctx.empty(N * a.n_const), ctx.empty(N)
    ctx.fill_synthetic(cpols, N * a.n_const, 0x5EED0201)
    ctx.geom_seq(x, N, 1, glo.lib().glo_w(a.log_n))
    rng = np.random.default_rng(4)
    chal, pub = glo.rand_fe(rng, 15), glo.rand_fe(rng, 8)
    prog.reserve(N)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        prog.run_base(pols, cpols, a.n_const, chal, pub, x, 1, 0, N)
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
    # sampled rows against the oracle over a sparse host copy of what they read
    import mmap
    dec, _ = cp.decode_base(ops, args)
    rows = [0, 1, 63, 64, N // 2, N - 2, N - 1] + [int(v) for v in rng.integers(0, N, 5)]
    rd, wr, ca = cp.touched_addresses_base(dec, rows, a.n_const)

    def sparse(n):
        return np.frombuffer(mmap.mmap(-1, n * 8, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000)), dtype=np.uint64)
    h_p, h_c, h_x = sparse(pols.numel()), sparse(N * a.n_const), sparse(N)
    inputs = sorted(i for i in rd if i < offs[3])
    h_p[np.array(inputs)] = ctx.to_host(pols[torch.tensor(inputs, dtype=torch.int64, device=pols.device)])
    h_c[np.array(sorted(ca))] = ctx.to_host(cpols[torch.tensor(sorted(ca), dtype=torch.int64, device=pols.device)])
    h_x[rows] = ctx.to_host(x[torch.tensor(rows, device=pols.device)])
    glo.chelpers_stepbase(ops, args, h_p, h_c, a.n_const, chal, pub, h_x, 1, rows)
    widx = sorted(wr)
    got = ctx.to_host(pols[torch.tensor(widx, dtype=torch.int64, device=pols.device)])
    ok = bool(np.array_equal(got, h_p[np.array(widx)]))
    print(json.dumps({"metric": "base_domain_constraint_step_ms", "value": min(times), "unit": "ms", "runs_ms": times, "rows": N,
                      "program": "synthetic, base-domain numbering, shaped and sized like the zkEVM step3 program", "translator_stats": prog.stats,
                      "lowering": prog.lower_stats(), "native": native, "translate_and_build_s": t_build,
                      "written_cells_of_sampled_rows_match_oracle": ok, "cells_checked": len(widx)}))
    prog.close()
    ctx.close()
    if not ok:
        raise SystemExit("mismatch against the oracle")


if __name__ == "__main__":
    main()
