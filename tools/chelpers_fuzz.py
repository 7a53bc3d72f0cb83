#!/usr/bin/env python3
"""Differential fuzz of the constraint-evaluator backends (run on a GPU box): random step42ns / step52ns / base-domain programs (every opcode,
zkEVM-shaped, Horner chains), random chunk sizes / batch sizes / row ranges, linear kernel forced on or off; the native backend
(generated kernels + linear kernel) and the interpreter against the oracle's opcode-by-opcode restatement.  The oracle is the
checker here, as in tests/.  Usage: chelpers_fuzz.py [count] [first seed]."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import numpy as np  # noqa: E402
import mi_stark  # noqa: E402
import glo  # noqa: E402
import chelpers_programs as cp  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    ctx = mi_stark.Context(0)
    bad = 0
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        for seed in range(seed0, seed0 + count):
            rng = np.random.default_rng(seed)
            nrows = int(rng.choice([128, 1024, 4096]))
            secs = [(0, 40), (nrows * 40, 9), (nrows * 49, 3)]
            sections = [(o, w, nrows) for o, w in secs]
            cc = int(rng.choice([0, 1500, 4000, 9000]))
            os.environ["MI_CHELPERS_LIN_MIN"] = str(int(rng.choice([1, 256])))
            os.environ["MI_CHELPERS_INPLACE_MAX"] = str(int(rng.choice([0, 3, 8, 64])))   # slabs read in place / through the tile-major copy
            ctx.set_chelpers_batch_rows(int(rng.choice([0, 64, 512])))
            r0 = int(rng.integers(0, nrows // 2))
            nr = int(rng.integers(1, nrows - r0 + 1))
            pols = glo.rand_fe(rng, nrows * 52, canonical=False)
            cpols = glo.rand_fe(rng, nrows * 7)
            want = np.zeros(nrows * 3, dtype=np.uint64)
            kind = seed % 4
            if kind == 3:       # a base-domain step (step2prev / step3prev / step3): results stored into polynomials, compiled kernels only
                out = (nrows * 52, 120)
                ops, args = cp.synthetic_program_base(rng, nrows, secs, out, 7, 5, 4, n_ops=int(rng.integers(120, 400)))
                full = np.zeros(nrows * (52 + 120), dtype=np.uint64)
                full[:nrows * 52] = pols
                chal, pub, x = glo.rand_fe(rng, 15), glo.rand_fe(rng, 4), glo.rand_fe(rng, nrows * 2)
                want = full.copy()
                glo.chelpers_stepbase(ops, args, want, cpols, 7, chal, pub, x, 2, np.arange(r0, r0 + nr))
                prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3)
                prog.build_native(cache_dir=td, chunk_cost=cc)
                d_pols = ctx.to_device(full)
                prog.run_base(d_pols, ctx.to_device(cpols), 7, chal, pub, ctx.to_device(x), 2, r0, nr)
                outs = [want, ctx.to_host(d_pols)]       # no interpreter form for these
                prog.close()
            elif kind == 2:
                ops, args = cp.synthetic_program52(rng, secs, 7, 6, length=int(rng.integers(20, 400)))
                chal, evals = glo.rand_fe(rng, 21), glo.rand_fe(rng, 18)
                xd, xdw = glo.rand_fe(rng, nrows * 3), glo.rand_fe(rng, nrows * 3)
                glo.chelpers_step52ns(ops, args, pols, cpols, 7, chal, evals, xd, xdw, want, r0, nr)
                dev = [ctx.to_device(a) for a in (pols, cpols, xd, xdw)]
                outs = []
                for native in (False, True):
                    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=52)
                    if native:
                        prog.build_native(cache_dir=td, chunk_cost=cc)
                    f = ctx.zeros(nrows * 3)
                    prog.run52(dev[0], dev[1], 7, chal, evals, dev[2], dev[3], f, r0, nr)
                    outs.append(ctx.to_host(f))
                    prog.close()
            else:
                if kind == 0:
                    ops, args = cp.synthetic_program(rng, nrows, secs, 7, 5, 4, passes=int(rng.integers(1, 4)))
                else:
                    # (r04: also constraints over extension-valued polynomials -- opcodes 74 / 75 / 44 / 41 / 72 -- and uniformly picked shifted reads)
                    p3 = float(rng.choice([0.0, 0.5, 1.0]))
                    ops, args = cp.synthetic_program_zkevm_shape(rng, nrows, secs, 7, 4, field_ops=int(rng.integers(200, 2500)),
                                                                 long_lived=int(rng.integers(0, 40)), pol3_frac=p3, ext_frac=0.3 if p3 else 0.09,
                                                                 partition=bool(p3 and rng.random() < 0.5), sec_weights=[0.7, 0.2, 0.1] if p3 else None)
                chal, pub, x, zh = glo.rand_fe(rng, 15), glo.rand_fe(rng, 4), glo.rand_fe(rng, nrows * 2), glo.rand_fe(rng, 4)
                glo.chelpers_step42ns(ops, args, pols, cpols, 7, chal, pub, x, 2, zh, want, r0, nr)
                dev = [ctx.to_device(a) for a in (pols, cpols, x)]
                outs = []
                for native in (False, True):
                    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=sections, n_const=7, nrows_ext=nrows)
                    if native:
                        prog.build_native(cache_dir=td, chunk_cost=cc)
                    q = ctx.zeros(nrows * 3)
                    prog.run(dev[0], dev[1], 7, chal, pub, dev[2], 2, zh, q, r0, nr)
                    outs.append(ctx.to_host(q))
                    prog.close()
            ok = [bool(np.array_equal(o, want)) for o in outs]
            if not all(ok):
                bad += 1
            print("seed %d kind %d rows %d [%d,+%d) chunk_cost %d lin_min %s inplace_max %s: interpreter %s native %s  (%.0f s)" %
                  (seed, kind, nrows, r0, nr, cc, os.environ["MI_CHELPERS_LIN_MIN"], os.environ["MI_CHELPERS_INPLACE_MAX"], ok[0], ok[1], time.time() - t0), flush=True)
    print("fuzz: %d programs, %d mismatches" % (count, bad))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
