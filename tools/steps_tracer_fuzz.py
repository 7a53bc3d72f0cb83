#!/usr/bin/env python3
"""Differential fuzz of host/steps_tracer.hpp on the CPU: random Steps classes written out as generated per-row C++ (tests/gen_steps_cpp.py:
every opcode of the reference's three table formats, random polynomial maps, blow-up 2 / 4 / 8), compiled, RECORDED, the recordings
translated and run through the library's host executors beside the compiled functions themselves (tests/cpp/test_steps_tracer.cpp); any
stored word that differs is a mismatch.  No GPU.  Usage: steps_tracer_fuzz.py [count] [first seed]."""
import os
import pathlib
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import numpy as np  # noqa: E402
import test_steps_tracer as tst  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    bad, t0 = 0, time.time()
    for seed in range(seed0, seed0 + count):
        rng = np.random.default_rng(seed)
        kw = dict(nbits=int(rng.integers(4, 9)), ext=int(rng.integers(1, 4)), widths=(int(rng.integers(6, 30)), int(rng.integers(6, 20)), int(rng.integers(8, 40))),
                  tmpexp=int(rng.integers(8, 30)), n_const=int(rng.integers(6, 20)), n_pub=int(rng.integers(1, 9)), n_evals=int(rng.integers(4, 40)),
                  sizes=(int(rng.integers(115, 300)), int(rng.integers(115, 400)), int(rng.integers(20, 120))), with3prev=bool(rng.integers(0, 2)))
        with tempfile.TemporaryDirectory() as d:
            try:
                r, steps = tst.synthetic_case(pathlib.Path(d), seed, **kw)
                ok = r.returncode == 0 and r.stdout.strip().endswith("OK") and all(
                    re.search(r"%s: \d+ recorded operations .* 0 differ \(translated\) 0 differ \(lowered\)" % s, r.stdout) for s in steps)
                out = r.stdout
            except AssertionError as e:       # a generator case the table formats cannot express (e.g. too few output columns)
                ok, out = None, str(e)[-300:]
        if ok is False:
            bad += 1
        print("seed %d %s: %s  (%d s)" % (seed, kw, "skipped (generator)" if ok is None else "ok" if ok else "MISMATCH\n" + out, time.time() - t0), flush=True)
    print("steps tracer fuzz: %d cases, %d mismatches" % (count, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
