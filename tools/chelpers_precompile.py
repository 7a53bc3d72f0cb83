#!/usr/bin/env python3
"""Fill the in-tree code-object cache (merlin-zkevm-prover_amd/_chelpers_cache/) with the compiled constraint programs of
bench_starks.py's default (zkEVM-shaped) configuration and of the configurations the GPU tests prove.  Needs no GPU: hiprtc cross-compiles
gfx950.  A proving key's programs are compiled once; the GPU box then loads the code objects instead of spending its minutes in the
compiler (about 10 s per 25 000-instruction kernel).  The kernels of a program are compiled by --jobs processes in parallel (each
takes every jobs-th kernel: mi_chelpers_precompile_shard)."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))

# bench_starks.py (class Starks end to end): the default zkEVM-shaped STARK and the small one the GPU test runs
STARKS_CONFIGS = [
    [],
    ["--log-n", "12", "--widths", "37", "20", "40", "--tmpexp", "60", "--n-const", "11", "--n-evals", "24", "--n-queries", "16", "--n-lookups", "2", "2",
     "--n-products", "6", "--field-ops", "200", "300", "400", "1500", "700"],
    ["--shape", "recursive1"],
    ["--shape", "c12a"],
    # tests/test_genproof_parity.py: the shapes proved by the device AND by the oracle prover (byte-for-byte comparison)
    ["--log-n", "14", "--widths", "96", "40", "71", "--tmpexp", "110", "--n-const", "30", "--n-evals", "200", "--n-queries", "32", "--n-lookups", "3", "3",
     "--n-products", "12", "--field-ops", "300", "700", "900", "3000", "1200"],
    ["--log-n", "10", "--ext-bits", "3", "--qdeg", "7", "--widths", "18", "0", "39", "--tmpexp", "90", "--n-const", "9", "--n-evals", "30",
     "--n-queries", "8", "--n-lookups", "0", "0", "--n-products", "13", "--fri-steps", "13", "9", "5", "--field-ops", "0", "300", "0", "900", "400"],
    ["--log-n", "12", "--ext-bits", "3", "--qdeg", "7", "--widths", "18", "0", "39", "--tmpexp", "14", "--n-const", "52", "--n-evals", "118",
     "--n-queries", "43", "--n-lookups", "0", "0", "--n-products", "1", "--fri-steps", "15", "11", "7", "4", "--field-ops", "0", "201", "1761", "3483", "463"],
    ["--log-n", "12"],                                      # the zkEVM's full widths, counts and program sizes at 2^12 rows
    ["--log-n", "16"],                                      # ... and at 2^16, 2^18 (tests/test_genproof_parity.py's largest in the suite)
    ["--log-n", "18"],
    ["--log-n", "21"],                                      # ... and the once-per-round runs at 2^21 / 2^22 (MI_PARITY_LOG_N)
    ["--log-n", "22"],
]


def precompile_starks(jobs, only):
    import bench_starks
    for ci, argv in enumerate(STARKS_CONFIGS):
        if only >= 0 and ci != only:
            continue
        t0 = time.time()
        # both layouts of the image: the witness, the wide extended sections and the resident constants tile-major (one device), and all
        # row-major (MI_STARK_DEVICES, MI_STARK_TILED_WITNESS / _EXT / _CONSTS = 0, a STARK whose lookups read witness columns); for the
        # default configuration the A/B forms too (extension row-major; constants row-major).  Kernels that do not differ are found in the cache the second time.
        for tiled, tiled_ext, tiled_c in (("1", "1", "1"), ("0", "0", "0")) + ((("1", "0", "1"), ("1", "1", "0")) if ci == 0 else ()):
            env = dict(os.environ, MI_STARK_TILED_WITNESS=tiled, MI_STARK_TILED_EXT=tiled_ext, MI_STARK_TILED_CONSTS=tiled_c)
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench_starks.py")] + argv + ["--precompile", str(s), str(jobs)], env=env) for s in range(jobs)]
            if any(p.wait() for p in procs):
                raise SystemExit("a precompile shard of bench_starks.py failed")
        st = bench_starks.compiled_programs(bench_starks.parse(argv))
        print("precompiled bench_starks", argv or "(default)", {k: (v["kernels"], v["cache_hits"], v["code_bytes"]) for k, v in st.items()},
              "%.1f s" % (time.time() - t0), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--starks-only", type=int, default=-1, help="only the bench_starks.py configuration with this index (-1: all of them)")
    a = ap.parse_args()
    precompile_starks(a.jobs, a.starks_only)


if __name__ == "__main__":
    main()
