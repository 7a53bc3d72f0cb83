#!/usr/bin/env python3
"""Differential fuzz of mi_calculate_h1h2_dev / mi_calculate_z_dev / mi_calculate_z_batch_dev against the oracle (run on a GPU box): random row counts, dimensions,
strides, alphabets (from all-equal to all-distinct), duplicate table rows, heavy hitters, occasional missing values (the failing row
must be the oracle's), zero denominators.  The oracle is the checker here, as in tests/.  Usage: lookup_fuzz.py [count] [first seed]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import numpy as np  # noqa: E402
import mi_stark  # noqa: E402
import glo  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ctx = mi_stark.Context(0)
    bad, t0 = 0, time.time()
    for seed in range(seed0, seed0 + count):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 100, 1000, 4096, 5000, 1 << 14, 100000, 1 << 17]))
        if rng.random() < 0.3:
            n = int(rng.integers(1, 20000))
        dim = int(rng.choice([1, 3]))
        alpha = int(rng.choice([1, 2, 7, 100, 5000, 1 << 30]))
        alphabet = glo.rand_fe(rng, (min(alpha, 4 * n), dim))
        if rng.random() < 0.3:                                 # values that differ only in their last word / in one bit
            alphabet[:, :dim - 1] = alphabet[0, :dim - 1]
            alphabet[:, dim - 1] = np.arange(alphabet.shape[0], dtype=np.uint64)
        t = alphabet[rng.integers(0, alphabet.shape[0], size=n)]
        if rng.random() < 0.3:
            t = t[np.argsort(t[:, dim - 1], kind="stable")]    # a sorted table: long adjacent runs
        f = t[rng.integers(0, n, size=n)]
        if rng.random() < 0.4:
            f[rng.integers(0, n, size=max(1, n // 2))] = t[rng.integers(0, n)]
        missing = rng.random() < 0.15
        if missing:
            for _ in range(int(rng.integers(1, 4))):
                f[rng.integers(0, n), rng.integers(0, dim)] = np.uint64(0xFFFFFFFF00000000 - int(rng.integers(0, 50)))
        cf, ct, c1, c2 = (int(v) for v in rng.permutation(4))
        pad = int(rng.integers(0, 4))
        cols = 4 * dim + pad
        area = glo.rand_fe(rng, (n, cols))
        area[:, cf * dim:(cf + 1) * dim], area[:, ct * dim:(ct + 1) * dim] = f, t
        area = np.ascontiguousarray(area.reshape(-1))
        d = ctx.to_device(area)
        err = None
        try:
            ctx.calculate_h1h2(d[c1 * dim:], cols, d[c2 * dim:], cols, d[cf * dim:], cols, d[ct * dim:], cols, dim, n)
        except mi_stark.MiStarkError as e:
            err = str(e)
        wbad = glo.calculate_h1h2(area, c1 * dim, cols, c2 * dim, cols, cf * dim, cols, ct * dim, cols, dim, n)
        ok = np.array_equal(ctx.to_host(d), area) and ((err is None) if wbad == 0 else (err is not None and ("w=%d" % (wbad - 1)) in err))
        # grand product on the same box: random numerators / denominators, sometimes a permutation (closes), sometimes a zero denominator
        zc = 9 + pad
        za = glo.rand_fe(rng, (n, zc))
        mode = int(rng.integers(0, 3))
        if mode == 1:
            za[:, 3:6] = za[rng.permutation(n), 0:3]
        if mode == 2:
            za[rng.integers(0, n), 3:6] = 0
        za = np.ascontiguousarray(za.reshape(-1))
        dz = ctx.to_device(za)
        closes = ctx.calculate_z(dz[6:], zc, dz, zc, dz[3:], zc, n)
        wcloses = glo.calculate_z(za, 6, zc, 0, zc, 3, zc, n)
        okz = np.array_equal(ctx.to_host(dz), za) and closes == bool(wcloses) and (mode != 1 or closes)
        # the batch entry point: k products over one wide area (numerators / denominators side by side, z columns behind them), several zero
        # denominators, one product that closes; more products than one launch takes now and then
        k = int(rng.choice([1, 2, 5, 13, 33, 40]))
        nb = min(n, 20000)
        wt, wz = 6 * k + int(rng.integers(0, 3)), 3 * k + int(rng.integers(0, 3))
        ba = glo.rand_fe(rng, (nb * (wt + wz),))
        src = ba[:nb * wt].reshape(nb, wt)
        for _ in range(int(rng.integers(0, 4))):
            src[rng.integers(0, nb), 6 * int(rng.integers(0, k)) + 3:][:3] = 0
        cl = int(rng.integers(0, k))
        src[:, 6 * cl:6 * cl + 3] = src[rng.permutation(nb), 6 * cl + 3:6 * cl + 6]
        db = ctx.to_device(ba)
        got = ctx.calculate_z_batch([(db[nb * wt + 3 * i:], wz, db[6 * i:], wt, db[6 * i + 3:], wt) for i in range(k)], nb)
        want = [bool(glo.calculate_z(ba, nb * wt + 3 * i, wz, 6 * i, wt, 6 * i + 3, wt, nb)) for i in range(k)]
        okz = okz and np.array_equal(ctx.to_host(db), ba) and got == want
        if not (ok and okz):
            bad += 1
        if not (ok and okz) or seed % 25 == 0:
            print("seed %d n %d dim %d alphabet %d missing %s: h1h2 %s z %s (mode %d)  (%d s)" % (seed, n, dim, alpha, wbad != 0, ok, okz, mode, time.time() - t0), flush=True)
    print("lookup fuzz: %d cases, %d mismatches" % (count, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
