#!/usr/bin/env python3
"""MI_NTT_PERSISTENT=1 (the persistent double-buffered radix-256 pass, csrc/ntt.hip k_ntt_pass_pers) against the oracle: LDE and NTT / INTT
of widths that are multiples of 32 at sizes where the persistent form is taken (at least 1024 tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import glo, mi_stark
ctx = mi_stark.Context(0)
ok = True
for log_n, ncols in ((18, 96), (17, 64), (19, 32)):
    n, ne = 1 << log_n, 2 << log_n
    tr = glo.splitmix64(7 + log_n, n * ncols).reshape(n, ncols)
    d = ctx.to_device(tr)
    out = ctx.empty(ne * ncols)
    ctx.lde(out, d, ne, n, ncols)
    ctx.sync()
    got = ctx.to_host(out).reshape(ne, ncols)
    want = glo.extend_pol(tr, ne, n, ncols)
    e = np.array_equal(got, want)
    back = ctx.empty(ne * ncols)
    ctx.ntt(back, out, ne, ncols, inverse=True)
    ctx.ntt(out, back, ne, ncols)
    ctx.sync()
    e2 = np.array_equal(ctx.to_host(out).reshape(ne, ncols), want) and np.array_equal(ctx.to_host(back).reshape(ne, ncols), glo.ntt(want, ne, ncols, inverse=True))
    print("lde 2^%d x %d: %s; intt/ntt round trip: %s" % (log_n, ncols, "OK" if e else "MISMATCH", "OK" if e2 else "MISMATCH"), flush=True)
    ok = ok and e and e2
sys.exit(0 if ok else 1)
