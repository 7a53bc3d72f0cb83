#!/usr/bin/env python3
"""Does the LDE hide behind leaf hashing when the two run on different streams?  One 96-column chunk of the headline workload: K extensions
(2^23 -> 2^24 rows) on stream A, K absorbs of an extended chunk (2^24 rows x 96 columns) on stream B; wall time of the two batches one after
the other against both at once.  (The LDE leaves 18-25 % of the VALU issue slots idle while it waits for memory: profiles/r03_pmc_ntt.txt.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import torch  # noqa: E402
import mi_stark  # noqa: E402


def main():
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ctx = mi_stark.Context(0)
    with torch.cuda.stream(sa):        # a context launches on the stream that was current when it was made
        ctx_a = mi_stark.Context(0)
    with torch.cuda.stream(sb):
        ctx_b = mi_stark.Context(0)
    n, ne, w, K = 1 << 23, 1 << 24, 96, 6
    trace = ctx.empty(n * w)
    ctx.fill_synthetic(trace, n * w, 1)
    ext_a, ext_b = ctx.empty(ne * w), ctx.empty(ne * w)
    ctx.fill_synthetic(ext_b, ne * w, 2)
    dig = ctx.empty(ne * 4)
    torch.cuda.synchronize()

    def ldes():
        with torch.cuda.stream(sa):
            for _ in range(K):
                ctx_a.lde(ext_a, trace, ne, n, w)

    def absorbs():
        with torch.cuda.stream(sb):
            for _ in range(K):
                ctx_b.linear_hash_absorb(dig, [(ext_b, 0, w, w)], ne, True, False)

    def wall(fs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fs:
            f()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0)

    for f in (ldes, absorbs):
        f()
    torch.cuda.synchronize()
    t_l, t_a = wall([ldes]), wall([absorbs])
    t_seq = wall([ldes]) + wall([absorbs])
    t_both = min(wall([ldes, absorbs]) for _ in range(3))
    print("per chunk of %d columns: LDE %.2f ms, absorb %.2f ms, one after the other %.2f ms, on two streams at once %.2f ms (%.1f %% of the sum)"
          % (w, t_l / K, t_a / K, t_seq / K, t_both / K, 100 * t_both / t_seq))


if __name__ == "__main__":
    main()
