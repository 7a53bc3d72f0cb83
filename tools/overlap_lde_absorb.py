#!/usr/bin/env python3
"""Experiment: the headline step (2^23 x 665 -> LDE 2^24 -> leaf sponge -> tree) with the LDE of column chunk k + 1 on ONE stream beside the
leaf absorption of chunk k on ANOTHER (two contexts on the device = two streams), against the same chunks back to back on one stream and
against the product's order (whole LDE, then one leaf launch).  Both kernels want VALU issue slots; the question is whether the leaf
sponge's waves fill what the transform's barriers and LDS round trips leave idle.

    python tools/overlap_lde_absorb.py [--chunk 96] [--steps 3] [--log-n 23] [--ncols 665]

Prints one JSON line; the roots of the three forms must agree (and, at the default size, equal bench.ROOT_2P23_X665)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=96)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=23)
    ap.add_argument("--ncols", type=int, default=665)
    a = ap.parse_args()
    import torch
    import mi_stark
    import bench
    n, ne, w = 1 << a.log_n, 2 << a.log_n, a.ncols
    A, B = mi_stark.Context(0), mi_stark.Context(0)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()      # a context enqueues on the stream that is current when use_torch_stream() is called
    with torch.cuda.stream(sa):
        A.use_torch_stream()
    with torch.cuda.stream(sb):
        B.use_torch_stream()
    trace = A.empty(n * w)
    A.fill_synthetic_2d(trace, n, w, w, 0, 0x5EED0003)
    ext, nodes = A.empty(ne * w), A.empty((2 * ne - 1) * 4)
    chunks = [(c0, min(a.chunk, w - c0)) for c0 in range(0, w, a.chunk)]

    def sync():
        A.sync()
        B.sync()

    def root():
        return [int(x) & 0xFFFFFFFFFFFFFFFF for x in A.to_host(nodes)[-4:]]

    def product_order():
        A.lde(ext, trace, ne, n, w)
        A.linear_hash_rows(nodes, ext, w, ne)
        A.merkle_levels(nodes, ne)

    def chunked(two_streams):
        H = B if two_streams else A
        c0, cw = chunks[0]
        A.lde(ext, trace, ne, n, cw, out_pitch=w, in_pitch=w, out_off=c0, in_off=c0)
        for k, (c0, cw) in enumerate(chunks):
            if two_streams:
                A.sync()                   # chunk k is extended
            H.linear_hash_absorb(nodes, [(ext, c0, cw, w)], ne, k == 0, k + 1 == len(chunks))
            if k + 1 < len(chunks):
                d0, dw = chunks[k + 1]
                A.lde(ext, trace, ne, n, dw, out_pitch=w, in_pitch=w, out_off=d0, in_off=d0)
            if two_streams:
                H.sync()                   # (the sponge state of chunk k + 1 chains on chunk k's)
        sync()
        A.merkle_levels(nodes, ne)

    out = {}
    for name, fn in (("product_order_one_stream", product_order), ("chunked_one_stream", lambda: chunked(False)), ("chunked_two_streams", lambda: chunked(True))):
        fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        sync()
        out[name] = {"ms_per_step": (time.perf_counter() - t0) / a.steps * 1e3, "root": root()}
    roots = {tuple(v["root"]) for v in out.values()}
    res = {"what": "LDE of chunk k+1 beside the leaf absorption of chunk k (two streams) vs back to back", "rows": n, "cols": w, "chunk_cols": a.chunk,
           "ms_per_step": {k: round(v["ms_per_step"], 2) for k, v in out.items()}, "roots_agree": len(roots) == 1}
    if a.log_n == 23 and w == 665 and hasattr(bench, "ROOT_2P23_X665"):
        res["root_is_the_verified_one"] = list(roots)[0] == tuple(bench.ROOT_2P23_X665)
    print(json.dumps(res))
    A.close()
    B.close()


if __name__ == "__main__":
    main()
