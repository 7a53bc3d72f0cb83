#!/usr/bin/env python3
"""Would the NTT passes hold their time over COLUMN-MAJOR data (r04 next #4b)?  The tile of a pass is 256 strided groups x 32 batch
elements; over a row-major matrix the batch elements are 32 adjacent COLUMNS of one row (256-byte runs), over a column-major one they
would be 32 consecutive ROWS of one column -- which is exactly the tile the existing kernel takes for a single-column transform (TJ = 32,
TCP = 1).  So C single-column transforms over C contiguous columns ARE the column-major passes, with the narrow (8-byte) loads and stores
the single-column form has.  This probe times them against the C-column row-major transform of the same data volume, forward and
inverse, at the LDE's sizes.  Not a product path: a measurement.   python tools/ntt_colmajor_probe.py [--log-n 24] [--cols 96]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--cols", type=int, default=96)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import mi_stark
    ctx = mi_stark.Context(0)
    n, C = 1 << a.log_n, a.cols
    src = ctx.empty(n * C)
    ctx.fill_synthetic(src, n * C, 0x5EED0A00)
    dst = ctx.empty(n * C)
    out = {"log_n": a.log_n, "cols": C, "what": "C single-column transforms over contiguous columns (= the passes over column-major data, narrow loads) against one C-column row-major transform"}
    for inv in (False, True):
        def row_major():
            ctx.ntt(dst, src, n, C, inverse=inv)

        def col_major():
            for c in range(C):
                ctx.ntt(dst, src, n, 1, inverse=inv, dst_off=c * n, src_off=c * n)
        res = {}
        for name, fn in (("row_major_C_columns", row_major), ("column_major_C_single_columns", col_major)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name + "_ms"] = e0.elapsed_time(e1) / a.reps
        res["ratio"] = res["column_major_C_single_columns_ms"] / res["row_major_C_columns_ms"]
        out["inverse" if inv else "forward"] = res
    # ---- the experimental column-major pass (csrc/ntt.hip k_ntt_pass_cm: 32 consecutive rows of one column per tile row, the workgroup walks
    # over the columns with its twiddles in registers, 16-byte accesses along the rows): passes 2.. of the same transform over all C columns at
    # once, checked against the single-column transforms above, timed against the row-major passes (2 of the 3 passes of the row-major time)
    import ctypes
    import numpy as np
    L = mi_stark.lib()
    if a.log_n in (16, 24):
        ref = ctx.empty(n * C)
        for inv in (False, True):
            for c in range(C):
                ctx.ntt(ref, src, n, 1, inverse=inv, dst_off=c * n, src_off=c * n)
            ms = ctypes.c_float(0)
            best = None
            for _ in range(a.reps):
                mi_stark._check(L.mi_dbg_ntt_colmajor_dev(ctx.h, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), ctypes.c_uint64(n), ctypes.c_uint64(C),
                                                          ctypes.c_int(int(inv)), ctypes.byref(ms)))
                best = ms.value if best is None else min(best, ms.value)
            torch.cuda.synchronize()
            same = bool(torch.equal(dst, ref))
            key = "inverse" if inv else "forward"
            passes = a.log_n // 8
            out[key]["column_major_passes_after_the_first_with_shared_twiddles_ms"] = best
            out[key]["row_major_same_passes_ms_estimate"] = out[key]["row_major_C_columns_ms"] * (passes - 1) / passes
            out[key]["experimental_kernel_equals_the_product_transform"] = same
            # ... and the WHOLE transform in the transposed form: row-major source -> first pass (row-major in, column-major out through an LDS
            # transpose: k_ntt_first_rm2cm) -> the column-major passes; result column-major, compared with the row-major transform transposed
            ctx.ntt(ref, src, n, C, inverse=inv)
            want_cm = ref.view(n, C).t().contiguous().view(-1)
            best = None
            for _ in range(a.reps):
                mi_stark._check(L.mi_dbg_ntt_colmajor_dev(ctx.h, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), ctypes.c_uint64(n), ctypes.c_uint64(C),
                                                          ctypes.c_int(int(inv) | 2), ctypes.byref(ms)))
                best = ms.value if best is None else min(best, ms.value)
            torch.cuda.synchronize()
            out[key]["whole_transform_row_major_in_column_major_out_ms"] = best
            out[key]["whole_transform_equals_the_row_major_transform_transposed"] = bool(torch.equal(dst, want_cm))
            del want_cm
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
