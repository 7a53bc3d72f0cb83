"""Full-size verification of an LDE + Merkle tree held in HBM against the CPU oracle (TEST INFRASTRUCTURE: used by
tests/test_gpu_fullsize.py and by `bench.py --verify` outside the timed region; never by the product).

A 2^24 x 665 extension is 89 GB: the oracle cannot recompute it in a test's time.  What it CAN do in about a minute,
and what together pins every value of the device result to the oracle:

  (a) whole columns: the oracle re-extends sampled input columns (chunk / tile edges) and every one of the 2^24 outputs
      of each must match bit for bit;
  (b) every column at every row, through linearity: with random coefficients r_c, the oracle's extension of the single
      column sum_c r_c trace[:, c] must equal sum_c r_c ext[:, c] at all rows (a wrong value anywhere in ext survives with
      probability 2^-64).  The two combinations are formed on the device by mi_dbg_lincomb_cols_dev, which is itself
      checked against Python integers on sampled rows of both matrices;
  (c) leaves: for sampled rows (first, last, around n, random) the oracle's linear_hash of the device's extended row must
      equal the device's level-0 digest;
  (d) tree: the oracle rebuilds every level above the device's level-0 digests; the whole node array must match, so the
      device root is the oracle's root of those digests.

(a)+(b) pin the extension, (c) ties digests to rows on a sample and (d) pins the tree above the digests; (c) is a sample
because each leaf costs the oracle 84 permutations -- the kernel that hashes all rows is the same one the small-size tests
compare in full.
"""
import time

import numpy as np

import glo

P = glo.P


def _window_of(windows, c):
    c0 = 0
    for (t, off, w, pitch) in windows:
        if c < c0 + w:
            return t, off + (c - c0), pitch
        c0 += w
    raise IndexError(c)


def pull_column(ctx, windows, nrows, c):
    t, off, pitch = _window_of(windows, c)
    tmp = ctx.empty(nrows)
    ctx.copy_2d(tmp, t, nrows, 1, 1, pitch, src_off=off)
    return ctx.to_host(tmp)


def pull_row(ctx, windows, r):
    return np.concatenate([ctx.to_host(t[off + r * pitch: off + r * pitch + w]) for (t, off, w, pitch) in windows])


def lincomb(ctx, windows, nrows, coef_dev):
    out = ctx.empty(nrows)
    c0 = 0
    for i, (t, off, w, pitch) in enumerate(windows):
        ctx.dbg_lincomb_cols(out, t, nrows, w, coef_dev, pitch=pitch, src_off=off, coef_off=c0, accumulate=(i > 0))
        c0 += w
    return ctx.to_host(out)


def verify_lde_merkle(ctx, trace_windows, ext_windows, nodes, n, n_ext, ncols, cols=None, n_rows=64, seed=2024,
                      check_tree=True, log=lambda *_: None):
    """trace_windows / ext_windows: the n x ncols trace and the n_ext x ncols extension as column windows in column
    order, [(device tensor, element offset, width, pitch)] (a plain matrix is one window).  nodes: device node array
    ((2 n_ext - 1) * 4) or None to skip (c)/(d).  Raises AssertionError on the first mismatch; returns a summary."""
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    summary = {"n": n, "n_ext": n_ext, "ncols": ncols}
    if cols is None:
        cols = [0, 1, 31, 32, 95, 96, 191, 192, ncols - 2, ncols - 1]
    cols = sorted(set(c for c in cols if 0 <= c < ncols))

    # ---- (b) lincomb columns of both matrices (device) + their validation on sampled rows (Python ints)
    coef = glo.rand_fe(rng, ncols)
    coef_dev = ctx.to_device(coef)
    t_comb = lincomb(ctx, trace_windows, n, coef_dev)
    e_comb = lincomb(ctx, ext_windows, n_ext, coef_dev)
    ci = [int(v) for v in coef]
    for (wins, comb, nr) in ((trace_windows, t_comb, n), (ext_windows, e_comb, n_ext)):
        for r in sorted({0, 1, nr // 2 - 1, nr // 2, nr - 1} | {int(v) for v in rng.integers(0, nr, 5)}):
            row = pull_row(ctx, wins, r)
            assert sum(a * int(b) for a, b in zip(ci, row)) % P == int(comb[r]), ("lincomb kernel", nr, r)

    # ---- (a) + (b): one oracle call extends the sampled columns and the combination column
    inp = np.empty((n, len(cols) + 1), dtype=np.uint64)
    for j, c in enumerate(cols):
        inp[:, j] = pull_column(ctx, trace_windows, n, c)
    inp[:, -1] = t_comb
    want = glo.extend_pol(inp, n_ext, n, len(cols) + 1)
    for j, c in enumerate(cols):
        got = pull_column(ctx, ext_windows, n_ext, c)
        bad = np.nonzero(got != want[:, j])[0]
        assert bad.size == 0, ("extended column differs from the oracle", c, int(bad[0]), bad.size)
    bad = np.nonzero(e_comb != want[:, -1])[0]
    assert bad.size == 0, ("linear combination of all extended columns differs from the oracle's extension of the "
                           "combined trace column", int(bad[0]), bad.size)
    summary["columns_checked_in_full"] = cols
    summary["all_columns_all_rows_lincomb"] = True
    log(f"verify: {len(cols)} whole columns + random combination of all {ncols} columns match the oracle "
        f"({time.perf_counter() - t0:.1f} s)")

    if nodes is not None:
        # ---- (c) sampled leaves
        rows = sorted({0, 1, 2, n - 1, n, n + 1, n_ext - 2, n_ext - 1} | {int(v) for v in rng.integers(0, n_ext, max(0, n_rows - 8))})
        for r in rows:
            row = pull_row(ctx, ext_windows, r)
            assert np.array_equal(glo.linear_hash(row), ctx.to_host(nodes[4 * r: 4 * r + 4])), ("leaf digest", r)
        summary["leaf_rows_checked"] = len(rows)
        # ---- (d) the tree above the device's level-0 digests (linear_hash of 4 values is a copy: merkletree over
        # the digests as a 4-column source rebuilds exactly the upper levels)
        if check_tree:
            dev_nodes = ctx.to_host(nodes)
            want_nodes = glo.merkletree(dev_nodes[:4 * n_ext].reshape(n_ext, 4), 4, n_ext)
            assert np.array_equal(dev_nodes, want_nodes), "tree levels above the leaf digests differ from the oracle"
            summary["tree_levels_match_oracle"] = True
            summary["root"] = [int(v) for v in dev_nodes[-4:]]
        log(f"verify: {len(rows)} leaf digests" + (" + every upper tree level" if check_tree else "") +
            f" match the oracle ({time.perf_counter() - t0:.1f} s)")
    summary["seconds"] = time.perf_counter() - t0
    return summary
