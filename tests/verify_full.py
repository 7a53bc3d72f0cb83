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
  (c) leaves: for 2^16 rows (default) -- every 256th row at a scrambled offset, so that all 16 wave residue classes (row & 15),
      all 64 lane positions and all slab phases of the line-ring leaf kernel are hit equally often, from the first row to the
      element offsets above 2^33, plus the first / last rows and the rows around n -- the oracle's linear_hash of the device's
      extended row must equal the device's level-0 digest.  The rows are gathered on the device (the query-opening kernel) and
      hashed by the oracle on all host threads;
  (d) tree: the oracle rebuilds every level above the device's level-0 digests; the whole node array must match, so the
      device root is the oracle's root of those digests.

(a)+(b) pin the extension, (c) ties digests to rows on a stratified sample and (d) pins the tree above the digests; (c) is a
sample because each leaf costs the oracle 84 permutations (2^16 rows: a few seconds on the GPU box's host cores) -- the kernel
that hashes all rows is the same one the small-size tests compare in full.
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


def leaf_sample_rows(n, n_ext, n_rows, rng):
    """Stratified sample: one row out of every n_ext / n_rows consecutive ones at an offset that walks through all residues, plus
    the edge rows."""
    edge = {0, 1, 2, n - 1, n, n + 1, n_ext - 2, n_ext - 1} & set(range(n_ext))
    if n_rows >= n_ext:
        return sorted(range(n_ext))
    stride = n_ext // n_rows
    k = np.arange(n_rows, dtype=np.uint64)
    # offset inside the stride: an odd multiplier mod stride visits every residue of the low bits equally often; the added k // stride
    # term decorrelates it from the stride index
    offs = (k * np.uint64(0x9E3779B1) + k // np.uint64(max(stride, 1))) % np.uint64(stride)
    rows = set(int(v) for v in (k * np.uint64(stride) + offs))
    return sorted(rows | edge | {int(v) for v in rng.integers(0, n_ext, 8)})


def gather_rows(ctx, windows, nodes, n_ext, rows):
    """rows x ncols host matrix of the listed rows of a matrix given as column windows, gathered on the device."""
    idx = np.array(rows, dtype=np.uint64)
    levels = (n_ext - 1).bit_length()
    parts = []
    for (t, off, w, pitch) in windows:
        buf = ctx.empty(len(rows) * (w + 4 * levels))
        ctx.merkle_group_proofs(buf, nodes, t[off:], n_ext, w, idx, pitch=pitch)
        parts.append(ctx.to_host(buf).reshape(len(rows), w + 4 * levels)[:, :w])
    return np.ascontiguousarray(np.concatenate(parts, axis=1))


def verify_lde_merkle(ctx, trace_windows, ext_windows, nodes, n, n_ext, ncols, cols=None, n_rows=1 << 16, seed=2024,
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
        # ---- (c) sampled leaves: rows gathered on the device, leaf digests by the oracle (all host threads), compared with the
        # device's level-0 digests of those rows
        rows = leaf_sample_rows(n, n_ext, n_rows, rng)
        t_c = time.perf_counter()
        mat = gather_rows(ctx, ext_windows, nodes, n_ext, rows)
        got = ctx.to_host(nodes[:4 * n_ext]).reshape(n_ext, 4)[np.array(rows, dtype=np.int64)]
        # the oracle's merkletree hashes every row (OpenMP over rows) before it builds levels: its first 4 * rows words are the leaf
        # digests; pad to a power of two with copies of the first row
        npad = 1 << max(len(rows) - 1, 0).bit_length()
        padded = np.concatenate([mat, np.repeat(mat[:1], npad - len(rows), axis=0)]) if npad > len(rows) else mat
        want = glo.merkletree(padded, ncols, npad)[:4 * len(rows)].reshape(len(rows), 4)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, ("leaf digest", rows[int(bad[0])], bad.size)
        summary["leaf_rows_checked"] = len(rows)
        summary["leaf_rows_residues"] = sorted({r & 15 for r in rows})
        summary["leaf_seconds"] = time.perf_counter() - t_c
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


def verify_gathered(ctx, ext_full, dig_full, root, n, n_ext, ncols, seed, n_rows=1 << 12, log=lambda *_: None):
    """The result of the SHARDED path, gathered on one rank (shard.gather_sharded_result): the extension of the synthetic trace
    `seed` against the oracle through verify_lde_merkle's (a) + (b), sampled leaf digests against the oracle's linear_hash, and the
    sharded root against the oracle's tree over the gathered digests.  Returns a summary (raises on the first mismatch)."""
    trace = ctx.empty(n * ncols)
    ctx.fill_synthetic_2d(trace, n, ncols, ncols, 0, seed)
    ext = ext_full.contiguous().view(-1)
    summary = verify_lde_merkle(ctx, [(trace, 0, ncols, ncols)], [(ext, 0, ncols, ncols)], None, n, n_ext, ncols, log=log)
    rng = np.random.default_rng(seed)
    rows = leaf_sample_rows(n, n_ext, min(n_rows, n_ext), rng)
    dig = ctx.to_host(dig_full.contiguous().view(-1)).reshape(n_ext, 4)
    for r in rows[:: max(1, len(rows) // 256)]:
        assert np.array_equal(glo.linear_hash(ctx.to_host(ext[r * ncols:(r + 1) * ncols])), dig[r]), ("leaf digest of the sharded path", r)
    want_root = glo.merkletree(dig, 4, n_ext)[-4:]
    assert [int(v) for v in want_root] == [int(v) for v in root], "the sharded root is not the oracle's root over the gathered leaf digests"
    summary.update({"leaf_rows_checked": len(rows[:: max(1, len(rows) // 256)]), "root_matches_oracle_tree_over_gathered_digests": True})
    return summary
