"""The extended sections kept TILE-MAJOR ([rows / 64][columns][64 rows]; include/mi_stark.h: mi_lde_merkle_dev_tiled / _host_tiled,
mi_merkle_group_proofs_tiled_dev, mi_evmap_tiled_dev, mi_untile_dev): the layout Starks::genProof's constraint kernels read in place,
written by the leaf kernel while it absorbs a chunk.  Every entry point against the oracle and against its row-major twin on the same
seeded inputs, bit for bit: the tree, the section's words, the openings, the evaluations."""
import numpy as np
import pytest
import glo

pytestmark = pytest.mark.gpu
P = glo.P


@pytest.fixture(scope="module")
def ctx():
    import mi_stark
    c = mi_stark.Context(0)
    yield c
    c.close()


REV6 = np.array([int(format(i, "06b")[::-1], 2) for i in range(64)])


def tile_major(a, nrows, ncols):
    """element (row, col) at (row // 64 * ncols + col) * 64 + bit-reversal of (row % 64): include/mi_stark.h"""
    return np.asarray(a, dtype=np.uint64).reshape(nrows // 64, 64, ncols).transpose(0, 2, 1)[:, :, REV6].reshape(-1).copy()


@pytest.mark.parametrize("log_n,blow,ncols", [(5, 1, 5), (6, 1, 8), (10, 1, 9), (10, 1, 96), (10, 1, 128), (10, 1, 130), (11, 2, 200), (12, 1, 371), (10, 4, 24), (13, 1, 17)])
def test_lde_merkle_dev_tiled_is_the_row_major_commit_in_another_layout(ctx, log_n, blow, ncols):
    """extendPol + merkelize of a device section (starks.cpp:133-138): the nodes are the oracle's tree over the oracle's extension, the
    section holds that extension tile-major; widths below / at / above a chunk (128) and not a multiple of 8, blowups 2..16, the
    smallest extension a tile allows (64 rows), a source at a wider pitch."""
    rng = np.random.default_rng(1000 + log_n * 7 + ncols)
    n, n_ext = 1 << log_n, 1 << (log_n + blow)
    pitch = ncols + 3
    src = glo.rand_fe(rng, (n, pitch), canonical=False)
    want_ext = glo.extend_pol(np.ascontiguousarray(src[:, :ncols]), n_ext, n, ncols) if n_ext * ncols <= 1 << 21 else None
    d_src = ctx.to_device(src)
    nodes, ext_t = ctx.empty((2 * n_ext - 1) * 4), ctx.to_device(np.full(n_ext * ncols + 8, 0xABCD, dtype=np.uint64))
    ctx.lde_merkle_dev_tiled(nodes, ext_t, d_src, n, n_ext, ncols, src_pitch=pitch)
    # the row-major twin (mi_lde_dev + mi_merkle_build_dev are pinned to the oracle in test_gpu_parity.py; here directly, too)
    ext_r, nodes_r = ctx.empty(n_ext * ncols), ctx.empty((2 * n_ext - 1) * 4)
    ctx.lde(ext_r, d_src, n_ext, n, ncols, in_pitch=pitch)
    ctx.merkle_build(nodes_r, ext_r, ncols, n_ext)
    h_ext = ctx.to_host(ext_r)
    if want_ext is not None:
        assert np.array_equal(h_ext.reshape(n_ext, ncols), want_ext)
    got = ctx.to_host(ext_t)
    assert np.array_equal(got[:n_ext * ncols], tile_major(h_ext, n_ext, ncols)) and np.all(got[n_ext * ncols:] == 0xABCD)
    assert np.array_equal(ctx.to_host(nodes), ctx.to_host(nodes_r))
    if n_ext <= 1 << 11:
        assert np.array_equal(ctx.to_host(nodes), glo.merkletree(h_ext, ncols, n_ext))
    # the way back, whole and a ragged window
    back = ctx.empty(n_ext * ncols)
    ctx.untile(back, ext_t, ncols, n_ext)
    assert np.array_equal(ctx.to_host(back), h_ext)
    r0, nr, c0, nc = 37 % n_ext, min(70, n_ext - 37 % n_ext), ncols // 3, ncols - ncols // 3 - 1
    if nc > 0:
        win = ctx.empty(nr * nc)
        ctx.untile(win, ext_t, ncols, n_ext, col0=c0, row0=r0, nrows=nr, ncols=nc)
        assert np.array_equal(ctx.to_host(win).reshape(nr, nc), h_ext.reshape(n_ext, ncols)[r0:r0 + nr, c0:c0 + nc])


def test_lde_merkle_dev_tiled_with_a_lent_workspace_and_twice_in_a_row(ctx):
    """The compact chunk buffer comes out of the tail of a lent workspace (host/starks.hpp lends a dead section per stage); a second
    commit right behind the first reuses it; without a loan the context's staging allocation serves."""
    import torch
    rng = np.random.default_rng(77)
    n, n_ext, ncols = 1 << 12, 1 << 13, 150
    src = glo.rand_fe(rng, (n, ncols))
    d_src = ctx.to_device(src)
    ext_r, nodes_r = ctx.empty(n_ext * ncols), ctx.empty((2 * n_ext - 1) * 4)
    ctx.lde(ext_r, d_src, n_ext, n, ncols)
    ctx.merkle_build(nodes_r, ext_r, ncols, n_ext)
    want_t, want_nodes = tile_major(ctx.to_host(ext_r), n_ext, ncols), ctx.to_host(nodes_r)
    # 512 MiB: the widest chunk (128 columns: 8 MiB of compact extension + 12 MiB of transform scratch); 14 and 7 MiB: the loan holds
    # narrower chunks only (64, 32 columns) -- the call narrows them instead of allocating
    for loan_mib in (512, 14, 7, 0):
        loan = torch.zeros(loan_mib << 17, dtype=torch.int64, device="cuda") if loan_mib else None
        ctx.lend_workspace(loan)
        for _ in range(2):
            nodes, ext_t = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * ncols)
            ctx.lde_merkle_dev_tiled(nodes, ext_t, d_src, n, n_ext, ncols)
            assert np.array_equal(ctx.to_host(ext_t), want_t) and np.array_equal(ctx.to_host(nodes), want_nodes), loan_mib
        ctx.lend_workspace(None)


@pytest.mark.parametrize("log_n,ncols,pack", [(10, 70, -1), (12, 665, -1), (10, 33, 0), (11, 128, 0), (6, 12, -1)])
def test_lde_merkle_host_tiled_is_the_row_major_stage_one(ctx, log_n, ncols, pack):
    """Stage 1 (starks.cpp:48-59) with both sections tile-major: host trace up in column chunks (packed by host threads, or as strided
    copies), each chunk extended into the compact buffer and absorbed + written tile-major -- the same tree and words as
    mi_lde_merkle_host_keep."""
    import torch
    rng = np.random.default_rng(2000 + ncols)
    n, n_ext = 1 << log_n, 1 << (log_n + 1)
    host = torch.from_numpy(glo.rand_fe(rng, (n, ncols), canonical=False).view(np.int64)).pin_memory()
    ctx.set_host_pack_threads(pack)
    nodes_r, ext_r, base_r = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * ncols), ctx.empty(n * ncols)
    ctx.lde_merkle_host_keep(nodes_r, ext_r, base_r, host.data_ptr(), n, n_ext, ncols)
    nodes, ext_t, base_t = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * ncols), ctx.empty(n * ncols)
    ctx.lde_merkle_host_tiled(nodes, ext_t, base_t, host.data_ptr(), n, n_ext, ncols)
    ctx.set_host_pack_threads(-1)
    assert np.array_equal(ctx.to_host(nodes), ctx.to_host(nodes_r))
    assert np.array_equal(ctx.to_host(ext_t), tile_major(ctx.to_host(ext_r), n_ext, ncols))
    assert np.array_equal(ctx.to_host(base_t), tile_major(ctx.to_host(base_r), n, ncols))
    # the base-domain section row-major beside a tile-major extension
    nodes3, ext3, base3 = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * ncols), ctx.empty(n * (ncols + 2))
    ctx.lde_merkle_host_tiled(nodes3, ext3, base3, host.data_ptr(), n, n_ext, ncols, base_pitch=ncols + 2)
    assert np.array_equal(ctx.to_host(ext3), ctx.to_host(ext_t)) and np.array_equal(ctx.to_host(nodes3), ctx.to_host(nodes_r))
    assert np.array_equal(ctx.to_host(base3).reshape(n, ncols + 2)[:, :ncols], ctx.to_host(base_r).reshape(n, ncols))
    # without a base-domain copy
    nodes2, ext2 = ctx.empty((2 * n_ext - 1) * 4), ctx.empty(n_ext * ncols)
    ctx.lde_merkle_host_tiled(nodes2, ext2, None, host.data_ptr(), n, n_ext, ncols)
    assert np.array_equal(ctx.to_host(nodes2), ctx.to_host(nodes_r)) and np.array_equal(ctx.to_host(ext2), ctx.to_host(ext_t))


def test_tiled_entry_points_refuse_what_they_cannot_lay_out(ctx):
    import mi_stark
    nodes, ext, src = ctx.empty(1024), ctx.empty(4096), ctx.empty(4096)
    with pytest.raises(mi_stark.MiStarkError, match="multiple of 64 rows"):
        ctx.lde_merkle_dev_tiled(nodes, ext, src, 16, 32, 8)        # 32 rows: no tile
    with pytest.raises(mi_stark.MiStarkError, match="more than 4 columns"):
        ctx.lde_merkle_dev_tiled(nodes, ext, src, 64, 128, 4)       # linear_hash copies such rows: nothing is absorbed
    with pytest.raises(mi_stark.MiStarkError, match="outside the section"):
        ctx.untile(ext, src, 8, 128, col0=4, ncols=5)
    with pytest.raises(mi_stark.MiStarkError, match="more values asked for"):
        ctx.merkle_group_proofs_tiled(ext, nodes, src, 8, 128, 9, np.array([1], dtype=np.uint64))


def test_group_proofs_from_a_tile_major_section(ctx):
    rng = np.random.default_rng(15)
    h, w = 1 << 10, 18
    src = glo.rand_fe(rng, (h, w))
    d_src, d_t = ctx.to_device(src), ctx.to_device(tile_major(src, h, w))
    nodes = ctx.empty((2 * h - 1) * 4)
    ctx.merkle_build(nodes, d_src, w, h)
    idx = np.array([0, 1, 63, 64, 65, 511, 512, 1023, 700], dtype=np.uint64)
    stride = w + 4 * 10
    want, got = ctx.empty(idx.size * stride), ctx.empty(idx.size * stride)
    ctx.merkle_group_proofs(want, nodes, d_src, h, w, idx)
    ctx.merkle_group_proofs_tiled(got, nodes, d_t, w, h, w, idx)
    assert np.array_equal(ctx.to_host(got), ctx.to_host(want))
    h_nodes = ctx.to_host(nodes)
    g = ctx.to_host(got).reshape(idx.size, stride)
    for q, i in enumerate(idx):
        assert np.array_equal(g[q], glo.merkle_group_proof(h_nodes, src, h, w, int(i)))


@pytest.mark.parametrize("ext_bits", [1, 2, 0])
def test_evmap_over_tile_major_sections_matches_oracle(ctx, ext_bits):
    """evmap (starks.cpp:555-668) with committed polynomials in tile-major sections beside row-major ones (constants, the quotient):
    base- and extension-valued, at z and at z * w, the same column twice, more evaluations than one group of the kernel, a row count
    that leaves the last wave of a slice ragged."""
    rng = np.random.default_rng(52 + ext_bits)
    n, width, w2 = 1 << 12, 23, 6
    ne = n << ext_bits
    cm, cm2 = glo.rand_fe(rng, (ne, width), canonical=False), glo.rand_fe(rng, (ne, w2))
    q = glo.rand_fe(rng, (ne, 3))
    lev, lpev = glo.rand_fe(rng, (n, 3)), glo.rand_fe(rng, (n, 3))
    d_cm_t, d_cm2_t, d_q = ctx.to_device(tile_major(cm, ne, width)), ctx.to_device(tile_major(cm2, ne, w2)), ctx.to_device(q)
    cols1 = [0, 5, 5, 22, 7, 1, 2, 3, 4]
    pols_h = [(cm, c, 1, width) for c in cols1] + [(cm, 10, 3, width), (q, 0, 3, 3), (cm2, 3, 3, w2), (cm2, 0, 1, w2)]
    pols_d = [(d_cm_t, 64 * c, 1, 0) for c in cols1] + [(d_cm_t, 64 * 10, 3, 0), (d_q, 0, 3, 3), (d_cm2_t, 64 * 3, 3, 0), (d_cm2_t, 0, 1, 0)]
    tile_cols = [width] * len(cols1) + [width, 0, w2, w2]
    prime = [0, 1, 0, 1, 1, 0, 0, 1, 0, 0, 1, 1, 0]
    ev = ctx.empty(len(pols_d) * 3)
    ctx.evmap(ev, pols_d, prime, ctx.to_device(lev), ctx.to_device(lpev), n, ext_bits, tile_cols=tile_cols)
    want = glo.evmap([(a % np.uint64(P), c, d, s) for (a, c, d, s) in pols_h], prime, lev, lpev, n, ext_bits)
    assert np.array_equal(ctx.to_host(ev).reshape(-1, 3), want)
