"""One process, several devices (csrc/multi.hip): the stage commit sharded over G shards behind the C ABI -- the form a
one-process Prover can use -- against the single-device path, on ONE GPU with G logical shards mapped to device 0 (peer copies
degenerate to device copies; every other step is the code an 8-GPU node runs).  Root, every leaf digest, the openings (values and
siblings) and the row-major image must equal the single-device result bit for bit, and the single-device result is the oracle's
(tests/test_gpu_parity.py, tests/test_gpu_fullsize.py)."""
import numpy as np
import pytest

import glo
import mi_stark


def single_device(ctx, trace, n, n_ext, ncols):
    d_in = ctx.to_device(trace)
    ext, nodes = ctx.empty(n_ext * ncols), ctx.empty((2 * n_ext - 1) * 4)
    ctx.lde(ext, d_in, n_ext, n, ncols)
    ctx.merkle_build(nodes, ext, ncols, n_ext)
    ctx.sync()
    return ext, nodes


def test_multi_api_is_exported_and_refuses_without_a_gpu():
    import ctypes
    import torch
    L = mi_stark.lib()
    for s in ("mi_multi_create", "mi_multi_commit", "mi_multi_group_proofs", "mi_multi_last_stats"):
        assert hasattr(L, s)
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        devs = (ctypes.c_int * 2)(0, 0)
        assert L.mi_multi_create(ctypes.byref(h), devs, 2) == -1 and b"no CPU fallback" in L.mi_last_error()


def test_single_process_plan_is_the_torch_distributed_plan():
    """csrc/multi.hip and shard.py are two forms of ONE plan: the same rounds, the same tile of every shard in every round,
    for widths around every boundary (fewer columns than shards, one over a tile, the zkEVM's 665 / 128 / 371 / 6)."""
    import ctypes
    from shard import ShardPlan
    L = mi_stark.lib()
    L.mi_multi_plan_debug.restype = ctypes.c_int64
    buf = np.zeros(4096, dtype=np.uint64)
    for G in (1, 2, 4, 8, 16):
        for ncols in list(range(5, 80)) + [96, 127, 128, 129, 255, 256, 257, 371, 664, 665, 666, 1024]:
            n, n_ext = 1 << 12, 1 << 13
            got = L.mi_multi_plan_debug(ctypes.c_uint64(n), ctypes.c_uint64(n_ext), ctypes.c_uint64(ncols), ctypes.c_uint32(G),
                                        buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ctypes.c_uint64(buf.size))
            assert got > 0
            p = ShardPlan(n=n, n_ext=n_ext, ncols=ncols, world=G, rank=0)
            assert (int(buf[0]), int(buf[1]), int(buf[2])) == (p.n_rounds, p.per_rank, p.rows_per_rank), (G, ncols)
            k2 = 3
            for k in range(p.n_rounds):
                for g in range(G):
                    c0, w = p.round_cols(k, g)
                    assert int(buf[k2 + 1]) == w and (w == 0 or int(buf[k2]) == c0), (G, ncols, k, g)
                    k2 += 2


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,G", [(10, 37, 2), (12, 665, 4), (13, 96, 8), (11, 6, 2), (14, 371, 8), (18, 665, 2), (18, 665, 4), (18, 665, 8),
                                           (9, 5, 8), (10, 9, 4), (12, 263, 2), (11, 64, 16), (10, 129, 1)])   # fewer columns than shards, one column over a tile, 16 shards, one shard
def test_sharded_commit_from_host_equals_the_single_device_tree(log_n, ncols, G):
    n, n_ext = 1 << log_n, 2 << log_n
    trace = glo.splitmix64(0x5EED0500 + log_n, n * ncols).reshape(n, ncols)
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    want_nodes = ctx.to_host(nodes)
    m = mi_stark.Multi([0] * G)
    image = ctx.zeros(n_ext * ncols)
    base = ctx.zeros(n * ncols)
    ctx.sync()
    t = m.commit(trace.ctypes.data, n, n_ext, ncols, image_ptr=image.data_ptr(), base_ptr=base.data_ptr())
    assert [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
    assert t.shards == G and t.rows_per_shard == n_ext // G
    # every leaf digest, shard by shard
    for g in range(G):
        assert np.array_equal(t.leaf_digests(g).reshape(-1), want_nodes[4 * g * t.rows_per_shard:4 * (g + 1) * t.rows_per_shard]), g
    # the row-major image of the extension and the kept base-domain section
    assert np.array_equal(ctx.to_host(image), ctx.to_host(ext))
    assert np.array_equal(ctx.to_host(base), trace.reshape(-1))
    # openings: first / last row of every shard and random rows
    rng = np.random.default_rng(G)
    idx = np.array(sorted({0, n_ext - 1} | {g * t.rows_per_shard for g in range(G)} | {(g + 1) * t.rows_per_shard - 1 for g in range(G)} |
                          {int(v) for v in rng.integers(0, n_ext, size=24)}), dtype=np.uint64)
    got = t.group_proofs(idx)
    want = ctx.empty(idx.size * got.shape[1])
    ctx.merkle_group_proofs(want, nodes, ext, n_ext, ncols, idx)
    assert np.array_equal(got.reshape(-1), ctx.to_host(want))
    for q, i in enumerate(idx[:6]):                                     # ... and the oracle accepts them against the root
        assert glo.merkle_verify(t.root, got[q][:ncols], got[q][ncols:], int(i))
    # siblings alone, after the row buffers went back
    t.release_rows()
    sib = t.group_proofs(idx, with_values=False)
    assert np.array_equal(sib[:, ncols:], got[:, ncols:]) and not sib[:, :ncols].any()
    st = m.last_stats()
    assert len(st["per_shard"]) == G and all(s["absorb_ms"] > 0 for s in st["per_shard"]) and st["per_shard"][0]["lde_ms"] > 0   # (a shard may be dealt no columns)
    sent = sum(sum(s["bytes_sent_to_shard"]) for s in st["per_shard"])
    assert sent == 8 * n_ext * ncols * (G - 1) // G                     # every element leaves its shard for exactly (G - 1) / G of the rows
    t.free(); m.close(); ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,G,pitch_extra", [(12, 665, 4, 0), (13, 96, 8, 0), (14, 371, 2, 24), (10, 9, 4, 3), (16, 128, 8, 0)])
def test_a_page_locked_host_source_is_read_in_place_by_the_shards_dma(log_n, ncols, G, pitch_extra):
    """A page-locked witness (mi_host_register: what Starks does with pAddress's cm1_n when MI_STARK_DEVICES names several devices) is not
    packed by host threads: each shard's DMA engines read its tile's columns out of the caller's rows (strided 2-D copies, two streams per
    shard).  Same root, digests and image as the packed form and as one device; a pageable source still takes the packed form, and
    forcing the strided form on it is refused."""
    n, n_ext, pitch = 1 << log_n, 2 << log_n, ncols + pitch_extra
    raw = np.zeros(n * pitch + 2048, dtype=np.uint64)                    # (page-locking works on whole pages: a page-aligned range inside one array)
    skip = (-raw.ctypes.data % 4096) // 8 + (5 if pitch_extra else 0)    # ... or not: mi_host_register widens the range to whole pages
    buf = raw[skip:skip + (n * pitch + 511) // 512 * 512]
    view = buf[:n * pitch].reshape(n, pitch)
    view[:, :ncols] = glo.splitmix64(0x5EED0600 + log_n, n * ncols).reshape(n, ncols)
    view[:, ncols:] = 0xDEAD                                            # never read
    trace = np.ascontiguousarray(view[:, :ncols])
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    want_nodes = ctx.to_host(nodes)
    m = mi_stark.Multi([0] * G)
    image, base = ctx.zeros(n_ext * ncols), ctx.zeros(n * ncols)
    ctx.sync()
    # pageable: packed
    t = m.commit(buf.ctypes.data, n, n_ext, ncols, src_pitch=pitch)
    assert m.last_upload_mode() == "packed" and [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
    t.free()
    m.set_upload_mode(1)
    with pytest.raises(mi_stark.MiStarkError, match="not page-locked"):
        m.commit(buf.ctypes.data, n, n_ext, ncols, src_pitch=pitch)
    m.set_upload_mode(-1)
    ctx.host_register(buf.ctypes.data, buf.nbytes)
    try:
        t = m.commit(buf.ctypes.data, n, n_ext, ncols, src_pitch=pitch)  # auto: logical shards share ONE device and its one link -> packed
        assert m.last_upload_mode() == "packed" and [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
        t.free()
        m.set_upload_mode(1)                                            # ... what shards on two or more devices take by themselves
        t = m.commit(buf.ctypes.data, n, n_ext, ncols, src_pitch=pitch, image_ptr=image.data_ptr(), base_ptr=base.data_ptr())
        assert m.last_upload_mode() == "strided"
        assert [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
        for g in range(G):
            assert np.array_equal(t.leaf_digests(g).reshape(-1), want_nodes[4 * g * t.rows_per_shard:4 * (g + 1) * t.rows_per_shard]), g
        assert np.array_equal(ctx.to_host(image), ctx.to_host(ext)) and np.array_equal(ctx.to_host(base), trace.reshape(-1))
        st = m.last_stats()
        assert all(s["host_pack_ms"] == 0 for s in st["per_shard"])     # no host thread touched the data
        t.free()
        m.set_upload_mode(0)                                            # the packed form out of the same page-locked source (A/B runs)
        t = m.commit(buf.ctypes.data, n, n_ext, ncols, src_pitch=pitch)
        assert m.last_upload_mode() == "packed" and [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
        t.free()
    finally:
        ctx.host_unregister(buf.ctypes.data)
    d = ctx.to_device(trace)                                            # a device source says so
    t = m.commit(d.data_ptr(), n, n_ext, ncols, src_device=0)
    assert m.last_upload_mode() == "device" and [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
    t.free(); m.close(); ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,G,halo,pitch_extra", [(10, 37, 2, 2, 0), (12, 128, 4, 8, 5), (11, 665, 8, 2, 0), (9, 9, 4, 128, 0)])
def test_row_images_hold_every_shards_own_rows_and_halo(log_n, ncols, G, halo, pitch_extra):
    """mi_multi_set_row_images: beside the tree, the commit leaves on every shard that asks for it the shard's own rows of the extension and
    the `halo` rows after them (the last shard's wrap to row 0) row-major at their place in a full-height section -- what the row-sharded
    step42ns reads on that device (host/chelpers_steps.hpp) -- and nothing else; one-shot."""
    n, n_ext, pitch = 1 << log_n, 2 << log_n, ncols + pitch_extra
    trace = glo.splitmix64(0x5EED0700 + log_n, n * ncols).reshape(n, ncols)
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    want = ctx.to_host(ext).reshape(n_ext, ncols)
    m = mi_stark.Multi([0] * G)
    SENT = 0x5A5A5A5A5A5A5A5A
    imgs = [None] + [ctx.zeros(n_ext * pitch).fill_(SENT) for _ in range(1, G)]   # (shard 0 is the caller's device: it has the whole image)
    ctx.sync()
    m.set_row_images([0] + [t_.data_ptr() for t_ in imgs[1:]], pitch, halo)
    t = m.commit(trace.ctypes.data, n, n_ext, ncols)
    assert [int(v) for v in t.root] == [int(v) for v in ctx.to_host(nodes)[-4:]]
    R = n_ext // G
    for g in range(1, G):
        got = ctx.to_host(imgs[g]).reshape(n_ext, pitch)
        rows = np.zeros(n_ext, dtype=bool)
        rows[g * R:(g + 1) * R] = True
        rows[[(r % n_ext) for r in range((g + 1) * R, (g + 1) * R + halo)]] = True
        assert np.array_equal(got[rows][:, :ncols], want[rows]), g
        assert (got[~rows] == SENT).all() and (got[:, ncols:] == SENT).all(), g
    t.free()
    t = m.commit(trace.ctypes.data, n, n_ext, ncols)                      # one-shot: the next commit writes no row images
    before = [ctx.to_host(x) for x in imgs[1:]]
    for x in imgs[1:]:
        x.fill_(0)
    ctx.sync()
    t.free()
    t = m.commit(trace.ctypes.data, n, n_ext, ncols)
    assert all(not ctx.to_host(x).any() for x in imgs[1:]) and len(before) == G - 1
    t.free(); m.close(); ctx.close()


@pytest.mark.gpu
def test_sharded_commit_from_a_device_section_equals_the_single_device_tree():
    """Stages 2-4 of a proof: the section is already on a device (the image), at a row pitch wider than itself."""
    log_n, ncols, pitch, G = 12, 128, 200, 4
    n, n_ext = 1 << log_n, 2 << log_n
    wide = glo.splitmix64(0x5EED0600, n * pitch).reshape(n, pitch)
    trace = np.ascontiguousarray(wide[:, 40:40 + ncols])
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    d_wide = ctx.to_device(wide)
    image = ctx.zeros(n_ext * pitch)
    ctx.sync()
    m = mi_stark.Multi([0] * G)
    t = m.commit(d_wide.data_ptr() + 8 * 40, n, n_ext, ncols, src_device=0, src_pitch=pitch, image_ptr=image.data_ptr() + 8 * 40, image_pitch=pitch)
    assert [int(v) for v in t.root] == [int(v) for v in ctx.to_host(nodes)[-4:]]
    img = ctx.to_host(image).reshape(n_ext, pitch)
    assert np.array_equal(img[:, 40:40 + ncols].reshape(-1), ctx.to_host(ext)) and not img[:, :40].any() and not img[:, 40 + ncols:].any()
    assert np.array_equal(t.gather_rows(n_ext // 2 - 3, 7).reshape(-1), ctx.to_host(ext).reshape(n_ext, ncols)[n_ext // 2 - 3:n_ext // 2 + 4].reshape(-1))
    t.free(); m.close(); ctx.close()


@pytest.mark.gpu
def test_two_shards_at_full_size_reproduce_the_verified_root():
    """BASELINE configs[2] at full size (2^23 x 665 -> 2^24) through two logical shards on one GPU, the trace in pageable host memory:
    the root must be the one tests/test_gpu_fullsize.py verifies against the oracle (bench.py's ROOT_2P23_X665), sampled openings must
    verify against it, and the per-shard statistics must account for every byte that changed shards."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 300e9:
        pytest.skip("needs the MI355X's 288 GiB")
    import bench
    log_n, ncols, G = 23, 665, 2
    n, n_ext = 1 << log_n, 2 << log_n
    ctx = mi_stark.Context(0)
    host = torch.empty(n * ncols, dtype=torch.int64)                     # pageable, as a mapped pols file is
    d = ctx.empty(n * ncols)
    ctx.fill_synthetic_2d(d, n, ncols, ncols, 0, 0x5EED0003)             # bench.py's synthetic trace
    host.copy_(d)
    torch.cuda.synchronize()
    del d
    torch.cuda.empty_cache()
    m = mi_stark.Multi([0] * G)
    t = m.commit(host.data_ptr(), n, n_ext, ncols)
    assert [int(v) for v in t.root] == bench.ROOT_2P23_X665
    idx = np.array([0, 1, n_ext // 2 - 1, n_ext // 2, n_ext - 1, 12345678], dtype=np.uint64)
    got = t.group_proofs(idx)
    for q, i in enumerate(idx):
        assert glo.merkle_verify(t.root, got[q][:ncols], got[q][ncols:], int(i))
    st = m.last_stats()
    assert sum(sum(s["bytes_sent_to_shard"]) for s in st["per_shard"]) == 8 * n_ext * ncols // 2
    print(st)
    t.free(); m.close(); ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("big_enough", [True, False])
def test_a_lent_region_serves_the_shard_and_is_returned(big_enough):
    """mi_multi_lend: a caller that plans a device's HBM (Starks: the image fills device 0, which is also shard 0) hands the next commit a
    region that is not live; the shard's row buffers, staging and NTT workspace come out of it (nothing is allocated for that shard) and
    the region is the caller's again after mi_multi_tree_release_rows.  A region that is too small is ignored (the shard allocates)."""
    import torch
    log_n, ncols, G = 14, 160, 2
    n, n_ext = 1 << log_n, 2 << log_n
    trace = glo.splitmix64(0x5EED0700, n * ncols).reshape(n, ncols)
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    want_root = [int(v) for v in ctx.to_host(nodes)[-4:]]
    m = mi_stark.Multi([0] * G)
    per_rank = 80
    need = n_ext * per_rank + (n_ext // G) * G * per_rank + 2 * n * 32 + (2 * n + n_ext) * 32 + 8192   # row buffers + staging + the transforms' workspace (multi.hip)
    region = ctx.zeros(need if big_enough else need // 8)
    region[:] = 0x7777
    ctx.sync()
    free_before = torch.cuda.mem_get_info()[0]
    m.lend(0, region.data_ptr(), region.numel() * 8)
    t = m.commit(trace.ctypes.data, n, n_ext, ncols)
    assert [int(v) for v in t.root] == want_root
    idx = np.array([0, 5, n_ext // 2 - 1, n_ext // 2, n_ext - 1], dtype=np.uint64)
    got = t.group_proofs(idx)
    for q, i in enumerate(idx):
        assert glo.merkle_verify(t.root, got[q][:ncols], got[q][ncols:], int(i))
    used = bool((ctx.to_host(region[:1024]) != 0x7777).any())
    assert used == big_enough                                            # the lent region was written iff it was large enough to serve
    t.release_rows()
    region[:] = 1                                                       # ours again: overwriting it must not disturb the subtrees
    ctx.sync()
    sib = t.group_proofs(idx, with_values=False)
    assert np.array_equal(sib[:, ncols:], got[:, ncols:])
    t2 = m.commit(trace.ctypes.data, n, n_ext, ncols)                    # (the lend was for ONE commit: this one allocates or reuses its pool)
    assert [int(v) for v in t2.root] == want_root and bool((ctx.to_host(region[:1024]) == 1).all())
    t.free(); t2.free(); m.close(); ctx.close()
    assert free_before > 0
