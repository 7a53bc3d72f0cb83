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


# ------------------------------------------------------------------ round 5: transient commits, device groups, sparse memory, MI_MULTI_CHECK
@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,G,halo,pitch_extra,grouped,lend", [(10, 37, 2, 2, 0, False, False), (12, 128, 4, 8, 5, True, True), (11, 665, 8, 2, 0, True, False),
                                                                       (9, 9, 4, 128, 0, False, True), (14, 371, 8, 2, 0, False, False), (12, 665, 16, 2, 3, True, True),
                                                                       (10, 5, 8, 2, 0, True, False)])   # fewer columns than shards
def test_transient_commit_writes_every_row_once_and_gives_the_single_device_tree(log_n, ncols, G, halo, pitch_extra, grouped, lend):
    """mi_multi_set_transient (what a row-sharded Starks::genProof asks for): a row image for EVERY shard, tiles written once by a kernel
    of the extending shard into their owners' images (own rows + halo, wrapping), absorbed there at the image's pitch; the tree keeps
    subtrees only.  Root, every leaf digest and the siblings equal the single-device tree; every image holds its shard's rows and halo
    and nothing else; the base-domain section is kept; with and without device groups (shards of one physical device sharing streams,
    tile ring and workspace) and with the group's buffers lent by the caller."""
    n, n_ext, pitch = 1 << log_n, 2 << log_n, ncols + pitch_extra
    trace = glo.splitmix64(0x5EED0900 + log_n, n * ncols).reshape(n, ncols)
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    want = ctx.to_host(ext).reshape(n_ext, ncols)
    want_nodes = ctx.to_host(nodes)
    m = mi_stark.Multi([0] * G, group_same_device=grouped)
    SENT = 0x5A5A5A5A5A5A5A5A
    imgs = [ctx.zeros(n_ext * pitch).fill_(SENT) for _ in range(G)]
    base = ctx.zeros(n * ncols)
    need = mi_stark.Multi.transient_need(n, n_ext, ncols, G)
    region = ctx.zeros(need + 64) if lend else None
    ctx.sync()
    for rep in range(2):                                                # (the second commit takes its buffers from the pool)
        m.set_row_images([t_.data_ptr() for t_ in imgs], pitch, halo)
        m.set_transient()
        if lend:
            region[:] = 0x7777
            ctx.sync()
            m.lend(0, region.data_ptr(), region.numel() * 8)
        t = m.commit(trace.ctypes.data, n, n_ext, ncols, base_ptr=base.data_ptr())
        assert [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
        R = n_ext // G
        for g in range(G):
            assert np.array_equal(t.leaf_digests(g).reshape(-1), want_nodes[4 * g * R:4 * (g + 1) * R]), g
            got = ctx.to_host(imgs[g]).reshape(n_ext, pitch)
            rows = np.zeros(n_ext, dtype=bool)
            rows[g * R:(g + 1) * R] = True
            rows[[(r % n_ext) for r in range((g + 1) * R, (g + 1) * R + halo)]] = True
            assert np.array_equal(got[rows][:, :ncols], want[rows]), g
            assert (got[~rows] == SENT).all() and (got[:, ncols:] == SENT).all(), g
        assert np.array_equal(ctx.to_host(base), trace.reshape(-1))
        if lend:
            assert bool((ctx.to_host(region[:1024]) != 0x7777).any())    # the group's buffers came out of the lent region
        idx = np.array(sorted({0, n_ext - 1, R, R - 1, n_ext // 2}), dtype=np.uint64)
        sib = t.group_proofs(idx, with_values=False)
        ref = ctx.empty(idx.size * (ncols + 4 * (log_n + 1)))
        ctx.merkle_group_proofs(ref, nodes, ext, n_ext, ncols, idx)
        assert np.array_equal(sib[:, ncols:], ctx.to_host(ref).reshape(idx.size, -1)[:, ncols:])
        with pytest.raises(mi_stark.MiStarkError, match="released"):
            t.group_proofs(idx)                                          # no row values in a transient tree: the caller opens them from its images
        st = m.last_stats()
        assert sum(sum(s["bytes_sent_to_shard"]) for s in st["per_shard"]) == 0   # (one physical device: nothing crosses a link)
        t.free()
    # one-shot: the next commit is an ordinary one
    t = m.commit(trace.ctypes.data, n, n_ext, ncols)
    assert [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]] and t.group_proofs(np.array([1], dtype=np.uint64)).shape[0] == 1
    t.free()
    # a transient commit without an image for every shard is refused
    m.set_row_images([imgs[0].data_ptr()] + [0] * (G - 1), pitch, halo)
    m.set_transient()
    if G > 1:
        with pytest.raises(mi_stark.MiStarkError, match="EVERY shard"):
            m.commit(trace.ctypes.data, n, n_ext, ncols)
    m.close(); ctx.close()


@pytest.mark.gpu
def test_sparse_device_memory_backs_only_what_is_asked_for():
    """mi_vmm_*: an address range far larger than the device, physical memory under two parts of it (idempotent, additive, in 64 MiB
    pieces: equal sizes are what this driver's hipMemSetAccess takes at any distance, tools/vmm_probe3.hip), kernels and copies through
    the mapping, the device's free memory as the account."""
    import torch
    ctx = mi_stark.Context(0)
    free0 = torch.cuda.mem_get_info()[0]
    rng = 600 << 30                                                      # 600 GiB of addresses on a 288 GiB device
    base = ctx.vmm_reserve(rng)
    assert ctx.vmm_backed_bytes(base) == 0 and torch.cuda.mem_get_info()[0] > free0 - (64 << 20)
    a_off, b_off = (3 << 20) + 4096, (400 << 30) + 123 * 8                # unaligned on purpose
    a_len, b_len = 70 << 20, (2 << 20) + 8
    ctx.vmm_back(base, a_off, a_len)
    ctx.vmm_back(base, b_off, b_len)
    backed = ctx.vmm_backed_bytes(base)
    P = 64 << 20
    assert backed % P == 0 and a_len + b_len <= backed <= 3 * P + P            # a: two or three pieces, b: one
    ctx.vmm_back(base, a_off + 4096, a_len // 2)                          # inside what is backed: nothing new
    assert ctx.vmm_backed_bytes(base) == backed
    ctx.vmm_back(base, a_off + a_len - 4096, 70 << 20)                    # overlaps the end of what is backed: only the new pieces
    assert backed < ctx.vmm_backed_bytes(base) <= backed + 2 * P
    backed = ctx.vmm_backed_bytes(base)
    data = glo.splitmix64(0xABCD, 1 << 16)
    for off in (a_off, a_off + a_len - (1 << 19), b_off):
        ctx.copy_h2d(base + off, data)
        assert np.array_equal(ctx.copy_d2h(base + off, data.size), data)
    # a kernel through the mapping: a transform whose input and output live in the sparse range
    n, ncols = 1 << 12, 3
    src = glo.splitmix64(0x77, n * ncols)
    ctx.copy_h2d(base + a_off, src)
    L = mi_stark.lib()
    import ctypes
    mi_stark._check(L.mi_ntt_dev(ctx.h, ctypes.c_void_p(base + a_off + (1 << 20)), ctypes.c_uint64(ncols), ctypes.c_void_p(base + a_off), ctypes.c_uint64(ncols),
                                 ctypes.c_uint64(n), ctypes.c_uint64(ncols), ctypes.c_int(0)))
    ctx.sync()
    assert np.array_equal(ctx.copy_d2h(base + a_off + (1 << 20), n * ncols).reshape(n, ncols), glo.ntt(src.reshape(n, ncols), n, ncols))
    ctx.vmm_back(base, 123 << 30, 1)                                      # scattered single pieces, in any order
    ctx.vmm_back(base, 7 << 30, 1)
    ctx.vmm_back(base, 599 << 30, 1 << 20)
    assert ctx.vmm_backed_bytes(base) == backed + 3 * P
    backed += 3 * P
    assert free0 - torch.cuda.mem_get_info()[0] < backed + (512 << 20)    # (plus the context's own tables)
    with pytest.raises(mi_stark.MiStarkError, match="beyond the reserved range"):
        ctx.vmm_back(base, rng - 4096, 1 << 20)
    ctx.vmm_free(base)
    ctx.close()


def _run_py(code, env):
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pre = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n" % (os.path.join(root, "tests"), os.path.join(root, "merlin-zkevm-prover_amd"))
    return subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)


@pytest.mark.gpu
def test_multi_check_refuses_what_only_a_real_multi_gpu_node_would_show():
    """MI_MULTI_CHECK=1: on one GPU every logical shard is device 0 and every pointer is valid everywhere -- the check is what makes a
    wrong-shard operand fail HERE.  A buffer allocated for shard 1 handed to an entry point working for shard 0, mi_multi_copy with the
    wrong owner, a row image that is another shard's memory: refused, with the two shards named; a correct sharded commit with row
    images: many checks, no violation.  (The whole GPU suite runs with the check on: tests/conftest.py.)"""
    code = r'''
import ctypes, numpy as np, torch, glo, mi_stark
ctx = mi_stark.Context(0)            # (torch's HIP runtime first: the library binds to the one that is loaded)
L = mi_stark.lib()
m = mi_stark.Multi([0, 0, 0, 0])
L.mi_dev_alloc.restype = ctypes.c_void_p
c = [m.ctx_handle(g) for g in range(4)]
n = 1 << 10
a1 = L.mi_dev_alloc(c[1], ctypes.c_uint64(n * 8 * 4))          # memory of shard 1
a2 = L.mi_dev_alloc(c[2], ctypes.c_uint64(n * 8 * 4))          # memory of shard 2
u = ctypes.c_uint64
ok = L.mi_geom_seq_dev(c[1], ctypes.c_void_p(a1), u(n), u(1), u(7))                 # shard 1 on its own memory
bad = L.mi_geom_seq_dev(c[2], ctypes.c_void_p(a1), u(n), u(1), u(7))                # shard 2 on shard 1's memory
msg = L.mi_last_error().decode()
assert ok == 0 and bad != 0 and "logical shard 1" in msg and "shard 2" in msg, (ok, bad, msg)
assert L.mi_multi_copy(m.h, ctypes.c_void_p(a2), 2, ctypes.c_void_p(a1), 1, u(64)) == 0        # declared: dst of 2, src of 1
assert L.mi_multi_copy(m.h, ctypes.c_void_p(a2), 1, ctypes.c_void_p(a1), 1, u(64)) != 0        # dst is NOT shard 1's
assert L.mi_multi_copy(m.h, ctypes.c_void_p(a2), 2, ctypes.c_void_p(a1 + 8), 3, u(64)) != 0    # src is NOT shard 3's
# row images: imgs[q] must be shard q's memory
log_n, ncols = 10, 37
nn, ne = 1 << log_n, 2 << log_n
trace = glo.splitmix64(5, nn * ncols)
imgs = [L.mi_dev_alloc(c[g], u(ne * ncols * 8)) for g in range(4)]
m.set_row_images(imgs, ncols, 2); m.set_transient()
t = m.commit(trace.ctypes.data, nn, ne, ncols)
good_root = [int(v) for v in t.root]
t.free()
swapped = [imgs[0], imgs[2], imgs[1], imgs[3]]
m.set_row_images(swapped, ncols, 2); m.set_transient()
try:
    m.commit(trace.ctypes.data, nn, ne, ncols)
    raise SystemExit("a row image in another shard's memory was accepted")
except mi_stark.MiStarkError as e:
    assert "MI_MULTI_CHECK" in str(e), str(e)
st = mi_stark.multi_check_stats()
assert st["enabled"] and st["checks"] > 30 and st["violations"] == 4, st
print("OK", st, good_root)
'''
    r = _run_py(code, {"MI_MULTI_CHECK": "1"})
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    # with the check off the very same wrong-shard call goes through (every pointer IS valid on the one device) -- which is the point of the check
    off = r'''
import ctypes, torch, mi_stark
ctx = mi_stark.Context(0)
L = mi_stark.lib()
m = mi_stark.Multi([0, 0])
L.mi_dev_alloc.restype = ctypes.c_void_p
a1 = L.mi_dev_alloc(m.ctx_handle(1), ctypes.c_uint64(8192))
u = ctypes.c_uint64
assert L.mi_geom_seq_dev(m.ctx_handle(0), ctypes.c_void_p(a1), u(1024), u(1), u(7)) == 0
assert not mi_stark.multi_check_stats()["enabled"]
print("OK")
'''
    r = _run_py(off, {"MI_MULTI_CHECK": "0"})
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_the_suite_runs_with_the_multi_check_on():
    st = mi_stark.multi_check_stats()
    assert st["enabled"], "tests/conftest.py sets MI_MULTI_CHECK=1 for the GPU suite"
    assert st["violations"] == 0, st


def test_commit_buffer_needs_follow_the_plan():
    """What a caller must lend (no GPU needed): a transient commit takes a tile ring of two, two staging buffers and the NTT workspace per
    device GROUP -- 21.5 GB at 2^23 x 665 whatever the number of shards --, a windowed one the shard's tiles and row windows as well
    (36.5 GB at eight shards, 103 GB at two: ADVICE r04's shortfall against the 67 GB stage 1 lends)."""
    import ctypes
    L = mi_stark.lib()
    L.mi_multi_transient_need.restype = ctypes.c_uint64
    L.mi_multi_windowed_need.restype = ctypes.c_uint64
    u = ctypes.c_uint64
    n, ne = 1 << 23, 1 << 24
    t8 = L.mi_multi_transient_need(u(n), u(ne), u(665), ctypes.c_uint32(8)) * 8
    t2 = L.mi_multi_transient_need(u(n), u(ne), u(665), ctypes.c_uint32(2)) * 8
    assert t8 == t2 and 21.4e9 < t8 < 21.6e9
    w8 = L.mi_multi_windowed_need(u(n), u(ne), u(665), ctypes.c_uint32(8)) * 8
    w2 = L.mi_multi_windowed_need(u(n), u(ne), u(665), ctypes.c_uint32(2)) * 8
    assert 36.4e9 < w8 < 36.7e9 and 102e9 < w2 < 104e9 and w2 > 67e9
    small = L.mi_multi_transient_need(u(1 << 10), u(1 << 11), u(37), ctypes.c_uint32(2))
    assert small >= (1 << 17)                                           # (a lent NTT workspace is at least 1 MiB)
    assert L.mi_multi_transient_need(u(n), u(ne), u(665), ctypes.c_uint32(3)) == 0      # not a power of two


def test_mi_stark_devices_is_parsed_strictly(tmp_path):
    """MI_STARK_DEVICES (host/mi_runtime.hpp): digits and commas only -- atoi would read "a,b" as devices 0,0."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "p.cpp"
    src.write_text('#include "mi_runtime.hpp"\n#include <cstdio>\nint main() { int d[64]; int n = mi::parseDevices(d, 64); std::printf("%d:", n); for (int i = 0; i < n; i++) std::printf(" %d", d[i]); std::printf("\\n"); return 0; }\n')
    exe = tmp_path / "p"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "merlin-zkevm-prover_amd", "host"), str(src), "-o", str(exe),
                           "-L", os.path.join(root, "merlin-zkevm-prover_amd"), "-lmi_stark", "-Wl,-rpath," + os.path.join(root, "merlin-zkevm-prover_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    run = lambda v: subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, MI_STARK_DEVICES=v))
    assert run("0,1,2,3").stdout.strip() == "4: 0 1 2 3" and run("7").stdout.strip() == "1: 7" and run("").stdout.strip() == "0:"
    for bad in ("a,b", "0,,1", "0,1,", "0;1", "-1,0", "0 ,1"):
        r = run(bad)
        assert r.returncode != 0 and "MI_STARK_DEVICES" in r.stderr, (bad, r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_transient_commit_from_a_device_section_at_a_wider_pitch():
    """Stages 2-3 of a row-sharded proof: the section is on a device (the image) at a row pitch wider than itself, the row images have the
    section's own pitch inside a wider mirror, the buffers are lent; root, digests, images and halo as from one device."""
    log_n, ncols, pitch, G, halo = 12, 128, 200, 4, 2
    n, n_ext = 1 << log_n, 2 << log_n
    wide = glo.splitmix64(0x5EED0B00, n * pitch).reshape(n, pitch)
    trace = np.ascontiguousarray(wide[:, 40:40 + ncols])
    ctx = mi_stark.Context(0)
    ext, nodes = single_device(ctx, trace, n, n_ext, ncols)
    want = ctx.to_host(ext).reshape(n_ext, ncols)
    want_nodes = ctx.to_host(nodes)
    d_wide = ctx.to_device(wide)
    SENT = 0x3C3C3C3C3C3C3C3C
    imgs = [ctx.zeros(n_ext * pitch).fill_(SENT) for _ in range(G)]
    region = ctx.zeros(mi_stark.Multi.transient_need(n, n_ext, ncols, G) + 64)
    ctx.sync()
    for grouped in (False, True):
        m = mi_stark.Multi([0] * G, group_same_device=grouped)
        for x in imgs:
            x.fill_(SENT)
        ctx.sync()
        m.set_row_images([t_.data_ptr() + 8 * 40 for t_ in imgs], pitch, halo)
        m.set_transient()
        m.lend(0, region.data_ptr(), region.numel() * 8)
        t = m.commit(d_wide.data_ptr() + 8 * 40, n, n_ext, ncols, src_device=0, src_pitch=pitch)
        assert [int(v) for v in t.root] == [int(v) for v in want_nodes[-4:]]
        R = n_ext // G
        for g in range(G):
            assert np.array_equal(t.leaf_digests(g).reshape(-1), want_nodes[4 * g * R:4 * (g + 1) * R]), g
            got = ctx.to_host(imgs[g]).reshape(n_ext, pitch)
            rows = np.zeros(n_ext, dtype=bool)
            rows[g * R:(g + 1) * R] = True
            rows[[(r % n_ext) for r in range((g + 1) * R, (g + 1) * R + halo)]] = True
            assert np.array_equal(got[rows][:, 40:40 + ncols], want[rows]), g
            assert (got[~rows] == SENT).all() and (got[:, :40] == SENT).all() and (got[:, 40 + ncols:] == SENT).all(), g
        t.free(); m.close()
    ctx.close()
