"""The N>1 orchestration (merlin-zkevm-prover_amd/shard.py) on CPU: world_size 2 and 4 over gloo, with the
CPU oracle standing in for the device ops.  The sharded path must reproduce the single-process root and
tree levels bit-for-bit (column-sharded LDE -> all-to-all -> row-sharded Merkle -> root all-gather)."""
import os, socket, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import glo
from shard import ShardPlan, lde_merkle_sharded, group_proofs_sharded, fri_commit_sharded, merkle_levels_of, exchange_bytes_to_peers, gather_sharded_result


def test_tile_dealing_and_buffer_layouts():
    """Columns are dealt in rounds of equal-width tiles, balanced up to rounding to 8; every rank derives the same number
    of rounds; what a rank sends a peer is what that peer expects; a round's windows are consecutive columns of the row,
    in order; the receive windows do not overlap."""
    p = ShardPlan(n=1 << 23, n_ext=1 << 24, ncols=665, world=8, rank=6)
    assert p.per_rank == 88 and p.round_w == [32, 32, 24] and p.n_rounds == 3
    assert [ShardPlan(n=1 << 23, n_ext=1 << 24, ncols=665, world=8, rank=r).my_cols for r in range(8)] == [88] * 6 + [73, 64]
    assert [p.width(2, r) for r in range(8)] == [24] * 6 + [9, 0] and p.max_cols == 88 and p.my_tile_cols() == [(192, 32), (448, 32), (656, 9)]
    assert ShardPlan(n=1 << 23, n_ext=1 << 24, ncols=665, world=1, rank=0).round_w == [32] * 21        # one GPU: 20 x 32 + 25
    assert ShardPlan(n=1 << 23, n_ext=1 << 24, ncols=665, world=2, rank=1).round_w == [32] * 10 + [16]
    assert ShardPlan(n=8, n_ext=16, ncols=7, world=2, rank=1, tile=8).my_cols == 0        # fewer columns than one tile
    for (ncols, world, tile) in ((665, 8, 32), (665, 4, 32), (665, 2, 32), (70, 2, 8), (13, 4, 8), (128, 8, 32), (5, 4, 8), (64, 2, 8),
                                 (100, 8, 8), (371, 8, 32), (6, 2, 32)):
        plans = [ShardPlan(n=64, n_ext=128, ncols=ncols, world=world, rank=r, tile=tile) for r in range(world)]
        assert len({q.n_rounds for q in plans}) == 1 and sum(q.my_cols for q in plans) == ncols
        assert max(q.my_cols for q in plans) == plans[0].max_cols <= -(-ncols // (8 * world)) * 8     # balanced
        for q in plans:
            col = 0
            seen = []
            for k in range(q.n_rounds):
                wins = q.windows(k)
                for i, (name, off, w, pitch) in enumerate(wins):
                    assert pitch == w and w > 0
                    assert w % 8 == 0 or (k == q.n_rounds - 1 and i == len(wins) - 1)   # zero padding only at the row's end
                    seen.append((name, off, off + q.rows_per_rank * w if name == "recv" else None))
                    col += w
                assert col == min(ncols, world * sum(q.round_w[:k + 1]))     # a contiguous prefix of the row after every round
                for peer in plans:
                    if peer.rank != q.rank:
                        assert q.send_block(k, peer.rank)[1] == peer.recv_window(k, q.rank)[1]
                        assert q.send_block(k, peer.rank)[0] + q.send_block(k, peer.rank)[1] <= q.ext_elems()
            assert col == ncols and [w for (_, _, w, _) in q.row_windows()] == [w for k in range(q.n_rounds) for (_, _, w, _) in q.windows(k)]
            recv = sorted((a, b) for (nm, a, b) in seen if nm == "recv")
            assert all(recv[i][1] <= recv[i + 1][0] for i in range(len(recv) - 1)) and (not recv or recv[-1][1] <= q.recv_elems())
            loc = 0
            for k in range(q.n_rounds):                                       # my tiles sit side by side in my trace shard
                if q.width(k, q.rank):
                    assert q.local_col(k) == loc
                    loc += q.width(k, q.rank)


class OracleOps:
    """CPU stand-in for the device ops (test infrastructure): int64 torch tensors as u64 containers."""

    @staticmethod
    def _np(t):
        return t.numpy().view(np.uint64)

    @staticmethod
    def lde(out, inp, n_ext, n, ncols, out_pitch=None, in_pitch=None, out_off=0, in_off=0, chunk=0):
        out_pitch, in_pitch = out_pitch or ncols, in_pitch or ncols
        src = OracleOps._np(inp)
        cols = np.stack([src[in_off + r * in_pitch:in_off + r * in_pitch + ncols] for r in range(n)])
        o = glo.extend_pol(np.ascontiguousarray(cols), n_ext, n, ncols)
        dst = OracleOps._np(out)
        for r in range(n_ext):
            dst[out_off + r * out_pitch:out_off + r * out_pitch + ncols] = o[r]

    @staticmethod
    def absorb(digests, windows, nrows, first, final, chunk=0):
        """The streaming sponge of mi_linear_hash_absorb_dev, restated with the oracle's permutation."""
        d = OracleOps._np(digests)
        for r in range(nrows):
            cap = np.zeros(4, dtype=np.uint64) if first else d[r * 4:r * 4 + 4].copy()
            for wi, (t, off, w, pitch) in enumerate(windows):
                v = OracleOps._np(t)[off + r * pitch:off + r * pitch + w]
                assert w % 8 == 0 or (final and wi == len(windows) - 1)
                for c in range(0, w, 8):
                    st = np.zeros(12, dtype=np.uint64)
                    blk = v[c:c + 8]
                    st[:len(blk)] = blk
                    st[8:] = cap
                    cap = glo.perm(st)[:4]
            d[r * 4:r * 4 + 4] = cap

    @staticmethod
    def merkle_build(nodes, src, ncols, nrows):
        OracleOps._np(nodes)[:(2 * nrows - 1) * 4] = glo.merkletree(OracleOps._np(src)[:nrows * ncols].copy(), ncols, nrows)

    @staticmethod
    def merkle_levels(nodes, nleaves):
        v = OracleOps._np(nodes)
        lvl, n = 0, nleaves
        while n > 1:
            for i in range(n // 2):
                inp = np.zeros(12, dtype=np.uint64)
                inp[:8] = v[lvl + i * 8:lvl + i * 8 + 8]
                v[lvl + (n + i) * 4:lvl + (n + i) * 4 + 4] = glo.perm(inp)[:4]
            lvl += n * 4
            n //= 2


    # ---- query openings and FRI (stand-ins for mi_merkle_group_proofs_dev / mi_fri_fold*_dev / mi_fri_transpose_dev)
    @staticmethod
    def zeros(k):
        return torch.zeros(k, dtype=torch.int64)

    @staticmethod
    def to_host(t):
        return t.numpy().view(np.uint64).copy()

    @staticmethod
    def gather_rows(windows, rows):
        out = []
        for r in rows:
            out.append(np.concatenate([OracleOps._np(t)[off + r * pitch:off + r * pitch + w] for (t, off, w, pitch) in windows]))
        return torch.from_numpy(np.stack(out).view(np.int64))

    @staticmethod
    def merkle_paths(nodes, height, idx):
        v = OracleOps._np(nodes)
        lv = merkle_levels_of(height)
        out = np.zeros((len(idx), 4 * lv), dtype=np.uint64)
        for j, i in enumerate(idx):
            off, n = 0, height
            for l in range(lv):
                out[j, 4 * l:4 * l + 4] = v[off + ((i >> l) ^ 1) * 4:off + ((i >> l) ^ 1) * 4 + 4]
                off += n * 4
                n //= 2
        return torch.from_numpy(out.view(np.int64))

    @staticmethod
    def fri_fold(nxt, pol, prev_bits, cur_bits, nbits_ext, x):
        OracleOps._np(nxt)[:3 << cur_bits] = glo.fri_fold(OracleOps._np(pol)[:3 << prev_bits], prev_bits, cur_bits, nbits_ext, x).reshape(-1)

    @staticmethod
    def fri_fold_range(nxt, pol, prev_bits, cur_bits, nbits_ext, x, g0, cnt):
        full = glo.fri_fold(OracleOps._np(pol)[:3 << prev_bits], prev_bits, cur_bits, nbits_ext, x).reshape(-1)
        OracleOps._np(nxt)[3 * g0:3 * (g0 + cnt)] = full[3 * g0:3 * (g0 + cnt)]

    @staticmethod
    def fri_transpose(src, pol, degree, tbits):
        OracleOps._np(src)[:3 * degree] = glo.fri_transpose(OracleOps._np(pol)[:3 * degree], degree, tbits)


def _worker(rank, world, port, n, ncols, q):
    import shard
    if ncols == 70:
        shard.MAX_MSG_BYTES = 8 * 100      # many message rounds per tile (the production cap is 256 MiB)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_ext = 2 * n
    plan = ShardPlan(n=n, n_ext=n_ext, ncols=ncols, world=world, rank=rank, tile=8)
    full = glo.splitmix64(0x5EED0003, n * ncols).reshape(n, ncols)
    shard_cols = [full[:, c0:c0 + w] for (c0, w) in plan.my_tile_cols()]
    shard_np = np.ascontiguousarray(np.concatenate(shard_cols, axis=1)) if shard_cols else np.zeros((n, 0), dtype=np.uint64)
    trace = torch.from_numpy(shard_np.view(np.int64).reshape(-1).copy())
    z = lambda k: torch.zeros(max(k, 1), dtype=torch.int64)
    bufs = {"ext": z(plan.ext_elems()), "nodes": z((2 * plan.rows_per_rank - 1) * 4), "recv": z(plan.recv_elems()),
            "roots": z((2 * world - 1) * 4)}
    root = lde_merkle_sharded(plan, OracleOps, dist, trace, bufs)
    # query openings over the row-sharded tree: rows of every rank, first / last rows of shards, a repeated index
    idx = sorted({0, 1, n_ext - 1, plan.rows_per_rank - 1, plan.rows_per_rank % n_ext, (3 * n_ext) // 4, 5 % n_ext}) + [1]
    proofs = group_proofs_sharded(plan, OracleOps, dist, bufs, idx).numpy().view(np.uint64).copy()
    # FRI commit with the first fold sharded by output index (min_per_rank lowered so that the small test sizes shard)
    fb = 10
    pol = torch.from_numpy(glo.splitmix64(0xF00D, 3 << fb).view(np.int64).copy())
    tr = glo.Transcript()
    tr.put(root.numpy().view(np.uint64))
    final, trees, chal = fri_commit_sharded(world, rank, OracleOps, dist, tr, pol, [fb, fb - 3, fb - 5, fb - 7], fb, min_per_rank=8)
    fri = (final.numpy().view(np.uint64)[:3 << (fb - 7)].copy(), [t[0].numpy().view(np.uint64)[-4:].copy() for t in trees], tr.get_fields1())
    # the diagnostics bench.py prints for N > 1: what this rank sends each peer per step, and the whole result gathered on rank 0
    sent = exchange_bytes_to_peers(plan)
    ext_full, dig_full = gather_sharded_result(plan, OracleOps, dist, bufs)
    gathered = None if rank else (ext_full.numpy().view(np.uint64).copy(), dig_full.numpy().view(np.uint64).copy())
    q.put((rank, root.numpy().view(np.uint64).copy(), bufs["nodes"].numpy().view(np.uint64)[:plan.rows_per_rank * 4].copy(), idx, proofs, fri, sent, gathered))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,ncols", [(2, 7), (4, 13), (2, 70), (4, 65), (2, 64), (8, 100)])
def test_sharded_path_reproduces_single_process_tree(world, ncols):
    n = 64
    full = glo.splitmix64(0x5EED0003, n * ncols).reshape(n, ncols)
    ext = glo.extend_pol(full, 2 * n, n, ncols)
    nodes = glo.merkletree(ext, ncols, 2 * n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ncols, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = 2 * n // world
    # single-process FRI commit of the same polynomial with the same transcript
    fb = 10
    tr = glo.Transcript()
    tr.put(nodes[-4:])
    pol = torch.from_numpy(glo.splitmix64(0xF00D, 3 << fb).view(np.int64).copy())
    want_final, want_trees, _ = fri_commit_sharded(1, 0, OracleOps, None, tr, pol, [fb, fb - 3, fb - 5, fb - 7], fb)
    want_fri = (want_final.numpy().view(np.uint64)[:3 << (fb - 7)], [t[0].numpy().view(np.uint64)[-4:] for t in want_trees], tr.get_fields1())
    sent_matrix = [r[6] for r in res]
    for a in range(world):                                                   # what a sends b is b's slice of a's columns, and nothing to itself
        assert sent_matrix[a][a] == 0
        pa = ShardPlan(n=n, n_ext=2 * n, ncols=ncols, world=world, rank=a, tile=8)
        assert all(sent_matrix[a][b] == 8 * rows * pa.my_cols for b in range(world) if b != a)
    g_ext, g_dig = res[0][7]
    assert np.array_equal(g_ext, ext) and np.array_equal(g_dig.reshape(-1), nodes[:2 * n * 4])     # rank 0 holds the whole extension and level 0
    assert all(r[7] is None for r in res[1:])
    for rank, root, leaves, idx, proofs, fri, _sent, _g in res:
        assert np.array_equal(root, nodes[-4:]), rank                       # every rank ends with the global root
        assert np.array_equal(leaves, nodes[rank * rows * 4:(rank + 1) * rows * 4])   # and owns its slice of level 0
        for j, i in enumerate(idx):                                         # every rank holds every opening = the single-process one
            assert np.array_equal(proofs[j], glo.merkle_group_proof(nodes, ext, 2 * n, ncols, i)), (rank, i)
            assert glo.merkle_verify(nodes[-4:], proofs[j][:ncols], proofs[j][ncols:], i)
        assert np.array_equal(fri[0], want_fri[0]) and all(np.array_equal(a, b) for a, b in zip(fri[1], want_fri[1])) and fri[2] == want_fri[2], rank


def test_world_one_pipelined_path_without_peers():
    """always_exchange at world 1: every window is read in place, nothing is sent; same root as the plain build."""
    n, ncols = 32, 21
    full = glo.splitmix64(0x5EED0003, n * ncols)
    plan = ShardPlan(n=n, n_ext=2 * n, ncols=ncols, world=1, rank=0, tile=8)
    z = lambda k: torch.zeros(k, dtype=torch.int64)
    bufs = {"ext": z(plan.ext_elems()), "recv": z(plan.recv_elems()), "nodes": z((4 * n - 1) * 4), "roots": z(4)}
    root = lde_merkle_sharded(plan, OracleOps, None, torch.from_numpy(full.view(np.int64).copy()), bufs, always_exchange=True)
    want = glo.merkletree(glo.extend_pol(full.reshape(n, ncols), 2 * n, n, ncols), ncols, 2 * n)
    assert np.array_equal(root.numpy().view(np.uint64), want[-4:])
    assert np.array_equal(bufs["nodes"].numpy().view(np.uint64), want)


def test_world_one_is_plain_merkle_build():
    n, ncols = 32, 5
    full = glo.splitmix64(0x5EED0003, n * ncols)
    plan = ShardPlan(n=n, n_ext=2 * n, ncols=ncols, world=1, rank=0)
    z = lambda k: torch.zeros(k, dtype=torch.int64)
    bufs = {"ext": z(2 * n * ncols), "nodes": z((4 * n - 1) * 4)}
    root = lde_merkle_sharded(plan, OracleOps, None, torch.from_numpy(full.view(np.int64).copy()), bufs)
    want = glo.merkletree(glo.extend_pol(full.reshape(n, ncols), 2 * n, n, ncols), ncols, 2 * n)
    assert np.array_equal(root.numpy().view(np.uint64), want[-4:])
