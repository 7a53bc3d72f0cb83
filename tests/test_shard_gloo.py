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
from shard import ShardPlan, column_partition, lde_merkle_sharded


def test_column_partition():
    assert column_partition(665, 8) == [(0, 84), (84, 83), (167, 83), (250, 83), (333, 83), (416, 83), (499, 83), (582, 83)]
    assert column_partition(6, 4) == [(0, 2), (2, 2), (4, 1), (5, 1)]
    for nc, w in ((665, 1), (665, 2), (128, 8), (3, 2)):
        parts = column_partition(nc, w)
        assert parts[0][0] == 0 and sum(x[1] for x in parts) == nc
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
    p = ShardPlan(n=8, n_ext=16, ncols=7, world=2, rank=1)
    assert (p.col0, p.my_cols) == (4, 3) and p.chunks == [[4], [3]] and p.n_chunks == 1
    assert p.send_block(0, 0) == (0, 8 * 3) and p.send_block(0, 1) == (8 * 3, 8 * 3)
    assert p.recv_slab(0, 0) == (0, 8 * 4) and p.recv_slab(0, 1) == (8 * 4, 8 * 3)


def test_pipeline_chunks_and_exchange_layout():
    """Every rank derives the same number of pipeline chunks; the send blocks tile bufs['ext'] and the receive slabs
    tile bufs['recv'] without gaps or overlaps; a rank with fewer chunks takes part with an empty block."""
    from shard import pipeline_chunks
    assert pipeline_chunks(84, 84) == [32, 32, 20] and pipeline_chunks(83, 84) == [32, 32, 19]
    assert pipeline_chunks(333, 333) == [96, 96, 96, 45] and pipeline_chunks(33, 33) == [32, 1] and pipeline_chunks(32, 33) == [32]
    for (ncols, world) in ((665, 8), (665, 2), (65, 2), (13, 4), (128, 8), (5, 4)):
        plans = [ShardPlan(n=64, n_ext=128, ncols=ncols, world=world, rank=r) for r in range(world)]
        assert len({p.n_chunks for p in plans}) == 1 and plans[0].n_chunks <= 4
        for p in plans:
            assert sum(p.chunks[p.rank]) == p.my_cols
            sent = sorted(p.send_block(k, peer) for k in range(p.n_chunks) for peer in range(world))
            pos = 0
            for off, cnt in sent:
                if cnt:
                    assert off == pos
                    pos += cnt
            assert pos == p.n_ext * p.my_cols
            got = sorted(p.recv_slab(k, peer) for k in range(p.n_chunks) for peer in range(world))
            pos = 0
            for off, cnt in got:
                if cnt:
                    assert off == pos
                    pos += cnt
            assert pos == p.rows_per_rank * ncols
            for k in range(p.n_chunks):   # what I send to a peer is what that peer expects from me
                for peer in plans:
                    assert p.send_block(k, peer.rank)[1] == peer.recv_slab(k, p.rank)[1]


class OracleOps:
    """CPU stand-in for the device ops (test infrastructure): int64 torch tensors as u64 containers."""

    @staticmethod
    def _np(t):
        return t.numpy().view(np.uint64)

    @staticmethod
    def lde(out, inp, n_ext, n, ncols, out_pitch=None, in_pitch=None, out_off=0, in_off=0, chunk=0):
        out_pitch, in_pitch = out_pitch or ncols, in_pitch or ncols
        src = OracleOps._np(inp)[in_off:in_off + (n - 1) * in_pitch + ncols]
        cols = np.stack([src[r * in_pitch:r * in_pitch + ncols] for r in range(n)])
        o = glo.extend_pol(np.ascontiguousarray(cols), n_ext, n, ncols)
        dst = OracleOps._np(out)
        for r in range(n_ext):
            dst[out_off + r * out_pitch:out_off + r * out_pitch + ncols] = o[r]

    @staticmethod
    def copy_2d(dst, src, nrows, ncols, dst_pitch, src_pitch, dst_off=0, src_off=0):
        d, s = OracleOps._np(dst), OracleOps._np(src)
        for r in range(nrows):
            d[dst_off + r * dst_pitch:dst_off + r * dst_pitch + ncols] = s[src_off + r * src_pitch:src_off + r * src_pitch + ncols]

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


def _worker(rank, world, port, n, ncols, q):
    import shard
    if ncols == 70:
        shard.MAX_MSG_BYTES = 8 * 100      # many message rounds per chunk (the production cap is 256 MiB)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_ext = 2 * n
    plan = ShardPlan(n=n, n_ext=n_ext, ncols=ncols, world=world, rank=rank)
    full = glo.splitmix64(0x5EED0003, n * ncols).reshape(n, ncols)
    shard = np.ascontiguousarray(full[:, plan.col0:plan.col0 + plan.my_cols])
    trace = torch.from_numpy(shard.view(np.int64).reshape(-1).copy())
    z = lambda k: torch.zeros(k, dtype=torch.int64)
    bufs = {"ext": z(max(n_ext * plan.max_cols, plan.rows_per_rank * ncols)), "nodes": z((2 * plan.rows_per_rank - 1) * 4),
            "recv": z(plan.rows_per_rank * ncols), "roots": z((2 * world - 1) * 4)}
    root = lde_merkle_sharded(plan, OracleOps, dist, trace, bufs)
    q.put((rank, root.numpy().view(np.uint64).copy(), bufs["nodes"].numpy().view(np.uint64)[:plan.rows_per_rank * 4].copy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,ncols", [(2, 7), (4, 13), (2, 70), (2, 65)])
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
    for rank, root, leaves in res:
        assert np.array_equal(root, nodes[-4:]), rank                       # every rank ends with the global root
        assert np.array_equal(leaves, nodes[rank * rows * 4:(rank + 1) * rows * 4])   # and owns its slice of level 0


def test_world_one_is_plain_merkle_build():
    n, ncols = 32, 5
    full = glo.splitmix64(0x5EED0003, n * ncols)
    plan = ShardPlan(n=n, n_ext=2 * n, ncols=ncols, world=1, rank=0)
    z = lambda k: torch.zeros(k, dtype=torch.int64)
    bufs = {"ext": z(2 * n * ncols), "nodes": z((4 * n - 1) * 4)}
    root = lde_merkle_sharded(plan, OracleOps, None, torch.from_numpy(full.view(np.int64).copy()), bufs)
    want = glo.merkletree(glo.extend_pol(full.reshape(n, ncols), 2 * n, n, ncols), ncols, 2 * n)
    assert np.array_equal(root.numpy().view(np.uint64), want[-4:])
