"""Multi-GPU sharding of the LDE + Merkleize path (BASELINE config 5).

One process per GPU.  The path has exactly one real exchange step:

  1. LDE is independent per column      -> rank g extends a contiguous COLUMN range of the trace;
  2. a Merkle leaf is a sponge over a whole row (linear_hash chains across all columns)
     -> leaves need ROWS, so the extended columns are redistributed all-to-all: rank g sends peer p the
     block rows[p] x cols[g]; with point-to-point xGMI every GPU talks to its 7 peers at once (batched
     isend/irecv, messages of at most 256 MiB).  Steps 1 and 2 are pipelined over up to four column chunks
     per rank: the exchange of chunk k (async, on the communicator's own stream) runs while chunk k+1 is
     being extended;
  3. rank p hashes its rows and builds the subtree over them (2^k rows -> the subtree root is one node
     of level log2(n_ext/G) of the global tree);
  4. the G subtree roots (4 u64 each) are all-gathered and every rank hashes the top log2(G) levels,
     giving the same root as a single-GPU build.

The compute steps are injected (`ops`), so the same orchestration runs on RCCL with the HIP library
(bench.py) and on gloo with CPU tensors in tests/test_shard_gloo.py.
"""
import os
from dataclasses import dataclass
from typing import List


def column_partition(ncols: int, world: int) -> List[tuple]:
    """Contiguous column ranges, sizes differing by at most one: [(col0, width)] per rank."""
    base, rem = divmod(ncols, world)
    out, c = [], 0
    for g in range(world):
        w = base + (1 if g < rem else 0)
        out.append((c, w))
        c += w
    return out


PIPE_TILE = 32      # columns per NTT tile: pipeline chunks are whole tiles
PIPE_DEPTH = 4      # at most this many LDE -> all-to-all pipeline stages per rank


def pipeline_chunks(width: int, max_width: int) -> List[int]:
    """Column chunks in which a rank extends and ships its columns, so that the all-to-all of chunk k runs while
    chunk k+1 is being extended.  The chunk width is derived from the WIDEST rank (every rank computes the same
    value), a multiple of the NTT tile width, giving at most PIPE_DEPTH chunks; the last chunk takes the remainder."""
    tiles = -(-max_width // PIPE_TILE)
    cw = PIPE_TILE * -(-tiles // PIPE_DEPTH)
    out = [cw] * (width // cw)
    if width % cw:
        out.append(width % cw)
    return out


@dataclass
class ShardPlan:
    n: int          # trace rows
    n_ext: int      # extended rows
    ncols: int      # total committed columns
    world: int
    rank: int

    def __post_init__(self):
        assert self.world >= 1 and (self.world & (self.world - 1)) == 0, "world size must be a power of two"
        assert self.n_ext % self.world == 0
        self.cols = column_partition(self.ncols, self.world)
        self.col0, self.my_cols = self.cols[self.rank]
        self.rows_per_rank = self.n_ext // self.world
        self.row0 = self.rank * self.rows_per_rank
        self.chunks = [pipeline_chunks(w, self.max_cols) for (_, w) in self.cols]   # per rank
        self.n_chunks = max(len(c) for c in self.chunks)                            # collectives per step (same on every rank)

    @property
    def max_cols(self):
        return max(w for (_, w) in self.cols)

    def chunk_width(self, rank: int, k: int) -> int:
        c = self.chunks[rank]
        return c[k] if k < len(c) else 0      # a rank with fewer chunks takes part with an empty block

    def chunk_col(self, rank: int, k: int) -> int:
        """First column (inside the rank's own range) of its chunk k."""
        return sum(self.chunks[rank][:k])

    # ---- layouts of the two exchange buffers (element offsets)
    def ext_chunk_base(self, k: int) -> int:
        """bufs['ext'] holds this rank's extended chunks back to back, chunk k as [n_ext x cw_k] with pitch cw_k."""
        return self.n_ext * self.chunk_col(self.rank, k)

    def send_block(self, k: int, peer: int):
        """(offset, count) in bufs['ext'] of what goes to `peer` from chunk k: its rows of my chunk."""
        cw = self.chunk_width(self.rank, k)
        return self.ext_chunk_base(k) + peer * self.rows_per_rank * cw, self.rows_per_rank * cw

    def recv_chunk_base(self, k: int) -> int:
        return self.rows_per_rank * sum(self.chunk_width(p, kk) for kk in range(k) for p in range(self.world))

    def recv_slab(self, k: int, peer: int):
        """(offset, count) in bufs['recv'] of the slab [rows_per_rank x cw] that arrives from `peer` for chunk k."""
        off = self.recv_chunk_base(k) + self.rows_per_rank * sum(self.chunk_width(p, k) for p in range(peer))
        return off, self.rows_per_rank * self.chunk_width(peer, k)


def phase_lde_chunk(plan: ShardPlan, ops, trace_shard, bufs, k: int):
    """Step 1 for pipeline chunk k: extend my columns [chunk_col, +cw) into bufs['ext'] (contiguous, pitch cw)."""
    cw = plan.chunk_width(plan.rank, k)
    if cw:
        ops.lde(bufs["ext"], trace_shard, plan.n_ext, plan.n, cw, out_pitch=cw, in_pitch=plan.my_cols,
                out_off=plan.ext_chunk_base(k), in_off=plan.chunk_col(plan.rank, k), chunk=k)


MAX_MSG_BYTES = int(os.environ.get("MI_SHARD_MAX_MSG_BYTES", 256 << 20))   # cap of one point-to-point message (see phase_exchange_chunk)


def exchange_messages(plan: ShardPlan, k: int):
    """The point-to-point messages of pipeline chunk k as (peer, send_off, send_cnt, recv_off, recv_cnt) element
    ranges of bufs['ext'] / bufs['recv'], grouped into rounds; every rank derives the same number of rounds.
    A peer's block [rows_per_rank x cw] is cut by rows into pieces of at most MAX_MSG_BYTES."""
    p = plan
    max_cw = max(p.chunk_width(peer, k) for peer in range(p.world))
    rows_per_msg = max(1, (MAX_MSG_BYTES // 8) // max(max_cw, 1))
    rounds = []
    for r0 in range(0, p.rows_per_rank, rows_per_msg):
        r1 = min(p.rows_per_rank, r0 + rows_per_msg)
        msgs = []
        cw_me = p.chunk_width(p.rank, k)
        for peer in range(p.world):
            cw_peer = p.chunk_width(peer, k)
            s_off = p.send_block(k, peer)[0] + r0 * cw_me
            r_off = p.recv_slab(k, peer)[0] + r0 * cw_peer
            msgs.append((peer, s_off, (r1 - r0) * cw_me, r_off, (r1 - r0) * cw_peer))
        rounds.append(msgs)
    return rounds


def phase_exchange_chunk(plan: ShardPlan, dist, bufs, k: int):
    """Step 2 for pipeline chunk k: columns -> rows.  Returns the async work handles; the transfers are ordered after
    the chunk's LDE on the current stream and run beside the next chunk's LDE.

    Point-to-point sends/receives in batches (one NCCL group per round: every GPU talks to all its peers at once over
    its xGMI links), not all_to_all_single, for two reasons measured on this stack (RCCL 2.26): a send-to-self of
    more than 1 GiB delivers only its first half, and message sizes should not depend on the world size.  So the
    block a rank keeps is a plain device copy, and every message is at most MAX_MSG_BYTES."""
    p = plan
    works = []
    for msgs in exchange_messages(p, k):
        ops = []
        for (peer, s_off, s_cnt, r_off, r_cnt) in msgs:
            if peer == p.rank:
                if s_cnt:
                    bufs["recv"][r_off:r_off + r_cnt].copy_(bufs["ext"][s_off:s_off + s_cnt])
                continue
            if s_cnt:
                ops.append(dist.P2POp(dist.isend, bufs["ext"][s_off:s_off + s_cnt], peer))
            if r_cnt:
                ops.append(dist.P2POp(dist.irecv, bufs["recv"][r_off:r_off + r_cnt], peer))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
    return works


def phase_merkle_local(plan: ShardPlan, ops, bufs):
    """Step 3: repack the slabs into row-major rows (reusing bufs['ext']) and build my subtree.  Returns my root."""
    p = plan
    rows, recv = bufs["ext"], bufs["recv"]
    for k in range(p.n_chunks):
        for peer, (c0, _) in enumerate(p.cols):
            cw = p.chunk_width(peer, k)
            if cw:
                ops.copy_2d(rows, recv, p.rows_per_rank, cw, dst_pitch=p.ncols, src_pitch=cw,
                            dst_off=c0 + p.chunk_col(peer, k), src_off=p.recv_slab(k, peer)[0])
    ops.merkle_build(bufs["nodes"], rows, p.ncols, p.rows_per_rank)
    return bufs["nodes"][(2 * p.rows_per_rank - 2) * 4:(2 * p.rows_per_rank - 1) * 4]


def phase_top(plan: ShardPlan, ops, bufs):
    """Step 4b: bufs['roots'][:G*4] holds the G subtree roots in rank order; hash the top log2(G) levels."""
    ops.merkle_levels(bufs["roots"], plan.world)
    return bufs["roots"][(2 * plan.world - 2) * 4:(2 * plan.world - 1) * 4]


def lde_merkle_sharded(plan: ShardPlan, ops, dist, trace_shard, bufs, always_exchange=False):
    """Runs steps 1-4.  always_exchange: take the pipelined exchange path even for world == 1 (a one-rank
    communicator; used to exercise the collective calls on a single GPU).  `ops` provides lde / copy_2d / merkle_build / merkle_levels on the device the tensors live
    on; `dist` is torch.distributed (or None when world == 1).
    bufs: dict with 'ext' (max(n_ext*max_cols, rows_per_rank*ncols)), 'recv' (rows_per_rank*ncols),
    'nodes' ((2*rows_per_rank-1)*4), 'roots' ((2*world-1)*4).  Returns the tensor holding the global root (4 u64)."""
    p = plan
    if p.world == 1 and not always_exchange:
        ops.lde(bufs["ext"], trace_shard, p.n_ext, p.n, p.my_cols)
        ops.merkle_build(bufs["nodes"], bufs["ext"], p.ncols, p.n_ext)
        return bufs["nodes"][(2 * p.n_ext - 2) * 4:(2 * p.n_ext - 1) * 4]
    works = []
    for k in range(p.n_chunks):       # LDE of chunk k+1 overlaps the exchange of chunk k
        phase_lde_chunk(p, ops, trace_shard, bufs, k)
        works.extend(phase_exchange_chunk(p, dist, bufs, k))
    for w in works:
        w.wait()
    my_root = phase_merkle_local(p, ops, bufs)
    dist.all_gather_into_tensor(bufs["roots"][:p.world * 4], my_root.contiguous())
    return phase_top(p, ops, bufs)
