"""Multi-GPU sharding of the LDE + Merkleize path (BASELINE config 5).

One process per GPU.  The path has exactly one real exchange step:

  1. LDE is independent per column      -> a rank extends COLUMNS of the trace;
  2. a Merkle leaf is a sponge over a whole row (linear_hash chains across all columns)
     -> leaves need ROWS, so the extended columns are redistributed all-to-all: a rank sends peer p the
     rows[p] of its columns; with point-to-point xGMI every GPU talks to its 7 peers at once (batched
     isend/irecv, messages of at most 256 MiB);
  3. rank p hashes its rows and builds the subtree over them (2^k rows -> the subtree root is one node
     of level log2(n_ext/G) of the global tree);
  4. the G subtree roots (4 u64 each) are all-gathered and every rank hashes the top log2(G) levels,
     giving the same root as a single-GPU build.

Steps 1-3 are one pipeline.  Columns are dealt to the ranks in TILES of at most 32 (the NTT tile width), in rounds:
in round k every rank extends one tile of w_k columns (rank p the p-th of the round), ships it, and the G tiles of the
round -- a contiguous piece of every row -- arrive at each rank as G column windows.  Every rank gets the same number
of columns up to rounding to 8 (665 over 8 ranks: rounds of 32, 32 and 24 columns).
Because the sponge absorbs a row's columns in order, round k's windows can be absorbed as soon as they are
there (mi_linear_hash_absorb_dev keeps the running capacity in the digest buffer): the exchange of round k
runs on the communicator's stream beside the LDE of round k+1 and the absorption of round k-1, and the windows
are hashed where they land -- nothing is repacked.  The slower of "a rank's arithmetic" and "a rank's link
traffic", not their sum, bounds a step.

The compute steps are injected (`ops`), so the same orchestration runs on RCCL with the HIP library
(bench.py) and on gloo with CPU tensors in tests/test_shard_gloo.py.
"""
import os
from dataclasses import dataclass

# cap of one point-to-point message (see phase_exchange)
MAX_MSG_BYTES = int(os.environ.get("MI_SHARD_MAX_MSG_BYTES", 256 << 20))


@dataclass
class ShardPlan:
    n: int          # trace rows
    n_ext: int      # extended rows
    ncols: int      # total committed columns
    world: int
    rank: int
    tile: int = 32  # widest tile (columns a rank extends per round); a multiple of 8: the sponge absorbs whole 8-column blocks

    def __post_init__(self):
        assert self.world >= 1 and (self.world & (self.world - 1)) == 0, "world size must be a power of two"
        assert self.n_ext % self.world == 0 and self.tile % 8 == 0
        # Balanced dealing.  Every rank gets (up to) `per_rank` columns, the smallest multiple of 8 that covers ncols / world;
        # they are dealt in rounds: in round k every rank extends ONE tile of round_w[k] columns, rank p the columns
        # [round_c0[k] + p * w, + w) -- so a round is a contiguous piece of every row, in rank order.  All rounds use the
        # full tile width except the last, which takes the remainder: 665 columns over 8 ranks = 88 per rank = rounds
        # of 32, 32 and 24 columns (ranks 0..5 end with 88 columns, rank 6 with 73, rank 7 with 64), where dealing fixed
        # 32-column tiles round-robin left ranks with 96 and 64.  Only the globally last non-empty tile may be narrower
        # than its round (it is in the last round, so zero padding stays at the end of the row).
        self.per_rank = -(-self.ncols // (8 * self.world)) * 8
        self.round_w = [self.tile] * (self.per_rank // self.tile) + ([self.per_rank % self.tile] if self.per_rank % self.tile else [])
        self.n_rounds = len(self.round_w)                       # same on every rank
        self.round_c0 = [self.world * sum(self.round_w[:k]) for k in range(self.n_rounds)]
        self.local_c0 = [sum(self.round_w[:k]) for k in range(self.n_rounds)]
        self.rows_per_rank = self.n_ext // self.world
        self.row0 = self.rank * self.rows_per_rank
        self.my_cols = sum(self.width(k, self.rank) for k in range(self.n_rounds))

    # ---- tiles
    def round_cols(self, k: int, rank: int):
        """(first global column, width) of the tile `rank` extends in round k; width 0: that rank sits the round out."""
        c0 = self.round_c0[k] + rank * self.round_w[k]
        return c0, max(0, min(self.round_w[k], self.ncols - c0))

    def width(self, k: int, rank: int) -> int:
        return self.round_cols(k, rank)[1]

    def my_tile_cols(self):
        """[(first global column, width)] of my tiles, in round order (empty tiles left out)."""
        return [self.round_cols(k, self.rank) for k in range(self.n_rounds) if self.width(k, self.rank)]

    @property
    def max_cols(self) -> int:
        """Columns of the rank with the most (rank 0)."""
        return sum(self.width(k, 0) for k in range(self.n_rounds))

    # ---- buffer layouts (element offsets)
    # trace shard : [n x my_cols], my round-k tile at local column local_c0[k] (only the globally last tile is narrower
    #               than its round, and it is the last tile of its owner)
    # bufs['ext'] : my extended tiles back to back, round k's as [n_ext x w] with pitch w, at n_ext * local_c0[k]
    # bufs['recv']: window (k, p) = peer p's tile of round k restricted to my rows, [rows_per_rank x w] with pitch w,
    #               at rows_per_rank * (first global column of that tile); my own slot stays unused (my rows of my
    #               tile are read in place from bufs['ext'])
    def local_col(self, k: int) -> int:
        return self.local_c0[k]

    def ext_base(self, k: int) -> int:
        return self.n_ext * self.local_c0[k]

    def send_block(self, k: int, peer: int):
        """(offset, count) in bufs['ext'] of peer's rows of my round-k tile."""
        w = self.width(k, self.rank)
        return self.ext_base(k) + peer * self.rows_per_rank * w, self.rows_per_rank * w

    def recv_window(self, k: int, peer: int):
        """(offset, count) in bufs['recv'] of the window that arrives from `peer` in round k."""
        c0, w = self.round_cols(k, peer)
        return self.rows_per_rank * c0, self.rows_per_rank * w

    def windows(self, k: int):
        """Round k's column windows of MY rows in column order: [(buffer name, offset, width, pitch)]."""
        out = []
        for p in range(self.world):
            w = self.width(k, p)
            if not w:
                continue
            if p == self.rank:
                out.append(("ext", self.ext_base(k) + self.row0 * w, w, w))
            else:
                out.append(("recv", self.recv_window(k, p)[0], w, w))
        return out

    def row_windows(self):
        """All column windows of MY rows, in column order (the row-sharded extended trace as the leaf sponge and the
        query openings read it)."""
        return [w for k in range(self.n_rounds) for w in self.windows(k)]

    def ext_elems(self) -> int:
        return self.n_ext * self.per_rank

    def recv_elems(self) -> int:
        return self.rows_per_rank * self.world * self.per_rank if self.world > 1 else 0   # no peers: nothing arrives


def exchange_messages(plan: ShardPlan, k: int):
    """The point-to-point messages of round k as message rounds of (peer, send_off, send_cnt, recv_off, recv_cnt)
    element ranges of bufs['ext'] / bufs['recv']; every rank derives the same number of message rounds.  A peer's
    block [rows_per_rank x w] is cut by rows into pieces of at most MAX_MSG_BYTES."""
    p = plan
    rows_per_msg = max(1, (MAX_MSG_BYTES // 8) // p.round_w[k])
    out = []
    w_me = p.width(k, p.rank)
    for r0 in range(0, p.rows_per_rank, rows_per_msg):
        r1 = min(p.rows_per_rank, r0 + rows_per_msg)
        msgs = []
        for peer in range(p.world):
            if peer == p.rank:
                continue
            w_peer = p.width(k, peer)
            msgs.append((peer, p.send_block(k, peer)[0] + r0 * w_me, (r1 - r0) * w_me,
                         p.recv_window(k, peer)[0] + r0 * w_peer, (r1 - r0) * w_peer))
        out.append(msgs)
    return out


def phase_lde(plan: ShardPlan, ops, trace_shard, bufs, k: int):
    """Round k, step 1: extend my k-th tile into bufs['ext'] (contiguous, pitch = its width)."""
    w = plan.width(k, plan.rank)
    if w:
        ops.lde(bufs["ext"], trace_shard, plan.n_ext, plan.n, w, out_pitch=w, in_pitch=plan.my_cols,
                out_off=plan.ext_base(k), in_off=plan.local_col(k), chunk=k)


def phase_exchange(plan: ShardPlan, dist, bufs, k: int):
    """Round k, step 2: every rank sends each peer that peer's rows of its tile.  Returns the async work handles; the
    transfers are ordered after the tile's LDE on the current stream.

    Point-to-point sends/receives in batches (one NCCL group per message round: every GPU talks to all its peers at
    once over its xGMI links), not all_to_all_single: measured on this stack (RCCL 2.26) a send-to-self of more than
    1 GiB delivers only its first half, and message sizes should not depend on the world size.  The rows a rank keeps
    are not moved at all, and every message is at most MAX_MSG_BYTES."""
    works = []
    for msgs in exchange_messages(plan, k):
        p2p = []
        for (peer, s_off, s_cnt, r_off, r_cnt) in msgs:
            if s_cnt:
                p2p.append(dist.P2POp(dist.isend, bufs["ext"][s_off:s_off + s_cnt], peer))
            if r_cnt:
                p2p.append(dist.P2POp(dist.irecv, bufs["recv"][r_off:r_off + r_cnt], peer))
        if p2p:
            works.extend(dist.batch_isend_irecv(p2p))
    return works


def phase_absorb(plan: ShardPlan, ops, bufs, k: int):
    """Round k, step 3: absorb the round's columns [tile * G * k, ...) of my rows into the running leaf digests
    (level 0 of my subtree in bufs['nodes'])."""
    wins = [(bufs[name], off, w, pitch) for (name, off, w, pitch) in plan.windows(k)]
    if wins:
        ops.absorb(bufs["nodes"], wins, plan.rows_per_rank, first=(k == 0), final=(k == plan.n_rounds - 1), chunk=k)


def phase_subtree(plan: ShardPlan, ops, bufs):
    """Levels above my leaves.  Returns my subtree root (a view)."""
    ops.merkle_levels(bufs["nodes"], plan.rows_per_rank)
    return bufs["nodes"][(2 * plan.rows_per_rank - 2) * 4:(2 * plan.rows_per_rank - 1) * 4]


def phase_top(plan: ShardPlan, ops, bufs):
    """Step 4b: bufs['roots'][:G*4] holds the G subtree roots in rank order; hash the top log2(G) levels."""
    ops.merkle_levels(bufs["roots"], plan.world)
    return bufs["roots"][(2 * plan.world - 2) * 4:(2 * plan.world - 1) * 4]


def _wait_all(ops, pending):
    """The compute stream waits for a round's transfers; ops may bracket the wait (wait_begin / wait_end: stream-side timers that
    show how long the kernels stood still for the links)."""
    k, works = pending
    if hasattr(ops, "wait_begin"):
        ops.wait_begin(k)
    for w in works:
        w.wait()
    if hasattr(ops, "wait_end"):
        ops.wait_end(k)


def lde_merkle_sharded(plan: ShardPlan, ops, dist, trace_shard, bufs, always_exchange=False):
    """Runs steps 1-4.  `ops` provides lde / absorb / merkle_build / merkle_levels on the device the tensors live on;
    `dist` is torch.distributed (or None when world == 1).  always_exchange: take the pipelined path even for
    world == 1 (then there are no peers and every window is read in place; used to rehearse the path on one GPU).
    bufs: 'ext' (plan.ext_elems(); world 1 without always_exchange: n_ext * ncols), 'recv' (plan.recv_elems()),
    'nodes' ((2*rows_per_rank-1)*4), 'roots' ((2*world-1)*4).  Returns the tensor holding the global root (4 u64)."""
    p = plan
    assert p.ncols > 4, "linear_hash copies rows of at most 4 elements instead of hashing them: nothing to shard"
    if p.world == 1 and not always_exchange:
        ops.lde(bufs["ext"], trace_shard, p.n_ext, p.n, p.ncols)
        ops.merkle_build(bufs["nodes"], bufs["ext"], p.ncols, p.n_ext)
        return bufs["nodes"][(2 * p.n_ext - 2) * 4:(2 * p.n_ext - 1) * 4]
    pending = None
    for k in range(p.n_rounds):
        phase_lde(p, ops, trace_shard, bufs, k)
        works = phase_exchange(p, dist, bufs, k) if p.world > 1 else []
        if pending is not None:          # round k-1 has arrived (its transfers ran beside this round's LDE)
            _wait_all(ops, pending)
            phase_absorb(p, ops, bufs, pending[0])
        pending = (k, works)
    _wait_all(ops, pending)
    phase_absorb(p, ops, bufs, pending[0])
    my_root = phase_subtree(p, ops, bufs)
    if p.world == 1:
        return my_root
    dist.all_gather_into_tensor(bufs["roots"][:p.world * 4], my_root.contiguous())
    return phase_top(p, ops, bufs)


# ------------------------------------------------------------------ diagnostics for a first run on real hardware
def exchange_bytes_to_peers(plan: ShardPlan):
    """Bytes this rank sends to every peer per step (its own entry is 0): what the exchange should show on each xGMI link."""
    out = [0] * plan.world
    for k in range(plan.n_rounds):
        for msgs in exchange_messages(plan, k):
            for (peer, _s_off, s_cnt, _r_off, _r_cnt) in msgs:
                out[peer] += 8 * s_cnt
    return out


def gather_sharded_result(plan: ShardPlan, ops, dist, bufs):
    """The row-sharded extension and leaf digests assembled on rank 0 (None elsewhere): ext [n_ext, ncols], digests [n_ext, 4].
    For verification against the oracle at sizes where one GPU / host holds the whole result; every rank calls it."""
    p = plan
    wins = [(bufs[name], off, w, pitch) for (name, off, w, pitch) in p.row_windows()]
    mine = ops.gather_rows(wins, list(range(p.rows_per_rank))).contiguous()            # [rows_per_rank, ncols]
    dig = bufs["nodes"][:4 * p.rows_per_rank].reshape(p.rows_per_rank, 4).contiguous()
    if p.world == 1:
        return mine, dig
    ext_all = [ops.zeros(p.rows_per_rank * p.ncols).view(p.rows_per_rank, p.ncols) for _ in range(p.world)] if p.rank == 0 else None
    dig_all = [ops.zeros(p.rows_per_rank * 4).view(p.rows_per_rank, 4) for _ in range(p.world)] if p.rank == 0 else None
    dist.gather(mine, ext_all, dst=0)
    dist.gather(dig, dig_all, dst=0)
    if p.rank != 0:
        return None, None
    import torch
    return torch.cat(ext_all, dim=0), torch.cat(dig_all, dim=0)


# ------------------------------------------------------------------ query openings over the row-sharded tree
def merkle_levels_of(n: int) -> int:
    return max(n - 1, 0).bit_length()


def group_proofs_sharded(plan: ShardPlan, ops, dist, bufs, idx):
    """MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35) for a batch of query rows `idx` (host list) of the sharded tree
    that lde_merkle_sharded built: proof = the row's ncols values, then the sibling of every level, leaves upward.

    The row lives on ONE rank (rows_per_rank consecutive rows each) as column windows (plan.row_windows()); that rank
    gathers the row's values out of its windows and the siblings of the lower log2(rows_per_rank) levels out of its
    subtree; the top log2(world) levels come from the G subtree roots every rank holds (bufs['roots']).  Every rank
    fills the proofs of the rows it owns into a zero tensor and one all-reduce (sum) of that small tensor leaves the
    complete set on every rank (nq x (ncols + 4 levels) words: 128 queries of the 665-column tree = 780 KB).
    ops: gather_rows(windows, rows) -> [len(rows), ncols] tensor; merkle_paths(nodes, height, idx) -> [len(idx), 4 levels];
    zeros(n) -> tensor on the ops' device."""
    p = plan
    lv_sub, lv_top = merkle_levels_of(p.rows_per_rank), merkle_levels_of(p.world)
    stride = p.ncols + 4 * (lv_sub + lv_top)
    nq = len(idx)
    out = ops.zeros(nq * stride).view(nq, stride)
    mine = [q for q in range(nq) if p.row0 <= int(idx[q]) < p.row0 + p.rows_per_rank]
    if mine:
        local = [int(idx[q]) - p.row0 for q in mine]
        wins = [(bufs[name], off, w, pitch) for (name, off, w, pitch) in p.row_windows()]
        vals = ops.gather_rows(wins, local)                                   # [len(mine), ncols]
        sub = ops.merkle_paths(bufs["nodes"], p.rows_per_rank, local)         # [len(mine), 4 * lv_sub]
        for j, q in enumerate(mine):
            out[q, :p.ncols] = vals[j]
            out[q, p.ncols:p.ncols + 4 * lv_sub] = sub[j]
            if lv_top:
                out[q, p.ncols + 4 * lv_sub:] = ops.merkle_paths(bufs["roots"], p.world, [p.rank])[0]
    if p.world > 1:
        dist.all_reduce(out)     # sum of one rank's values and zeros elsewhere: exact in the int64 container
    return out


# ------------------------------------------------------------------ FRI commit, fold sharded by output index
def fri_fold_sharded(plan_world: int, rank: int, ops, dist, nxt, pol, prev_bits: int, cur_bits: int, nbits_ext: int, x, min_per_rank: int = 1024):
    """One fold step (friProve.cpp:44-108).  Output g depends on pol[i * 2^cur + g], i < 2^(prev - cur): the outputs are
    independent, every rank holds the whole input, so rank r folds the outputs [r * cnt, (r + 1) * cnt) and the slices are
    all-gathered; small steps (fewer than min_per_rank outputs per rank) are folded whole by every rank --
    cheaper than a collective and the ranks stay in lockstep by construction."""
    n_out = 1 << cur_bits
    if plan_world == 1 or prev_bits == cur_bits or n_out // plan_world < min_per_rank:
        ops.fri_fold(nxt, pol, prev_bits, cur_bits, nbits_ext, x)
        return
    cnt = n_out // plan_world
    g0 = rank * cnt
    ops.fri_fold_range(nxt, pol, prev_bits, cur_bits, nbits_ext, x, g0, cnt)
    mine = nxt[3 * g0:3 * (g0 + cnt)].clone()                  # the gather writes the whole of nxt, my slice included
    dist.all_gather_into_tensor(nxt[:3 * n_out], mine)


def fri_commit_sharded(plan_world: int, rank: int, ops, dist, transcript, pol, steps_bits, nbits_ext: int, min_per_rank: int = 1024):
    """The fold / commit loop of FRIProve::prove (friProve.cpp:20-134) on every rank in lockstep: per step a challenge from
    the (replicated, deterministic) transcript, the fold -- sharded by output index while it is large -- then the step
    tree over the transposed polynomial (friProve.cpp:110-126), whose root goes into the transcript.  The trees after the
    first fold are tiny (2^19 leaves at zkEVM size) and are built by every rank.  Returns (final polynomial tensor,
    [(nodes, transposed source, groups, group size)] per step tree, [challenges])."""
    trees, challenges = [], []
    pol_bits = nbits_ext
    cur_pol = pol
    for si, cur in enumerate(steps_bits):
        x = transcript.get_field()
        challenges.append(x)
        nxt = ops.zeros(3 << cur)
        fri_fold_sharded(plan_world, rank, ops, dist, nxt, cur_pol, pol_bits, cur, nbits_ext, x, min_per_rank)
        if si < len(steps_bits) - 1:
            nb = steps_bits[si + 1]
            groups, gsz = 1 << nb, (1 << (cur - nb)) * 3
            src = ops.zeros(3 << cur)
            ops.fri_transpose(src, nxt, 1 << cur, nb)
            nodes = ops.zeros((2 * groups - 1) * 4)
            ops.merkle_build(nodes, src, gsz, groups)
            transcript.put(ops.to_host(nodes[(2 * groups - 2) * 4:(2 * groups - 1) * 4]))
            trees.append((nodes, src, groups, gsz))
        else:
            transcript.put(ops.to_host(nxt[:3 << cur]))
        cur_pol, pol_bits = nxt, cur
    return cur_pol, trees, challenges


# ------------------------------------------------------------------ the orchestration's compute steps on the HIP library
def device_ops(ctx):
    """`ops` for lde_merkle_sharded / group_proofs_sharded / fri_commit_sharded over an mi_stark.Context (torch int64
    tensors as u64 containers, work enqueued on torch's current stream)."""
    import numpy as np
    import torch

    class Ops:
        @staticmethod
        def lde(out, inp, ne, nn, c, out_pitch=None, in_pitch=None, out_off=0, in_off=0, chunk=0):
            ctx.lde(out, inp, ne, nn, c, out_pitch=out_pitch, in_pitch=in_pitch, out_off=out_off, in_off=in_off)

        @staticmethod
        def absorb(digests, windows, nrows, first, final, chunk=0):
            ctx.linear_hash_absorb(digests, windows, nrows, first, final)

        merkle_build = staticmethod(lambda nodes, src, c, rows: ctx.merkle_build(nodes, src, c, rows))
        merkle_levels = staticmethod(ctx.merkle_levels)
        zeros = staticmethod(ctx.zeros)
        to_host = staticmethod(ctx.to_host)
        fri_fold = staticmethod(ctx.fri_fold)
        fri_fold_range = staticmethod(ctx.fri_fold_range)
        fri_transpose = staticmethod(ctx.fri_transpose)

        @staticmethod
        def gather_rows(windows, rows):
            """rows of the row-sharded extended trace out of its column windows (device-side strided gathers: plumbing)"""
            r = torch.tensor(rows, dtype=torch.int64, device=ctx.device)
            parts = [torch.as_strided(t, (int(r.numel()) and (t.numel() - off - w) // pitch + 1, w), (pitch, 1), off)[r] for (t, off, w, pitch) in windows]
            return torch.cat(parts, dim=1)

        @staticmethod
        def merkle_paths(nodes, height, idx):
            lv = merkle_levels_of(height)
            out = ctx.zeros(max(len(idx) * 4 * lv, 1))
            if lv:
                ctx.merkle_paths(out, nodes, height, np.asarray(idx, dtype=np.uint64))
            return out[:len(idx) * 4 * lv].view(len(idx), 4 * lv)

    return Ops
