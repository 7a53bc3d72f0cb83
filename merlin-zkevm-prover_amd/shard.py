"""Multi-GPU sharding of the LDE + Merkleize path (BASELINE config 5).

One process per GPU.  The path has exactly one real exchange step:

  1. LDE is independent per column      -> rank g extends a contiguous COLUMN range of the trace;
  2. a Merkle leaf is a sponge over a whole row (linear_hash chains across all columns)
     -> leaves need ROWS, so the extended columns are redistributed with one all-to-all: rank g sends
     peer p the block rows[p] x cols[g]; with point-to-point xGMI every GPU talks to its 7 peers at once;
  3. rank p hashes its rows and builds the subtree over them (2^k rows -> the subtree root is one node
     of level log2(n_ext/G) of the global tree);
  4. the G subtree roots (4 u64 each) are all-gathered and every rank hashes the top log2(G) levels,
     giving the same root as a single-GPU build.

The compute steps are injected (`ops`), so the same orchestration runs on RCCL with the HIP library
(bench.py) and on gloo with CPU tensors in tests/test_shard_gloo.py.
"""
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

    @property
    def send_splits(self):   # elements sent to each peer: its row block of my columns
        return [self.rows_per_rank * self.my_cols] * self.world

    @property
    def recv_splits(self):   # elements received from each peer: my row block of its columns
        return [self.rows_per_rank * w for (_, w) in self.cols]

    @property
    def max_cols(self):
        return max(w for (_, w) in self.cols)


def phase_lde(plan: ShardPlan, ops, trace_shard, bufs):
    """Step 1: extend my column range.  bufs['ext'] receives [n_ext x my_cols] (pitch my_cols)."""
    ops.lde(bufs["ext"], trace_shard, plan.n_ext, plan.n, plan.my_cols)


def phase_exchange(plan: ShardPlan, dist, bufs):
    """Step 2: columns -> rows.  After it bufs['recv'] holds G slabs [rows_per_rank x cols_g], in rank order."""
    p = plan
    dist.all_to_all_single(bufs["recv"][:p.rows_per_rank * p.ncols], bufs["ext"][:p.n_ext * p.my_cols],
                           output_split_sizes=p.recv_splits, input_split_sizes=p.send_splits)


def phase_merkle_local(plan: ShardPlan, ops, bufs):
    """Step 3: repack the slabs into row-major rows (reusing bufs['ext']) and build my subtree.  Returns my root."""
    p = plan
    rows, recv, off = bufs["ext"], bufs["recv"], 0
    for (c0, w) in p.cols:
        ops.copy_2d(rows, recv, p.rows_per_rank, w, dst_pitch=p.ncols, src_pitch=w, dst_off=c0, src_off=off)
        off += p.rows_per_rank * w
    ops.merkle_build(bufs["nodes"], rows, p.ncols, p.rows_per_rank)
    return bufs["nodes"][(2 * p.rows_per_rank - 2) * 4:(2 * p.rows_per_rank - 1) * 4]


def phase_top(plan: ShardPlan, ops, bufs):
    """Step 4b: bufs['roots'][:G*4] holds the G subtree roots in rank order; hash the top log2(G) levels."""
    ops.merkle_levels(bufs["roots"], plan.world)
    return bufs["roots"][(2 * plan.world - 2) * 4:(2 * plan.world - 1) * 4]


def lde_merkle_sharded(plan: ShardPlan, ops, dist, trace_shard, bufs):
    """Runs steps 1-4.  `ops` provides lde / copy_2d / merkle_build / merkle_levels on the device the tensors live
    on; `dist` is torch.distributed (or None when world == 1).
    bufs: dict with 'ext' (max(n_ext*max_cols, rows_per_rank*ncols)), 'recv' (rows_per_rank*ncols),
    'nodes' ((2*rows_per_rank-1)*4), 'roots' ((2*world-1)*4).  Returns the tensor holding the global root (4 u64)."""
    p = plan
    phase_lde(p, ops, trace_shard, bufs)
    if p.world == 1:
        ops.merkle_build(bufs["nodes"], bufs["ext"], p.ncols, p.n_ext)
        return bufs["nodes"][(2 * p.n_ext - 2) * 4:(2 * p.n_ext - 1) * 4]
    phase_exchange(p, dist, bufs)
    my_root = phase_merkle_local(p, ops, bufs)
    dist.all_gather_into_tensor(bufs["roots"][:p.world * 4], my_root.contiguous())
    return phase_top(p, ops, bufs)
