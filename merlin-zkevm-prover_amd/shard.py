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


def lde_merkle_sharded(plan: ShardPlan, ops, dist, trace_shard, bufs):
    """Runs steps 1-4.  `ops` provides lde / copy_2d / merkle_build / merkle_levels / root helpers on the
    device the tensors live on; `dist` is torch.distributed (or None when world == 1).
    bufs: dict with 'ext' (n_ext*max_cols), 'recv' (rows_per_rank*ncols), 'nodes' ((2*rows_per_rank-1)*4),
    'roots' (world*4 + tree above them).  Returns the tensor holding the global root (4 u64)."""
    p = plan
    ext, nodes = bufs["ext"], bufs["nodes"]
    ops.lde(ext, trace_shard, p.n_ext, p.n, p.my_cols)                     # [n_ext x my_cols], pitch my_cols
    if p.world == 1:
        ops.merkle_build(nodes, ext, p.ncols, p.n_ext)
        return nodes[(2 * p.n_ext - 2) * 4:(2 * p.n_ext - 1) * 4]
    recv = bufs["recv"]
    send = ext[:p.n_ext * p.my_cols]
    dist.all_to_all_single(recv[:p.rows_per_rank * p.ncols], send, output_split_sizes=p.recv_splits,
                           input_split_sizes=p.send_splits)
    # recv holds G column slabs [rows_per_rank x cols_g]; repack into row-major rows (reuses `ext`)
    rows = ext
    off = 0
    for (c0, w) in p.cols:
        ops.copy_2d(rows, recv, p.rows_per_rank, w, dst_pitch=p.ncols, src_pitch=w, dst_off=c0, src_off=off)
        off += p.rows_per_rank * w
    ops.merkle_build(nodes, rows, p.ncols, p.rows_per_rank)
    my_root = nodes[(2 * p.rows_per_rank - 2) * 4:(2 * p.rows_per_rank - 1) * 4]
    roots = bufs["roots"]
    dist.all_gather_into_tensor(roots[:p.world * 4], my_root.contiguous())
    ops.merkle_levels(roots, p.world)                                      # top log2(G) levels, same on every rank
    return roots[(2 * p.world - 2) * 4:(2 * p.world - 1) * 4]
