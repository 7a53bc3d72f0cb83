import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, mi_stark, glo
from bench_genproof import Transcript
ctx = mi_stark.Context(0)
nbits_ext = 24; NE = 1 << nbits_ext; N = NE >> 1
steps = [24, 19, 14, 10, 6]
f = ctx.empty(NE * 3); ctx.fill_synthetic(f, NE * 3, 5)
nxt, spare = ctx.zeros(NE * 3), ctx.zeros(NE * 3)
src = [ctx.zeros((1 << c) * 3) for c in steps[:-1]]
nodes = [ctx.zeros((2 * (1 << nb) - 1) * 4) for nb in steps[1:]]
lk = torch.randint(0, 1 << 62, (N * 12,), device=ctx.device, dtype=torch.int64)
lkv = lk.view(N, 12); lkv[:, 3:6] = lkv[torch.randint(0, N, (N,), device=ctx.device)][:, 0:3]
zq = torch.randint(0, 1 << 62, (N * 9,), device=ctx.device, dtype=torch.int64)
def fri(detail=False):
    tr = Transcript(ctx); tr.put(np.arange(8, dtype=np.uint64))
    pol, nx, pb = f, nxt, nbits_ext
    marks = []
    def mark(name):
        torch.cuda.synchronize(); marks.append((name, time.perf_counter()))
    mark("start")
    for si, cur in enumerate(steps):
        x = tr.get_field(); mark("chal")
        ctx.fri_fold(nx, pol, pb, cur, nbits_ext, x); mark("fold%d" % si)
        if si < len(steps) - 1:
            nb = steps[si + 1]; groups, gsz = 1 << nb, (1 << (cur - nb)) * 3
            ctx.fri_transpose(src[si], nx, 1 << cur, nb); mark("transpose")
            ctx.merkle_build(nodes[si], src[si], gsz, groups); mark("tree%d" % si)
            tr.put(ctx.to_host(nodes[si][-4:])); mark("put")
        else:
            tr.put(ctx.to_host(nx[:(1 << cur) * 3])); mark("putfinal")
        pol, nx = nx, (pol if pol is not f else spare); pb = cur
    t = [(n, (b - a) * 1e3) for (n, b), (_, a) in zip(marks[1:], marks[:-1])]
    return sum(v for _, v in t), t
for rep in range(3): tot, t = fri()
print("before: %.2f ms" % tot, [(n, round(v, 2)) for n, v in t])
ctx.calculate_h1h2(lk[6:], 12, lk[9:], 12, lk[3:], 12, lk, 12, 3, N)
for rep in range(2): tot, t = fri()
print("after h1h2: %.2f ms" % tot, [(n, round(v, 2)) for n, v in t])
ctx.calculate_z(zq[6:], 9, zq, 9, zq[3:], 9, N)
for rep in range(2): tot, t = fri()
print("after z: %.2f ms" % tot, [(n, round(v, 2)) for n, v in t])
