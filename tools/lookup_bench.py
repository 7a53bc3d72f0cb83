"""Times mi_calculate_h1h2_dev and mi_calculate_z_dev at 2^23 rows (the zkEVM's N) on random and run-heavy inputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import torch, mi_stark
ctx = mi_stark.Context(0)
n = 1 << 23
g = torch.Generator(device=ctx.device); g.manual_seed(1)
for dim in (1, 3):
    for kind in ("random", "heavy"):
        cols = 4 * dim
        area = torch.randint(0, 1 << 62, (n * cols,), generator=g, device=ctx.device, dtype=torch.int64)
        a = area.reshape(n, cols)
        if kind == "heavy":
            a[:, 0:dim] = a[(torch.arange(n, device=ctx.device) >> 10) << 10][:, 0:dim]   # runs of 1024 equal rows in t
        idx = torch.randint(0, n, (n,), generator=g, device=ctx.device)
        if kind == "heavy":
            idx[: n // 2] = 777
        a[:, dim:2 * dim] = a[idx][:, 0:dim]
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.calculate_h1h2(area[2 * dim:], cols, area[3 * dim:], cols, area[dim:], cols, area, cols, dim, n)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        print("h1h2 dim", dim, kind, "%.2f ms" % ((t1 - t0) * 1e3), flush=True)
z = torch.randint(0, 1 << 62, (n * 9,), generator=g, device=ctx.device, dtype=torch.int64)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.calculate_z(z[6:], 9, z, 9, z[3:], 9, n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("z %.2f ms" % ((t1 - t0) * 1e3))
