import torch, time
x = torch.empty(1 << 31, dtype=torch.int64, device="cuda")  # 16 GiB
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): y.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("copy 16 GiB: %.2f ms, %.2f TB/s (read+write)" % (dt * 1e3, 2 * x.numel() * 8 / dt / 1e12))
