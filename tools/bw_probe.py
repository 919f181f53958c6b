"""Practical HBM rates of this box with plain torch copies / fills (what a kernel that only moves bytes achieves): python tools/bw_probe.py
The last line has EAG's byte mix (read 96 MB, write 288 MB)."""
import torch, time
dev = "cuda"
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (96, 288, 1024):
    a = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    dt = t(lambda: b.copy_(a)); print(f"copy {mb} MB: {2 * mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s moved ({dt * 1e3:.4f} ms)")
    dt = t(lambda: b.fill_(1.0)); print(f"fill {mb} MB: {mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s")
    dt = t(lambda: a.sum()); print(f"read(sum) {mb} MB: {mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s")
# the EAG mix: read 96 MB, write 288 MB
a = torch.empty(24_000_000, dtype=torch.float32, device=dev).normal_()
o = torch.empty(24_000_000 * 3, dtype=torch.float32, device=dev)
def mix():
    o.view(3, -1)[0].copy_(a); o.view(3, -1)[1].copy_(a); o.view(3, -1)[2].copy_(a)
dt = t(mix); print(f"3 copies of 96 MB (read 288, write 288): {0.576 / dt / 1e3:.2f} TB/s, {dt * 1e3:.4f} ms")
x = a.view(-1, 1).expand(-1, 3)
dt = t(lambda: o.view(-1, 3).copy_(x)); print(f"broadcast 96 MB -> 288 MB (EAG's byte mix): {0.384 / dt / 1e3:.2f} TB/s, {dt * 1e3:.4f} ms")
