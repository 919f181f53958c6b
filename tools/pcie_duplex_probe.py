"""What the host link gives a download while an upload runs next to it (the cadence of the banded host pipelines, DESIGN.md 7.R5 b'): python tools/pcie_duplex_probe.py
Page-locked host tensors, two HIP streams, 24 MP sizes: 288 MB down in 16 chunks; alone, with the 96 MB mosaic going up in 16 chunks next to it (the pipeline's mix),
and with the upload direction saturated as well."""
import time
import torch

dev = "cuda"
MB = 1 << 20
down_dev = torch.empty(288 * MB // 4, dtype=torch.float32, device=dev).normal_()
down_host = torch.empty(288 * MB // 4, dtype=torch.float32).pin_memory()
up_host = torch.empty(288 * MB // 4, dtype=torch.float32).pin_memory().normal_()
up_dev = torch.empty(288 * MB // 4, dtype=torch.float32, device=dev)
s_down, s_up = torch.cuda.Stream(), torch.cuda.Stream()
NCH = 16


def run(up_mb: float, n: int = 10) -> float:
    """n rounds of 16 download chunks on one stream; next to every chunk `up_mb`/16 MB go up on the other stream.  Returns ms per round (wall)."""
    dch = down_dev.numel() // NCH
    uch = int(up_mb * MB // 4) // NCH
    ts = []
    for it in range(n + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(NCH):
            with torch.cuda.stream(s_down):
                down_host[k * dch:(k + 1) * dch].copy_(down_dev[k * dch:(k + 1) * dch], non_blocking=True)
            if uch:
                with torch.cuda.stream(s_up):
                    up_dev[k * uch:(k + 1) * uch].copy_(up_host[k * uch:(k + 1) * uch], non_blocking=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for up_mb, what in ((0, "download alone"), (48, "+ 48 MB up (uint16 mosaic)"), (96, "+ 96 MB up (float32 mosaic: the pipeline's mix)"), (288, "+ 288 MB up (both directions saturated)")):
    ms = run(up_mb)
    print("288 MB down in 16 chunks, %-50s %.2f ms = %.1f GB/s down%s" % (what + ":", ms, 288 * MB / 1e9 / (ms * 1e-3), (", %.1f GB/s up" % (up_mb * MB / 1e9 / (ms * 1e-3))) if up_mb else ""), flush=True)
t = []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); down_host.copy_(down_dev, non_blocking=True); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
print("288 MB down in one copy: %.2f ms = %.1f GB/s" % (sorted(t)[4], 288 * MB / 1e9 / (sorted(t)[4] * 1e-3)))
