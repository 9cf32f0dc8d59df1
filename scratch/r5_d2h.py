"""device -> pinned host bandwidth: one stream against the same bytes split over several streams"""
import time, torch
n = 1 << 29  # 2 GiB of float32
src = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
dst = torch.empty(n, dtype=torch.float32, pin_memory=True)
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        step = n // ns
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                dst[k * step:(k + 1) * step].copy_(src[k * step:(k + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{ns} stream(s): {n * 4 / best / 1e9:.1f} GB/s", flush=True)
