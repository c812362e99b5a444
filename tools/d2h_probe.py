#!/usr/bin/env python3
"""Why do the pipeline's downloads run at a third of the link?  D2H of one chunk's coverage bytes (52 MB) and of the whole
array, into page-locked memory at aligned / odd destination offsets, from aligned / odd sources, on streams of both priorities."""
import time
import torch

n_all = 568_000_000
h = torch.empty(n_all + 4096, dtype=torch.uint8, pin_memory=True)
d = torch.ones(n_all + 4096, dtype=torch.uint8, device="cuda:0")
lo_p, hi_p = -1, 0
streams = {"default-priority stream": torch.cuda.Stream(), "low-priority stream": torch.cuda.Stream(priority=0), "high-priority stream": torch.cuda.Stream(priority=-1)}


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


for name, st in streams.items():
    for n in (52_000_000, n_all):
        for dst_off, src_off in ((0, 0), (1, 0), (2, 0), (4, 0), (64, 0), (0, 1), (1, 1), (3, 7)):
            def go():
                with torch.cuda.stream(st):
                    h[dst_off:dst_off + n].copy_(d[src_off:src_off + n], non_blocking=True)
            t = timed(go)
            print(f"{name:24s} {n / 1e6:6.0f} MB  dst+{dst_off:<3d} src+{src_off:<3d}: {t * 1e3:7.2f} ms  {n / t / 1e9:6.1f} GB/s")
# H2D for comparison, odd offsets
for dst_off, src_off in ((0, 0), (1, 0), (0, 1), (4, 4)):
    n = 52_000_000
    t = timed(lambda: d[dst_off:dst_off + n].copy_(h[src_off:src_off + n], non_blocking=True))
    print(f"H2D {n / 1e6:6.0f} MB  dst+{dst_off:<3d} src+{src_off:<3d}: {t * 1e3:7.2f} ms  {n / t / 1e9:6.1f} GB/s")
