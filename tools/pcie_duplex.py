#!/usr/bin/env python3
"""PCIe copy rates on the box: H2D alone, D2H alone, both at once on two streams (is the link used full duplex?),
and the same with the transfers cut into chunks.  Page-locked host memory throughout."""
import time
import torch

GB = 1 << 30
n = 2 * GB
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_in = torch.empty(n, dtype=torch.uint8, device="cuda:0")
d_out = torch.ones(n, dtype=torch.uint8, device="cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


def h2d(chunks=1):
    with torch.cuda.stream(s1):
        for c in range(chunks):
            a, b = n * c // chunks, n * (c + 1) // chunks
            d_in[a:b].copy_(h_in[a:b], non_blocking=True)


def d2h(chunks=1):
    with torch.cuda.stream(s2):
        for c in range(chunks):
            a, b = n * c // chunks, n * (c + 1) // chunks
            h_out[a:b].copy_(d_out[a:b], non_blocking=True)


for chunks in (1, 8, 64):
    t1 = timed(lambda: h2d(chunks))
    t2 = timed(lambda: d2h(chunks))
    t3 = timed(lambda: (h2d(chunks), d2h(chunks)))
    print(f"chunks {chunks:3d}: H2D {n / t1 / 1e9:6.1f} GB/s  D2H {n / t2 / 1e9:6.1f} GB/s  both at once {2 * n / t3 / 1e9:6.1f} GB/s total "
          f"({t3 * 1e3:.1f} ms vs {t1 * 1e3:.1f} + {t2 * 1e3:.1f})")
