#!/usr/bin/env python3
"""What page-locking costs on the box: hipHostMalloc (torch pin_memory) and hipHostRegister of touched / untouched pages,
and the H2D rate of pageable, registered and allocated-pinned memory.  Decides how the CLI gets its page-locked buffers."""
import sys
import time

import numpy as np
import torch

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(GB * (1 << 30))
torch.cuda.init()
dev = torch.empty(n, dtype=torch.uint8, device="cuda:0")
rt = torch.cuda.cudart()


def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for rep in range(2):
    dt, pinned = t(lambda: torch.empty(n, dtype=torch.uint8, pin_memory=True))
    print(f"hipHostMalloc {GB:.1f} GB: {dt * 1e3:8.1f} ms  ({GB / dt:.2f} GB/s)")
    dt, _ = t(lambda: dev.copy_(pinned, non_blocking=True))
    print(f"  H2D from it:        {dt * 1e3:8.1f} ms  ({GB / dt:.1f} GB/s)")
    t0 = time.perf_counter()
    pinned.copy_(dev, non_blocking=True)
    t_call = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"  D2H to hipHostMalloc: call returns after {t_call * 1e3:8.2f} ms, done after {t_all * 1e3:8.1f} ms  ({GB / t_all:.1f} GB/s)")
    del pinned
    a = np.empty(n, np.uint8)
    dt, _ = t(lambda: a.fill(1))
    print(f"first touch (1 thread): {dt * 1e3:8.1f} ms  ({GB / dt:.2f} GB/s)")
    ta = torch.from_numpy(a)
    dt, _ = t(lambda: dev.copy_(ta))
    print(f"  H2D pageable:       {dt * 1e3:8.1f} ms  ({GB / dt:.1f} GB/s)")
    dt, rc = t(lambda: rt.cudaHostRegister(a.ctypes.data, n, 0))
    print(f"hipHostRegister (touched pages): {dt * 1e3:8.1f} ms  ({GB / dt:.2f} GB/s) rc={rc}")
    dt, _ = t(lambda: dev.copy_(ta, non_blocking=True))
    print(f"  H2D registered:     {dt * 1e3:8.1f} ms  ({GB / dt:.1f} GB/s)")
    t0 = time.perf_counter()
    ta.copy_(dev, non_blocking=True)
    t_call = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"  D2H to registered:  call returns after {t_call * 1e3:8.2f} ms, done after {t_all * 1e3:8.1f} ms  ({GB / t_all:.1f} GB/s)")
    dt, _ = t(lambda: rt.cudaHostUnregister(a.ctypes.data))
    print(f"hipHostUnregister:    {dt * 1e3:8.1f} ms")
    b = np.empty(n, np.uint8)
    dt, rc = t(lambda: rt.cudaHostRegister(b.ctypes.data, n, 0))
    print(f"hipHostRegister (untouched pages): {dt * 1e3:8.1f} ms  ({GB / dt:.2f} GB/s) rc={rc}")
    rt.cudaHostUnregister(b.ctypes.data)
    del a, b, ta
