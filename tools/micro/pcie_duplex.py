#!/usr/bin/env python3
"""How much of the host link both directions get at the same time: 224 MB up, 224 MB down, one after the other and
together on two streams (pinned host memory), in pieces of 16 MB as the host-vector entry point sends them."""
import time
import torch

n = 28_000_000
hx = torch.empty(n, dtype=torch.float64).pin_memory()
hy = torch.empty(n, dtype=torch.float64).pin_memory()
dx = torch.empty(n, dtype=torch.float64, device="cuda")
dy = torch.empty(n, dtype=torch.float64, device="cuda")
up, down = torch.cuda.Stream(), torch.cuda.Stream()
P = 2 * 1024 * 1024


def run(do_up, do_down):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        for off in range(0, n, P):
            if do_up:
                with torch.cuda.stream(up):
                    dx[off:off + P].copy_(hx[off:off + P], non_blocking=True)
            if do_down:
                with torch.cuda.stream(down):
                    hy[off:off + P].copy_(dy[off:off + P], non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3


for name, a, b in (("up only", True, False), ("down only", False, True), ("both at once", True, True)):
    run(a, b)
    print("%-14s %.2f ms per 224 MB each way" % (name, run(a, b)))

# the same with a client's malloc'ed arrays page-locked where they lie (hipHostRegister: what spx.vec.register does)
import numpy as np
rt = torch.cuda.cudart()
ax, ay = np.random.rand(n), np.zeros(n)
for arr in (ax, ay):
    assert int(rt.cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)) == 0
hx, hy = torch.from_numpy(ax), torch.from_numpy(ay)
for name, a, b in (("registered: up only", True, False), ("registered: down only", False, True), ("registered: both at once", True, True)):
    run(a, b)
    print("%-26s %.2f ms per 224 MB each way" % (name, run(a, b)))
for arr in (ax, ay):
    rt.cudaHostUnregister(arr.ctypes.data)
