#!/usr/bin/env python3
"""What page-locking a client's array in place costs (hipHostRegister / hipHostUnregister), per size: next to nothing
on this platform (0.6 ms for 224 MB the client has touched) -- which is why spx.vec.register locks a view's buffer at
its first product."""
import time
import numpy as np
import torch

torch.cuda.init()
rt = torch.cuda.cudart()
for mb in (32, 224, 1024):
    a = np.random.rand(mb * (1 << 20) // 8)
    t = []
    for rep in range(3):
        t0 = time.perf_counter()
        assert int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) == 0
        t1 = time.perf_counter()
        rt.cudaHostUnregister(a.ctypes.data)
        t2 = time.perf_counter()
        t.append((t1 - t0, t2 - t1))
    print("%5d MB: register %.1f / %.1f / %.1f ms, unregister %.1f / %.1f / %.1f ms" %
          (mb, *(1e3 * x[0] for x in t), *(1e3 * x[1] for x in t)))
