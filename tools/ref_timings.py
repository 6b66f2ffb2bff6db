import time, torch
def timeit(f, n=500):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
a = torch.zeros(1, device="cuda")
print("tiny kernel (add_ on 1 elem): %.2f us" % timeit(lambda: a.add_(1)))
for mb in [38, 276]:
    n = mb * 1000 * 1000 // 8
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x[: n // 2])
    print("%d MB: sum %.2f us; copy half->half (same total traffic) %.2f us; mul_ in place (2x traffic) %.2f us" % (
        mb, timeit(lambda: x.sum()), timeit(lambda: y.copy_(x[: n // 2])), timeit(lambda: x.mul_(1.0))))
