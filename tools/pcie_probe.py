#!/usr/bin/env python3
"""PCIe / host-memory microbenchmark behind the host-path design: pageable vs pinned H2D / D2H, cost of
pinning (hipHostRegister via torch.cuda.cudart), first touch of a fresh result buffer, multi-threaded memcpy."""
import time, ctypes, threading
import numpy as np, torch

def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

n = 300_000_000          # int32 -> 1.2 GB
a = np.arange(n, dtype=np.int32)
d = torch.empty(n, dtype=torch.int32, device="cuda")
ta = torch.from_numpy(a)
print("pageable H2D  %.1f GB/s" % (a.nbytes / t(lambda: d.copy_(ta)) / 1e9))
t0 = time.perf_counter(); p = torch.empty(n, dtype=torch.int32, pin_memory=True); print("pin alloc 1.2 GB %.3f s" % (time.perf_counter() - t0))
print("memcpy to pinned (1 thread) %.1f GB/s" % (a.nbytes / t(lambda: p.numpy().__setitem__(slice(None), a)) / 1e9))
print("pinned H2D    %.1f GB/s" % (a.nbytes / t(lambda: d.copy_(p, non_blocking=True)) / 1e9))
print("pinned D2H    %.1f GB/s" % (a.nbytes / t(lambda: p.copy_(d, non_blocking=True)) / 1e9))
out = torch.empty(n, dtype=torch.int32)
print("pageable D2H (touched)  %.1f GB/s" % (a.nbytes / t(lambda: out.copy_(d)) / 1e9))
def fresh():
    o = torch.empty(n, dtype=torch.int32); o.copy_(d)
print("pageable D2H (fresh buffer) %.1f GB/s" % (a.nbytes / t(fresh) / 1e9))
rt = torch.cuda.cudart()
b = np.arange(n, dtype=np.int32)
t0 = time.perf_counter(); rc = rt.cudaHostRegister(b.ctypes.data, b.nbytes, 0); dt = time.perf_counter() - t0
print("hipHostRegister 1.2 GB: rc %s %.3f s" % (rc, dt))
tb = torch.from_numpy(b)
print("registered H2D %.1f GB/s" % (a.nbytes / t(lambda: d.copy_(tb, non_blocking=True)) / 1e9))
t0 = time.perf_counter(); rt.cudaHostUnregister(b.ctypes.data); print("unregister %.3f s" % (time.perf_counter() - t0))
def par_copy(dst, src, k):
    step = (len(src) + k - 1) // k
    th = [threading.Thread(target=lambda i=i: dst.__setitem__(slice(i * step, (i + 1) * step), src[i * step:(i + 1) * step])) for i in range(k)]
    [x.start() for x in th]; [x.join() for x in th]
for k in (2, 4, 8, 16):
    print("memcpy to pinned (%d threads) %.1f GB/s" % (k, a.nbytes / t(lambda: par_copy(p.numpy(), a, k)) / 1e9))
