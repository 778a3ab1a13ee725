import torch, time, numpy as np
n = 1 << 29  # 512 MiB
a = torch.empty(n, dtype=torch.uint8)            # pageable
a.fill_(1)
p = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
print("pageable H2D  %.1f GB/s" % (n / t(lambda: d.copy_(a)) / 1e9))
print("pinned   H2D  %.1f GB/s" % (n / t(lambda: d.copy_(p, non_blocking=True)) / 1e9))
print("host memcpy pageable->pinned (1 thread) %.1f GB/s" % (n / t(lambda: p.copy_(a)) / 1e9))
t0 = time.perf_counter(); q = a.pin_memory(); print("pin_memory() of 512 MiB (alloc+copy): %.3f s" % (time.perf_counter() - t0))
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
buf = np.ones(n, dtype=np.uint8)
t0 = time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n), 0); t1 = time.perf_counter()
print("hipHostRegister 512 MiB rc=%d: %.3f s" % (rc, t1 - t0))
tt = torch.from_numpy(buf)
print("registered H2D %.1f GB/s" % (n / t(lambda: d.copy_(tt, non_blocking=True)) / 1e9))
t0 = time.perf_counter(); hip.hipHostUnregister(ctypes.c_void_p(buf.ctypes.data)); print("unregister %.3f s" % (time.perf_counter() - t0))
