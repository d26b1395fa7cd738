#!/usr/bin/env python3
"""What the host link gives on this box (for the host-buffer path of dspi_process): D2H / H2D into pinned and
pageable memory, first-touch cost of fresh pages, hipHostRegister cost, multi-threaded memcpy out of a pinned buffer."""
import ctypes, time, threading
import numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
N = 1 << 30
dev = torch.empty(N, dtype=torch.uint8, device="cuda"); dev.fill_(3)
pin = torch.empty(N, dtype=torch.uint8, pin_memory=True)
def t(f, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
print("D2H pinned      %.1f GB/s" % (N / t(lambda: pin.copy_(dev, non_blocking=True)) / 1e9))
print("H2D pinned      %.1f GB/s" % (N / t(lambda: dev.copy_(pin, non_blocking=True)) / 1e9))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
dev2 = torch.empty(N, dtype=torch.uint8, device="cuda"); pin2 = torch.empty(N, dtype=torch.uint8, pin_memory=True)
def both():
    with torch.cuda.stream(s0): pin.copy_(dev, non_blocking=True)
    with torch.cuda.stream(s1): dev2.copy_(pin2, non_blocking=True)
print("D2H + H2D at once (1 GiB each) %.1f GB/s each" % (N / t(both) / 1e9))
page = np.empty(N, dtype=np.uint8); page[:] = 1
print("D2H pageable (touched)   %.1f GB/s" % (N / t(lambda: hip.hipMemcpy(page.ctypes.data, dev.data_ptr(), N, 2)) / 1e9))
print("H2D pageable             %.1f GB/s" % (N / t(lambda: hip.hipMemcpy(dev.data_ptr(), page.ctypes.data, N, 1)) / 1e9))
def fresh():
    z = np.zeros(N, dtype=np.uint8); hip.hipMemcpy(z.ctypes.data, dev.data_ptr(), N, 2)
print("D2H pageable (np.zeros each time) %.1f GB/s" % (N / t(fresh) / 1e9))
t0 = time.perf_counter(); rc = hip.hipHostRegister(page.ctypes.data, N, 0); t1 = time.perf_counter()
print("hipHostRegister 1 GiB: rc %d, %.1f ms" % (rc, (t1 - t0) * 1e3))
if rc == 0:
    print("D2H into registered pageable %.1f GB/s" % (N / t(lambda: hip.hipMemcpy(page.ctypes.data, dev.data_ptr(), N, 2)) / 1e9))
    t0 = time.perf_counter(); hip.hipHostUnregister(page.ctypes.data); print("hipHostUnregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
src = pin.numpy(); dst = np.empty(N, dtype=np.uint8); dst[:] = 0
for nt in (1, 2, 4, 8, 16):
    def work(i): np.copyto(dst[i * (N // nt):(i + 1) * (N // nt)], src[i * (N // nt):(i + 1) * (N // nt)])
    def run():
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]; [x.start() for x in th]; [x.join() for x in th]
    print("memcpy pinned -> pageable, %2d threads: %.1f GB/s" % (nt, N / t(run) / 1e9))
