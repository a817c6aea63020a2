#!/usr/bin/env python3
"""Does hipMemcpyAsync(pinned host <- device) return before the copy is done?  Time of the call itself against the stream synchronisation behind it,
200 MB, both directions, on an idle device and behind a queued kernel-free event wait.  python3 tools/d2h_async_probe.py  (GPU box)"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def chk(e):
    if e: raise RuntimeError("hip error %d" % e)
n = 200 << 20
dev, host, st = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
chk(hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(n)))
chk(hip.hipHostMalloc(ctypes.byref(host), ctypes.c_size_t(n), ctypes.c_uint(0)))
chk(hip.hipStreamCreateWithFlags(ctypes.byref(st), ctypes.c_uint(1)))
chk(hip.hipMemset(dev, 1, ctypes.c_size_t(n)))
chk(hip.hipDeviceSynchronize())
for name, dst, src, kind in (("D2H", host, dev, 2), ("H2D", dev, host, 1), ("D2H", host, dev, 2), ("H2D", dev, host, 1)):
    t0 = time.perf_counter()
    chk(hip.hipMemcpyAsync(dst, src, ctypes.c_size_t(n), ctypes.c_int(kind), st))
    t1 = time.perf_counter()
    chk(hip.hipStreamSynchronize(st))
    t2 = time.perf_counter()
    print("%s 200 MB: the call %.3f ms, the synchronisation behind it %.3f ms (%.1f GB/s overall)" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t0) / 1e9))
