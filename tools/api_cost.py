"""Host-side cost of the runtime calls the step loop makes (per call, microseconds)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import ccn_loader; ccn_loader.load()
from cuda_convnet_amd import capi
from kutil import dev, dev_empty
L = capi.lib()
s1, s2, e1, e2 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
L.call('mcn_stream_create', C.byref(s1)); L.call('mcn_stream_create', C.byref(s2))
L.call('mcn_event_create_sync', C.byref(e1)); L.call('mcn_event_create_sync', C.byref(e2))
a = dev(np.zeros(1024, np.float32)); b = dev_empty((1024,))
scale = L.dll.mcn_scale; rec = L.dll.mcn_event_record; wait = L.dll.mcn_stream_wait_event
def bench(name, fn, n=2000):
    L.call('mcn_device_sync'); t = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t; L.call('mcn_device_sync'); tot = time.perf_counter() - t
    print('%-44s host %6.2f us/iter   incl. drain %6.2f us/iter' % (name, dt / n * 1e6, tot / n * 1e6), flush=True)
f2 = C.c_float(2.0)
bench('kernel launch (1 stream)', lambda: scale(a.ptr, b.ptr, 4, f2, s1))
bench('launch + record', lambda: (scale(a.ptr, b.ptr, 4, f2, s1), rec(e1, s1)))
bench('launch, record, wait(other), launch(other)', lambda: (scale(a.ptr, b.ptr, 4, f2, s1), rec(e1, s1), wait(s2, e1), scale(a.ptr, b.ptr, 4, f2, s2)))
bench('fork + join per iter', lambda: (scale(a.ptr, b.ptr, 4, f2, s1), rec(e1, s1), wait(s2, e1), scale(a.ptr, b.ptr, 4, f2, s2), rec(e2, s2), wait(s1, e2)))
bench('2 launches, 2 streams, no events', lambda: (scale(a.ptr, b.ptr, 4, f2, s1), scale(a.ptr, b.ptr, 4, f2, s2)))
# two host threads, one stream each (ctypes releases the GIL inside the call)
import threading
def worker(s, n):
    for _ in range(n): scale(a.ptr, b.ptr, 4, f2, s)
for nthreads in (1, 2):
    L.call('mcn_device_sync'); t = time.perf_counter()
    ths = [threading.Thread(target=worker, args=((s1, s2)[i], 20000)) for i in range(nthreads)]
    [x.start() for x in ths]; [x.join() for x in ths]
    dt = time.perf_counter() - t; L.call('mcn_device_sync'); tot = time.perf_counter() - t
    print('%d thread(s) x 20000 launches, own stream: host %6.2f us per launch per thread, drained %6.2f' % (nthreads, dt / 20000 * 1e6, tot / 20000 * 1e6), flush=True)
