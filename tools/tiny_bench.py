import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import kbench as kb
from kutil import call, dev, dev_empty
ws = kb.capi.workspace(64 << 20)
a = dev(np.zeros(1024, np.float32)); b = dev_empty((1024,))
print('scale 4 elems      %.2f us' % kb.timeit(lambda: call('mcn_scale', a, b, 4, 2.0, None), 200))
for (M, N, K, ta, tb) in [(16, 64, 16, 0, 0), (16, 64, 64, 0, 0), (160, 128, 64, 0, 0), (64, 64, 16, 1, 0), (64, 64, 640, 1, 0), (1024, 128, 160, 1, 0), (16, 16, 16, 0, 1), (160, 1024, 128, 0, 1)]:
    A = dev(np.zeros((max(M, K) * max(M, K),), np.float32)[:M * K].copy()) if True else None
    B = dev(np.zeros(K * N, np.float32)); Cc = dev_empty((M * N,))
    print('gemm M=%d N=%d K=%d ta=%d tb=%d   %.2f us' % (M, N, K, ta, tb, kb.timeit(lambda: call('mcn_gemm', A, B, Cc, M, N, K, ta, tb, 1.0, 0.0, ws, ws.nbytes, None), 200)))
