"""fp64 MFMA GEMM throughput of the Cholesky trailing update shape (C -= A A^T, lower triangle) against K, size and batch."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
s = dv.stream_ptr()
res = []
def run(M, K, batch, lower, algo, ld=None):
    ld = ld or (M + K)
    A = torch.randn((batch, M, ld), dtype=torch.float64, device="cuda")
    call = lambda: _lib.call("pta_dgemm", 1, M, M, K, ctypes.c_double(-1.0), dv.ptr(A), ld, 1, dv.ptr(A), ld, ctypes.c_double(1.0),
                             ctypes.c_void_p(A.data_ptr() + 8 * K), ld, lower, batch, M * ld, M * ld, M * ld, algo, s)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): call()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    fl = 2.0 * M * M * K * batch * (0.5 if lower else 1.0)
    return round(fl / t / 1e12, 2), round(t * 1e3, 3)
for (M, K, b) in [(4096, 256, 32), (4096, 512, 32), (4096, 1024, 32), (4096, 4096, 8), (2048, 256, 68), (1024, 256, 68), (4096, 64, 32), (4096, 128, 32)]:
    for lower in (1, 0):
        for algo in (1, 2):
            tf, ms = run(M, K, b, lower, algo)
            res.append({"M": M, "K": K, "batch": b, "lower": lower, "algo": algo, "TFLOPs_useful": tf, "ms": ms})
            print(json.dumps(res[-1]), flush=True)
