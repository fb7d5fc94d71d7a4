"""fp64 MFMA GEMM throughput of the Cholesky trailing update shape (C = beta C - A A^T, lower triangle) against K, size, batch and
beta (beta = 0: no read of C in the epilogue)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
s = dv.stream_ptr()
def run(M, K, batch, lower, beta, ld=None):
    ld = ld or (M + K)
    A = torch.randn((batch, M, ld), dtype=torch.float64, device="cuda")
    call = lambda: _lib.call("pta_dgemm", 1, M, M, K, ctypes.c_double(-1.0), dv.ptr(A), ld, 1, dv.ptr(A), ld, ctypes.c_double(beta),
                             ctypes.c_void_p(A.data_ptr() + 8 * K), ld, lower, batch, M * ld, M * ld, M * ld, 1, s)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): call()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    fl = 2.0 * M * M * K * batch * (0.5 if lower else 1.0)
    return round(fl / t / 1e12, 2), round(t * 1e3, 3)
for (M, K, b) in [(3968, 1024, 34), (3968, 512, 34), (3968, 256, 34), (4096, 4096, 8)]:
    for lower in (1, 0):
        for beta in (1.0, 0.0):
            tf, ms = run(M, K, b, lower, beta)
            print(json.dumps({"M": M, "K": K, "batch": b, "lower": lower, "beta": beta, "TFLOPs_useful": tf, "ms": ms}), flush=True)
