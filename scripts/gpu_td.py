"""TD mode at the headline size on the GPU box: covariance assembly GB/s, blocked Cholesky TFLOP/s, L.Z TFLOP/s,
with a correctness check of one matrix against LAPACK.  (Exploration/measurement harness, not product code.)"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
from oracle import pta_oracle as po

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
R = int(sys.argv[3]) if len(sys.argv) > 3 else 512
nm = 30
s = dv.stream_ptr()
rng = np.random.default_rng(5)
t = np.sort(rng.uniform(53000, 58478, N)) * 86400.0
Tspan = t.max() - t.min()
f = np.arange(1, nm + 1) / Tspan
phi = po.red_noise_prior(np.repeat(f, 2), -14.0, 3.0, Tspan)
epoch_of, ne, first, _ = po.quantize(t / 86400.0, dt=0.1)
sig2 = np.full(N, (0.5e-6) ** 2)
ec2 = np.full(N, (2e-7) ** 2)
t_d, f_d, phi_d, sig_d, ep_d, ec_d = dv.f64(t), dv.f64(f), dv.f64(phi), dv.f64(sig2), dv.i32(epoch_of), dv.f64(ec2)
Ft = dv.empty((2 * nm, N))
_lib.call("pta_rn_basis", dv.ptr(t_d), N, 0.0, dv.ptr(f_d), None, nm, 0, dv.ptr(Ft), N, s)
C = dv.zeros((B, N, N))


def timed(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


def assemble():
    for b in range(B):
        _lib.call("pta_td_cov_assemble", dv.ptr(Ft), N, N, 2 * nm, dv.ptr(phi_d), dv.ptr(sig_d), dv.ptr(ep_d), dv.ptr(ec_d),
                  ctypes.c_void_p(C.data_ptr() + 8 * b * N * N), N, s)


assemble()
ta = timed(assemble)
tri_bytes = 8.0 * N * (N + 64) / 2 * B
info = dv.zeros((B,), dtype=torch.int32)
tp = timed(lambda: _lib.call("pta_potrf_batched", dv.ptr(C), N, B, dv.ptr(info), s))
assert int(info.abs().sum().item()) == 0, info
flops = N ** 3 / 3.0 * B
z = dv.empty((R, N))
_lib.call("pta_rng_fill_normal", 1, 0, R, (5 << 24), N // 2, 1, dv.ptr(z), None, N, s)
out = dv.empty((R, N))
tt = timed(lambda: _lib.call("pta_td_trmm", dv.ptr(C), N, N, dv.ptr(z), N, R, dv.ptr(out), N, 0, s), reps=3)
res = {"N": N, "B": B, "R": R, "cov_assemble_ms": round(ta * 1e3, 2), "cov_GBps_lower_triangle": round(tri_bytes / ta / 1e9, 1),
       "potrf_ms": round(tp * 1e3, 2), "potrf_TFLOPs": round(flops / tp / 1e12, 2),
       "trmm_ms": round(tt * 1e3, 3), "trmm_TFLOPs_useful(N^2 R)": round(N * N * R / tt / 1e12, 2),
       "trmm_TFLOPs_executed(2 N^2 R)": round(2.0 * N * N * R / tt / 1e12, 2)}
# correctness of matrix 0 against LAPACK
Cref = po.td_covariance(t, -14.0, 3.0, nm, sig2, epoch_of, np.sqrt(ec2[first]) if False else np.full(ne, 2e-7))
Lref = np.linalg.cholesky(Cref)
L = C[0].cpu().numpy()
res["potrf_max_rel_err"] = float(np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)))
zr = z.cpu().numpy()[:4]
res["trmm_rel_err"] = float(np.max(np.abs(out.cpu().numpy()[:4] - zr @ Lref.T)) / np.sqrt(np.mean((zr @ Lref.T) ** 2)))
print(json.dumps(res))
