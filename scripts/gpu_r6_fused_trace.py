"""prepare_td of the 68 x 5000 array, ONE chain, last in the process: argv[1] = fused | twostep (rocprofv3 kernel traces of the two paths)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv
eng, psrs, noise = build_engine(68, 5000, seed=20260921)
eng.prepare_td()
P, n, ld = eng.P, eng.td_nst[0], eng.td_ld[0]
flags = _lib.POTRF_LEFT | _lib.POTRF_CHAINS(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
info = dv.zeros((P,), dtype=torch.int32)
need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, flags))
work = dv.empty((need,))
op = eng._td_fuse_ops
ec2 = (eng.d_ecorr_toa ** 2).contiguous()
for _ in range(2):
    torch.cuda.synchronize()
    if sys.argv[1] == "fused":
        _lib.call("pta_td_assemble_potrf", dv.ptr(op[1]), dv.ptr(op[2]), eng.plan.rn_k, dv.ptr(eng._td_sigma2), dv.ptr(eng.d_epoch_of), dv.ptr(ec2),
                  dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), flags, dv.ptr(work), need, dv.stream_ptr())
    else:
        eng.td_assemble()
        _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), flags, dv.ptr(work), need, dv.stream_ptr())
    torch.cuda.synchronize()
assert int(info.abs().sum().item()) == 0
