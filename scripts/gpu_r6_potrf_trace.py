"""one assembly + ONE factorisation of the P x N^2 TD covariances with the given pta_potrf_batched_ws flags, last in the process (for rocprofv3
kernel traces: scripts/potrf_schedule_classes.py takes the dispatches behind the LAST assembly launch).  usage: P N flags(hex)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import configure_engine, headline_array
from pta_replicator_amd import _lib, device as dv
from pta_replicator_amd.engine import ReplicaEngine
P, N, fl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0)
psrs, noise = headline_array(P, N)
eng = configure_engine(ReplicaEngine(psrs, seed=1), noise)
eng._gw = None
eng.prepare()
eng.prepare_td()
n, ld = eng.td_nst[0], eng.td_ld[0]
info = dv.zeros((P,), dtype=torch.int32)
need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, fl))
work = dv.empty((need,))
for _ in range(2):      # the second one is the one the classes are taken from (code objects, streams warm)
    eng.td_assemble()
    torch.cuda.synchronize()
    _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), fl, dv.ptr(work), need, dv.stream_ptr())
    torch.cuda.synchronize()
assert int(info.abs().sum().item()) == 0
