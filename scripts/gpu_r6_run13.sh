# round 6: do 3-4 chains lose because HIP multiplexes the streams onto 4 hardware queues?  GPU_MAX_HW_QUEUES A/B
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for q in 4 8 16; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q python - <<'PY'
import os, sys, time, json
sys.path.insert(0, '.')
import torch
from bench import configure_engine, headline_array
from pta_replicator_amd import _lib, device as dv
from pta_replicator_amd.engine import ReplicaEngine
L, S, LA, D64 = _lib.POTRF_LEFT, _lib.POTRF_LEFT_SPLIT, _lib.POTRF_DIAG_AHEAD, _lib.POTRF_DIAG64
psrs, noise = headline_array(68, 5000)
eng = configure_engine(ReplicaEngine(psrs, seed=1), noise); eng._gw = None; eng.prepare(); eng.prepare_td()
P, n, ld = 68, eng.td_nst[0], eng.td_ld[0]
info = dv.zeros((P,), dtype=torch.int32)
for name, fl in (("left c2", L), ("left c3", L | _lib.POTRF_CHAINS(3)), ("left c4", L | _lib.POTRF_CHAINS(4)),
                 ("left+diagLA c2", L | LA), ("left+diagLA c3", L | LA | _lib.POTRF_CHAINS(3)), ("left+diagLA c4", L | LA | _lib.POTRF_CHAINS(4)),
                 ("right LA c2", LA), ("right LA c3", LA | _lib.POTRF_CHAINS(3)), ("right LA c4", LA | _lib.POTRF_CHAINS(4))):
    need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, fl)); work = dv.empty((need,))
    ts = []
    for _ in range(4):
        eng.td_assemble(); torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), fl, dv.ptr(work), need, dv.stream_ptr())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name:28s} {min(ts)*1e3:7.2f} ms  frac {P*5000.0**3/3/min(ts)/1e12/78.6:.4f}  bad {int(info.abs().sum().item())}", flush=True)
PY
done
