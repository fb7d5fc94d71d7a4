# round 6: is k_diag128 starved for LDS beside the tile products?  A/B with the 64-column recursion (k_potf2: 1 KB of LDS) on the diagonal blocks
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python - <<'PY'
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
for name, fl in (("left c2", L), ("left c2 diag64", L | D64), ("left c3 diag64", L | D64 | _lib.POTRF_CHAINS(3)), ("left c4 diag64", L | D64 | _lib.POTRF_CHAINS(4)),
                 ("left+diagLA c2 diag64", L | LA | D64), ("left+diagLA c3 diag64", L | LA | D64 | _lib.POTRF_CHAINS(3)), ("right LA c2", LA), ("right LA c2 diag64", LA | D64),
                 ("right LA c3 diag64", LA | D64 | _lib.POTRF_CHAINS(3)), ("left c1 diag64", L | D64 | _lib.POTRF_CHAINS(1)), ("left c1", L | _lib.POTRF_CHAINS(1))):
    need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, fl)); work = dv.empty((need,))
    ts = []
    for _ in range(4):
        eng.td_assemble(); torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), fl, dv.ptr(work), need, dv.stream_ptr())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name:28s} {min(ts)*1e3:7.2f} ms  frac {P*5000.0**3/3/min(ts)/1e12/78.6:.4f}  bad {int(info.abs().sum().item())}", flush=True)
PY
