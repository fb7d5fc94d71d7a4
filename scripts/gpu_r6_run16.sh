# round 6: row-resident substitution (PTA_POTRF_SOLVE_ROWS): tests + A/B
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "potrf_workspace" 2>&1 | tail -4
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import torch
from bench import configure_engine, headline_array
from pta_replicator_amd import _lib, device as dv
from pta_replicator_amd.engine import ReplicaEngine
L, LA, RW = _lib.POTRF_LEFT, _lib.POTRF_DIAG_AHEAD, _lib.POTRF_SOLVE_ROWS
for P, N in ((68, 5000), (16, 10000), (3, 10000), (200, 5000)):
    psrs, noise = headline_array(P, N)
    eng = configure_engine(ReplicaEngine(psrs, seed=1), noise); eng._gw = None; eng.prepare(); eng.prepare_td()
    n, ld = eng.td_nst[0], eng.td_ld[0]
    info = dv.zeros((P,), dtype=torch.int32)
    ref = None
    for name, fl in (("left", L), ("left rows", L | RW), ("left c1", L | _lib.POTRF_CHAINS(1)), ("left c1 rows", L | RW | _lib.POTRF_CHAINS(1)), ("left c3 rows", L | RW | _lib.POTRF_CHAINS(3)),
                     ("right LA", LA), ("right LA rows", LA | RW), ("left rows nb768", L | RW | _lib.POTRF_NB(3)), ("left rows nb1536", L | RW | _lib.POTRF_NB(6)), ("left rows nb2048", L | RW | _lib.POTRF_NB(8))):
        need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, fl)); work = dv.empty((need,))
        ts = []
        for _ in range(4):
            eng.td_assemble(); torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), fl, dv.ptr(work), need, dv.stream_ptr())
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        Lm = torch.tril(eng.d_Ltd.view(P, n, ld)[P // 2, :, :n]).clone()
        if ref is None: ref = Lm
        err = float((Lm - ref).abs().max() / ref.abs().max())
        print(f"{P:4d} {N:6d} {name:18s} {min(ts)*1e3:8.2f} ms  frac {P*float(N)**3/3/min(ts)/1e12/78.6:.4f}  bad {int(info.abs().sum().item())}  diff {err:.1e}", flush=True)
        del work
    eng.d_Ltd = None; del eng; torch.cuda.empty_cache()
PY
