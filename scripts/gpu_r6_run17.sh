cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import torch
from bench import build_engine
from pta_replicator_amd import device as dv
eng, psrs, noise = build_engine(68, 5000, seed=20260921)
eng.prepare_td()
out = dv.empty((1024, eng.n_toa))
def wall(fn, reps=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for rep in range(3):
    for beside in (True, False):
        eng.td_fill_beside_gwb = beside
        print("beside" if beside else "serial", round(wall(lambda: eng.generate_td(1024, out=out)), 3), flush=True)
PY
