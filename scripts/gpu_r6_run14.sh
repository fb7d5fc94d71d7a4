cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import torch
from bench import build_engine
eng, psrs, noise = build_engine(68, 5000, seed=20260921)
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
print("gw grid factor first", wall(eng._prepare_gw_grid_factor))
eng._gw_grid_ready = False
print("gw grid factor second", wall(eng._prepare_gw_grid_factor))
print("prepare_td 1", wall(eng.prepare_td)); print("prepare_td 2", wall(eng.prepare_td))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); eng.prepare_td(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(14)
PY
