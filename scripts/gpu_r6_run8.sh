# round 6: fused assembly + left-looking factorisation (pta_td_assemble_potrf): TD tests, then prepare_td fused vs two-step at 68 x 5000
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6h; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_td.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
python - <<'PY' 2>&1 | tail -8
import time, torch, sys
sys.path.insert(0, '.')
from bench import build_engine
eng, psrs, noise = build_engine(68, 5000, seed=20260921)
def wall(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for fused in (False, True, False, True):
    eng.td_fused = fused
    eng.prepare_td()
    print("fused" if fused else "two-step", eng.td_cov_kernel_used, [round(wall(eng.prepare_td), 2) for _ in range(4)])
eng.td_fused = True
eng.prepare_td()
out = eng.generate_td(256)
print("finite", bool(torch.isfinite(out).all()), float(out.std()))
PY
