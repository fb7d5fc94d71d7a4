"""Multi-GPU tests (run with -m gpu): skipped on boxes with fewer GPUs than ranks - the development box and the driver's 1-GPU lease
have one - so that the first multi-GPU node that runs the suite exercises the RCCL point-to-point schedule of
pta_replicator_amd.distributed on hardware.  The multi-rank control flow is covered on CPU by
tests/test_distributed_gloo.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world,total", [(2, 37), (8, 37), (8, 5)])     # ragged shards; (8, 5): three ranks own no realisation
def test_rccl_ranks_reproduce_the_single_gpu_ensemble(world, total, tmp_path):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    from bench import configure_engine, headline_array
    from pta_replicator_amd.engine import ReplicaEngine
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multi_gpu_worker.py"), str(tmp_path), str(total)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(tmp_path / "rank0.npz")
    assert int(z["ranks_seen"]) == world
    psrs, noise = headline_array(5, 700)
    eng = configure_engine(ReplicaEngine(psrs, seed=31), noise).prepare()
    ref = eng.generate(total, r0=3).cpu().numpy()
    for k in ("a", "b", "c"):                      # plain gather, pipelined generate + gather, C-ABI gather: bit-identical
        assert np.array_equal(z[k], ref), k
    eng.prepare_td()
    assert np.array_equal(z["d"], eng.generate_td(total, r0=3).cpu().numpy())


def test_c_abi_gather_single_rank_and_rccl_communicator():
    """pta_gather_rank0 with one rank (a strided device copy, no communicator) and a real one-rank ncclComm_t created through ctypes on
    PyTorch's librccl.so - what a ctypes-only integrator would do."""
    import torch
    from pta_replicator_amd import _lib, device as dv
    from pta_replicator_amd.distributed import RcclComm, gather_to_rank0_abi
    x = torch.randn((7, 33), dtype=torch.float64, device="cuda")
    assert torch.equal(gather_to_rank0_abi(x, 7), x)
    comm = RcclComm(0, 1, RcclComm.unique_id())
    try:
        assert comm.ptr.value
        assert torch.equal(gather_to_rank0_abi(x, 7, comm), x)
        # leading dimensions on a single rank
        big = torch.zeros((7, 40), dtype=torch.float64, device="cuda")
        _lib.call("pta_gather_rank0", comm.ptr, 0, 1, 0, dv.ptr(x), 7, 33, 33, dv.ptr(big), 40, dv.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(big[:, :33], x) and float(big[:, 33:].abs().sum()) == 0.0
    finally:
        comm.destroy()
    with pytest.raises(_lib.PtaError):
        _lib.call("pta_gather_rank0", None, 0, 2, 0, dv.ptr(x), 14, 33, 33, dv.ptr(x), 33, dv.stream_ptr())   # two ranks need a communicator


def test_bench_two_ranks_dry_run_on_one_device():
    """bench.py's N > 1 branch (env handling, disjoint realisation ranges, barrier + max-over-ranks timing, the all_gather of
    per-rank step times, the pipelined generate + gather to rank 0, ONE JSON line from rank 0, clean exit of every rank) was executed by
    no test.  Launched exactly as the driver launches it - python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 - with the
    two dry-run switches of bench.py (both ranks on cuda:0, gloo as the control plane: a 1-GPU box has no second device for RCCL) and a
    batch small enough for gloo's ~50 MB/s device-tensor transport.  Claims nothing about scaling."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PTA_BENCH_SINGLE_DEVICE="1", PTA_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "32", "--no-td", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 only, one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "realizations/s"
    assert d["rccl_ranks_seen"] == 2 and d["backend"] == "gloo"
    assert len(d["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in d["ms_per_step_per_rank"])
    # value = the realisations ALL ranks generated over the max-over-ranks time
    assert abs(d["value"] - 2 * 32 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    g = d["gathered_to_rank0"]
    assert "error" not in g, g
    assert g["realisations"] == 64 and g["realisations_per_s"] > 0
    assert d["config"]["parallelism"] == "replica-shard x2"
    roof = d["roofline"]
    assert len(roof) <= 24 and all(len(k) <= 40 for k in roof) and roof["kernel"] and roof["frac"] > 0
