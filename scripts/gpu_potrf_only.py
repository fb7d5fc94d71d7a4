"""one assemble + factorisation of the 68 x 5000^2 TD covariances (for rocprofv3 timelines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import configure_engine, headline_array
from pta_replicator_amd.engine import ReplicaEngine
P, N = int(sys.argv[1]), int(sys.argv[2])
look = sys.argv[3] != "0"
psrs, noise = headline_array(P, N)
eng = configure_engine(ReplicaEngine(psrs, seed=1), noise)
eng._gw = None
eng.td_potrf_workspace = len(sys.argv) > 4 and sys.argv[4] in ("ws", "wsla")
if len(sys.argv) > 4 and sys.argv[4] == "wsla":
    from pta_replicator_amd import _lib
    eng.td_potrf_flags = _lib.POTRF_DIAG_AHEAD
eng.prepare_td(lookahead=look)
torch.cuda.synchronize()
