"""one generate_td(1024) of the 68 x 5000 headline array, last in the process (for a rocprofv3 kernel trace: which kernels sit in front of the product)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import device as dv
eng, psrs, noise = build_engine(68, 5000, seed=20260921)
eng.prepare_td()
out = dv.empty((1024, eng.n_toa))
for _ in range(3):
    eng.generate_td(1024, out=out)
    torch.cuda.synchronize()
