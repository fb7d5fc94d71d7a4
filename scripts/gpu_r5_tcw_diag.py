"""Which stream bounds the walking assembly kernel?  A probe build of the library (-DPTA_TCW_DIAG: scripts/probe_src/libpta_tcw_diag.so)
runs k_td_cov_walk on the 68 x 5000 array with one of its three streams removed (results wrong by construction):
    full | no global stores | no fragment loads inside the steps | no products (one fma per former MFMA)
for the 64-column (two workgroups per CU) and the 128-column (one per CU, two steps ahead) form.  -> profiles/r05_tcw_diag.txt
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DPTA_TCW_DIAG -shared -o scripts/probe_src/libpta_tcw_diag.so pta_replicator_amd/csrc/*.hip
    PTA_REPLICATOR_AMD_LIB=scripts/probe_src/libpta_tcw_diag.so python scripts/gpu_r5_tcw_diag.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "tcw_diag" in os.environ.get("PTA_REPLICATOR_AMD_LIB", ""), "run with PTA_REPLICATOR_AMD_LIB=scripts/probe_src/libpta_tcw_diag.so"
import bench
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
eng.prepare_td()
res = {}
for base, bname in ((1, "walk64"),):
    for diag, dname in ((0, "full"), (1, "no_stores"), (2, "no_fragment_loads"), (3, "no_products"), (4, "products_and_loads_only"), (5, "products_only"), (6, "products_loads_and_a_600_cycle_sleep_per_step")):
        eng.td_cov_walk_variant = base + 16 * diag
        eng._td_walk_items = None
        eng.td_assemble(kernel="walk")
        t = min(bench._wall(lambda: eng.td_assemble(kernel="walk")) for _ in range(5))
        res[f"{bname}_{dname}_ms"] = round(t * 1e3, 3)
print(json.dumps(res))
