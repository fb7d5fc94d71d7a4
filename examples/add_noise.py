#!/usr/bin/env python3
"""The reference's example notebook (examples/add_noise.ipynb, cells 2-14 and 17-23) as a script on the MI355X path.

Same calls, same keyword arguments, same seeds - only the import lines change (pta_replicator -> pta_replicator_amd).
Runs with PINT-backed pulsars when PINT is installed and with array-backed ones otherwise.  Needs an AMD GPU.

    python examples/add_noise.py /path/to/pta_replicator        # the checkout that holds test_partim/ and noise_dicts/
"""
import json
import sys

import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout without installing

from pta_replicator_amd.white_noise import add_measurement_noise, add_jitter
from pta_replicator_amd.red_noise import add_red_noise, add_gwb
from pta_replicator_amd.simulate import load_pulsar, make_ideal, simulate_pulsar
from pta_replicator_amd.engine import ReplicaEngine

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

# cell 2: one pulsar from par + tim
psr = load_pulsar(f"{ref}/test_partim/par/B1855+09.par", f"{ref}/test_partim/tim/B1855+09.tim")
psrs = [psr]

# cells 5-6: noise dictionary -> per-backend lists
with open(f"{ref}/noise_dicts/ng15_dict.json") as fp:
    noise_params = json.load(fp)
noise_dict = {}
for p in [q.name for q in psrs]:
    d = noise_dict[p] = {"log10_equads": [], "efacs": [], "log10_ecorrs": []}
    for ky, val in noise_params.items():
        if p in ky:
            if "equad" in ky:
                d["log10_equads"].append([ky.replace(p + "_", "").replace("_log10_t2equad", ""), val])
            if "efac" in ky:
                d["efacs"].append([ky.replace(p + "_", "").replace("_efac", ""), val])
            if "ecorr" in ky:
                d["log10_ecorrs"].append([ky.replace(p + "_", "").replace("_log10_ecorr", ""), val])
            if "gamma" in ky:
                d["rn_gamma"] = val
            if "log10_A" in ky:
                d["rn_log10_amp"] = val
    for k in ("log10_equads", "efacs", "log10_ecorrs"):
        d[k] = np.array(d[k])

# cell 8-9: seeds, then EFAC/EQUAD + ECORR + red noise per pulsar
seed_efac_equad, seed_jitter, seed_red, seed_gwb = 10660, 17763, 19870, 16672
for ii, psr in enumerate(psrs):
    make_ideal(psr)
    nd = noise_dict[psr.name]
    add_measurement_noise(psr, efac=nd["efacs"][:, 1].astype(float), log10_equad=nd["log10_equads"][:, 1].astype(float),
                          flagid="f", flags=nd["efacs"][:, 0], seed=seed_efac_equad + ii)
    add_jitter(psr, log10_ecorr=nd["log10_ecorrs"][:, 1].astype(float), flagid="f", flags=nd["log10_ecorrs"][:, 0],
               coarsegrain=1.0 / 86400.0, seed=seed_jitter + ii)
    add_red_noise(psr, log10_amplitude=nd["rn_log10_amp"], spectral_index=nd["rn_gamma"], components=30, seed=seed_red + ii)
    print(ii, psr.name)

# cell 11: common GWB
add_gwb(psrs, log10_amplitude=-15, spectral_index=13. / 3., seed=seed_gwb)

# cells 13-14: provenance + per-signal series
for psr in psrs:
    print(sorted(psr.added_signals))
    for sig, dt in psr.added_signals_time.items():
        print(f"  {sig:45s} rms = {np.sqrt(np.mean(np.asarray(dt.to_value('us')) ** 2)):.4f} us")
    print(f"  total residual rms = {np.sqrt(np.mean(psr.residuals.resids_value ** 2)) * 1e6:.4f} us")

# cells 17-21: custom observation times
psrs2 = [simulate_pulsar(f"{ref}/test_partim/par/{n}.par", np.linspace(54000, 59000, 100), np.ones(100) * 0.5) for n in ("B1855+09", "J1909-3744")]
for ii, psr in enumerate(psrs2):
    make_ideal(psr)
    add_measurement_noise(psr, efac=1.0, log10_equad=-6.5, seed=seed_efac_equad + ii)
    add_red_noise(psr, log10_amplitude=-14.0, spectral_index=3.0, libstempo_convention=True, components=30, seed=seed_red + ii)
add_gwb(psrs2, log10_amplitude=-15, spectral_index=13. / 3., seed=seed_gwb)
for psr in psrs2:
    print(psr.name, sorted(psr.added_signals))

# beyond the notebook: 1000 independent realisations of the same two-pulsar array in one call
eng = ReplicaEngine(psrs2, seed=1)
eng.set_white_noise(efac=1.0, log10_equad=-6.5)
eng.set_red_noise(-14.0, 3.0)
eng.set_gwb(-15, 13. / 3.)
ens = eng.generate(1000)
print("ensemble:", tuple(ens.shape), "rms per pulsar [us]:", [float(x.std()) * 1e6 for x in eng.split(ens)])
