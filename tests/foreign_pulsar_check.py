"""Run by tests/test_gpu_parity.py::test_foreign_pulsar_objects_with_astropy_style_units in a FRESH interpreter whose
sys.path starts with oracle/_stubs, so that ``import astropy`` inside pta_replicator_amd._compat succeeds (a stand-in, but a
foreign one: its Quantity / TimeDelta classes are not the product's) and the product's HAVE_ASTROPY branch is the one that runs.
The pulsars are foreign objects too - the float64/longdouble ``MockTOAs`` container the reference itself is run on under the
same stubs (oracle/run_reference.py), wrapped in a bare class with the duck-type surface of SURVEY.md §8b - i.e. what a
PINT-backed ``pta_replicator.simulate.SimulatedPulsar`` looks like to the injection functions.  The reference's test recipe
(tests/test_against_libstempo.py:19-53) must then reproduce the fixtures of the unmodified reference."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle", "_stubs"), ROOT, HERE]

import numpy as np  # noqa: E402
import astropy.units as au  # noqa: E402  (the stand-in)

import pta_replicator_amd._compat as compat  # noqa: E402
assert compat.HAVE_ASTROPY and compat.u is au, "the astropy branch of _compat did not engage"

from helpers import load, mjd_ld, relrms  # noqa: E402
from oracle.run_reference import MockTOAs  # noqa: E402
from pta_replicator_amd.deterministic import add_cgw  # noqa: E402
from pta_replicator_amd.red_noise import add_gwb, add_red_noise  # noqa: E402
from pta_replicator_amd.white_noise import add_jitter, add_measurement_noise  # noqa: E402


class ForeignPulsar:
    """name / loc / toas / added_signals(+_time) / update_added_signals / update_residuals: nothing else."""

    def __init__(self, name, toas, loc):
        self.name, self.toas, self.loc = name, toas, loc
        self.added_signals, self.added_signals_time = {}, {}       # what make_ideal leaves behind (simulate.py:200-201)

    def update_added_signals(self, signal_name, param_dict, dt=None):
        if signal_name in self.added_signals:
            raise ValueError(f"{signal_name} already exists in the model.")
        self.added_signals[signal_name] = param_dict
        if dt is not None:
            self.added_signals_time[signal_name] = dt

    def update_residuals(self):
        pass                                                        # PINT's job; the oracle residual is sum(shifts) - mean


z = load("c1_small.npz")
tag = "raw_"
psrs = [ForeignPulsar(str(z[tag + "names"][i]), MockTOAs(mjd_ld(z, tag, i), z[f"{tag}err_us_{i}"]),
                      {"RAJ": float(z[tag + "raj_hours"][i]), "DECJ": float(z[tag + "decj_deg"][i])}) for i in range(3)]
add_gwb(psrs, -14, 4.33, seed=123456)
for ii, psr in enumerate(psrs):
    add_measurement_noise(psr, efac=1.00, log10_equad=None, seed=54321 + ii, tnequad=False)
    add_jitter(psr, log10_ecorr=np.log10(3e-7), seed=54321 + ii)
for ii, psr in enumerate(psrs):
    add_red_noise(psr, -15, 4.2, components=30, Tspan=None, seed=12345 + ii, libstempo_convention=True)
for psr in psrs:
    add_cgw(psr, gwtheta=np.pi / 2, gwphi=2.5, mc=1e9, dist=5.0, fgw=1e-8, phase0=0.5, psi=1.5, inc=np.pi / 4, pdist=1.0, pphase=None,
            psrTerm=True, evolve=True, phase_approx=False, tref=53000 * 86400)
worst = 0.0
for i, psr in enumerate(psrs):
    assert isinstance(psr.added_signals_time[f"{psr.name}_red_noise"], au.Quantity)      # foreign Quantity objects were produced
    worst = max(worst, relrms(psr.toas.residuals_s(), z[tag + "residuals"][i]))
    for key in ("gwb", "measurement_noise", "jitter", "red_noise"):
        worst = max(worst, 10 * relrms(np.asarray(psr.added_signals_time[f"{psr.name}_{key}"].value, dtype=np.float64), z[tag + key][i]))
assert worst < 1e-10, worst
print(f"FOREIGN_OK {worst:.3e}")
