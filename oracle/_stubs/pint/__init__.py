"""Stand-in package for PINT — TEST INFRASTRUCTURE ONLY.

Only class *names* are needed: pta_replicator/simulate.py evaluates
``models.TimingModel`` / ``toa.TOAs`` / ``Residuals`` in dataclass annotations at
import time (simulate.py:29-31).  The oracle never calls into them.
"""
from . import residuals, toa, models, simulation, fitter  # noqa: F401
