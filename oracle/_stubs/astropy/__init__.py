"""Stand-in package for astropy — TEST INFRASTRUCTURE ONLY (see units.py)."""
from . import units  # noqa: F401
from . import time  # noqa: F401
from . import constants  # noqa: F401
