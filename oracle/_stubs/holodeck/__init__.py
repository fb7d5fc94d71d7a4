"""Stand-in for holodeck — TEST INFRASTRUCTURE ONLY (deterministic.py:8)."""
from . import utils, cosmo  # noqa: F401
