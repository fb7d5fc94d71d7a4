"""Stand-in for ``astropy.time`` — TEST INFRASTRUCTURE ONLY (see units.py)."""


class TimeDelta:
    """Holds a Quantity; MockTOAs.adjust_TOAs reads ``.quantity``."""

    def __init__(self, quantity):
        self.quantity = quantity
