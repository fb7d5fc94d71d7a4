class Residuals:  # placeholder (simulate.py:14)
    def __init__(self, *a, **k):
        raise RuntimeError("PINT is not available: oracle stubs only")
