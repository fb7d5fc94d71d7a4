class Pulsar:  # placeholder (simulate.py:20)
    def __init__(self, *a, **k):
        raise RuntimeError("enterprise is not available: oracle stubs only")
