def make_fake_toas_fromMJDs(*a, **k):  # placeholder (simulate.py:17)
    raise RuntimeError("PINT is not available: oracle stubs only")
