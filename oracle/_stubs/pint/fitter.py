class Fitter:  # placeholder (simulate.py:18)
    pass
