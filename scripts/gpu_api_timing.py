"""Wall time of ONE realisation of the 68 x 5000 array through the drop-in add_* API (host-owned pulsars, NumPy legacy draws on the
host in the reference's order, PCIe transfers included): the reference-style loop of per-pulsar calls and the list form."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import api_mode_timing, headline_array
psrs, noise = headline_array(68, 5000)
print(json.dumps(api_mode_timing(psrs, noise)))
