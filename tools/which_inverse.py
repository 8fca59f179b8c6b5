#!/usr/bin/env python3
"""Development aid: invert one DNA block and one block of random bytes through the C ABI (run it under
rocprofv3 --kernel-trace --stats with BWTGPU_INVERSE_PATH set to see which kernels a route launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bwt_amd import api

api.init(1)
rng = np.random.default_rng(1)
dna = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=70000)].tobytes()
other = rng.integers(1, 200, size=70000, dtype=np.uint8).tobytes()
for data in (dna, other):
    t = api.bwt_forward(data, 0)
    assert api.bwt_inverse(t, 0) == data
print("ok")
