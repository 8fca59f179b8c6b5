"""Quick check of the text-in-LDS forward kernel against the oracle (GPU box).  usage: python tools/lds_quick.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BWTGPU_FORWARD_PATH"] = "block"
import oracle  # noqa: E402
from bwt_amd import api  # noqa: E402

orc = oracle.oracle()
api.init(1)
rng = np.random.default_rng(1)
bad = 0
for name, arr, delim in [
    ("acgt 16384", np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=16384)], 0),
    ("acgt 65536", np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=65536)], 0),
    ("acgt 262144", np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=262144)], 0),
    ("acgt 262144 d255", np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=262144)], 255),
    ("acgt 100001 dD", np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100001)], ord("D")),
    ("bytes 65536", rng.integers(0, 255, size=65536, dtype=np.uint8), 255),
    ("bytes 30001", rng.integers(1, 256, size=30001, dtype=np.uint8), 0),
    ("ab 50000", np.frombuffer(b"ab", np.uint8)[rng.integers(0, 2, size=50000)], 255),
]:
    data = arr.astype(np.uint8).tobytes()
    want = orc.forward(data, delim)
    t = time.time()
    got = api.bwt_forward(data, delim)
    ok = got == want
    bad += not ok
    if not ok:
        diff = [i for i in range(len(want)) if got[i] != want[i]]
        print(name, "MISMATCH", len(diff), "rows differ, first", diff[:8])
    else:
        print(name, "ok %.1f ms" % ((time.time() - t) * 1e3))
sys.exit(1 if bad else 0)
