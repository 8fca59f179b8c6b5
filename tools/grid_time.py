#!/usr/bin/env python3
"""Times the device-wide forward path on a few single files / blocks (host-pointer API, best of 5).
usage: python tools/grid_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BWTGPU_FORWARD_PATH"] = "grid"
from bwt_amd import api, synth  # noqa: E402

api.init(1)
cases = []
for f, delim in [("alice29.txt", 0), ("ptt5", 10), ("lcet10.txt", 0)]:
    cases.append((f, open(os.path.join("data/canterbury_corpus", f), "rb").read(), 1 << 20, delim))
cases.append(("random 1 MiB", synth.generate("random255", 99, 7, 1 << 20).tobytes(), 1 << 20, 255))
cases.append(("acgt 16 x 1 MiB", synth.generate("acgt", 5, 0, 16 << 20).tobytes(), 1 << 20, 0))
cases.append(("alice29 128K", open("data/canterbury_corpus/alice29.txt", "rb").read(), 131072, 0))
for name, data, bs, delim in cases:
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        api.forward_batch(data, bs, delim)
        best = min(best, time.perf_counter() - t)
    print("%-18s %8.3f ms" % (name, best * 1e3), flush=True)
