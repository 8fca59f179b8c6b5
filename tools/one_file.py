#!/usr/bin/env python3
"""Transforms one file forth and back a few times through the batch API (for profiling a single
shape under rocprofv3).  usage: python tools/one_file.py <file> <block_size> [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bwt_amd import api

path, bs = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
raw = open(path, "rb").read()
api.init(1)
delim = api.find_unused_byte(raw)
for r in range(reps):
    t0 = time.perf_counter()
    f = api.forward_batch(raw, bs, delim)
    t1 = time.perf_counter()
    b = api.inverse_batch(f, bs, delim)
    t2 = time.perf_counter()
    print(f"rep {r}: forward {1e3 * (t1 - t0):.3f} ms, inverse {1e3 * (t2 - t1):.3f} ms, ok {b == raw}", flush=True)
