#!/usr/bin/env python3
"""Round trip of ONE very long block (the device-wide forward path, 64-bit inverse tables), timed.
usage: python tools/big_block.py [kind] [MiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bwt_amd import api, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "acgt"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
api.init(1)
data = synth.generate(kind, synth.base_seed(3), 0, mib << 20).tobytes()
delim = api.find_unused_byte(data)
t0 = time.perf_counter()
f = api.forward_batch(data, len(data), delim)
t1 = time.perf_counter()
b = api.inverse_batch(f, len(data), delim)
t2 = time.perf_counter()
print(f"{kind} one block of {mib} MiB: forward {t1 - t0:.3f} s, inverse {t2 - t1:.3f} s (host copies included), "
      f"round trip {'ok' if b == data else 'MISMATCH'}", flush=True)
sys.exit(0 if b == data else 1)
