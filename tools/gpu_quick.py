#!/usr/bin/env python3
"""Quick GPU bring-up check: a ladder of forward/inverse cases against the CPU oracle,
stopping at the first mismatch with enough detail to debug it.  (Development aid; the
real parity suite is tests/test_gpu_parity.py.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from bwt_amd import api, synth

orc = oracle.oracle()
print("init ->", api.init(1), flush=True)


def check(name, data, delim):
    t = time.time()
    got = api.bwt_forward(data, delim)
    t1 = time.time() - t
    want = orc.forward(data, delim)
    ok = got == want
    inv_ok = None
    if ok:
        t = time.time()
        back = api.bwt_inverse(want, delim)
        t2 = time.time() - t
        inv_ok = back == orc.inverse(want, delim)
    print(f"{name:40s} n={len(data):8d} fwd={'OK' if ok else 'MISMATCH'} ({t1*1e3:.1f} ms) inv={inv_ok}", flush=True)
    if not ok:
        g, w = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        bad = np.nonzero(g != w)[0]
        print("  first mismatches at", bad[:10], "of", bad.size, "got", g[bad[:10]], "want", w[bad[:10]])
        sys.exit(1)
    if not inv_ok:
        sys.exit(2)


check("banana", b"banana", ord("~"))
check("empty", b"", ord("~"))
check("a", b"a", ord("~"))
check("misuse ab~ab", b"ab~ab", ord("~"))
for kind in ["random255", "acgt", "period7", "period1", "period64"]:
    for n in [10, 64, 65, 1000, 5000, 65536, 262144]:
        d = synth.generate(kind, 99, 7, n).tobytes()
        check(f"{kind}", d, 0xFF if kind in ("random255", "period64") else 0)
raw = open("data/canterbury_corpus/alice29.txt", "rb").read()
check("alice29", raw, 0)
check("ptt5", open("data/canterbury_corpus/ptt5", "rb").read(), 10)
print("ALL OK")
