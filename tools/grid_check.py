#!/usr/bin/env python3
"""Bring-up check of the device-wide forward path: every case runs through both forward paths
(BWTGPU_FORWARD_PATH=grid / block) and the two results are compared; prints both timings.  (Development aid: the parity suite, tests/test_gpu_parity.py,
runs every one of its cases through both paths against the oracle.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bwt_amd import api, synth

print("init ->", api.init(1), flush=True)


def run(path, data, bs, delim, reps=3):
    os.environ["BWTGPU_FORWARD_PATH"] = path
    best = 1e9
    out = None
    for _ in range(reps):
        t = time.perf_counter()
        out = api.forward_batch(data, bs, delim)
        best = min(best, time.perf_counter() - t)
    return out, best


def check(name, data, bs, delim):
    got_g, tg = run("grid", data, bs, delim)
    got_b, tb = run("block", data, bs, delim)
    ok = got_g == got_b
    print(f"{name:28s} n={len(data):9d} bs={bs:8d} grid {tg*1e3:8.2f} ms  block {tb*1e3:8.2f} ms  "
          f"{'SAME' if ok else 'DIFFERENT'}", flush=True)
    if not ok:
        g, w = np.frombuffer(got_g, np.uint8), np.frombuffer(got_b, np.uint8)
        bad = np.nonzero(g != w)[0]
        print("  first mismatches at", bad[:10], "of", bad.size, "grid", g[bad[:10]], "block", w[bad[:10]])
        sys.exit(1)


for data, bs in [(b"banana", 6), (b"banana", 3), (b"a", 1), (b"abracadabra" * 3, 7), (b"aaaaaaaaaa", 10)]:
    check("tiny", data, bs, ord("~"))
for kind in ["random255", "acgt", "period7", "period1", "period64"]:
    for n, bs in [(1000, 1000), (5000, 1024), (70000, 65536), (262144, 262144), (600000, 262144), (1048576, 1048576)]:
        d = synth.generate(kind, 99, 7, n).tobytes()
        check(kind, d, bs, 0xFF if kind in ("random255", "period64") else 0)
for f, delim in [("alice29.txt", 0), ("ptt5", 10), ("lcet10.txt", 0), ("plrabn12.txt", 0), ("sum", 0xEB), ("cp.html", 0)]:
    raw = open(os.path.join("data/canterbury_corpus", f), "rb").read()
    for bs in [65536, 1048576]:
        check(f, raw, bs, delim)
for f, delim, sizes in [("fields.c", 0, [16384]), ("xargs.1", 0, [16384]), ("grammar.lsp", 0, [16384]),
                        ("lcet10.txt", 0, [16384, 32768, 49152]), ("plrabn12.txt", 0, [16384, 32768, 49152]),
                        ("ptt5", 10, [16384, 32768]), ("alice29.txt", 0, [8192, 16384, 32768])]:
    raw = open(os.path.join("data/canterbury_corpus", f), "rb").read()
    for bs in sizes:
        check(f, raw, bs, delim)
for n, bs in [(4 << 20, 262144), (6 << 20, 524288), (16 << 20, 1048576), (2 << 20, 2097151)]:
    d = synth.generate("acgt", 5, 0, n).tobytes()
    check("acgt many", d, bs, 0)
print("ALL OK")
