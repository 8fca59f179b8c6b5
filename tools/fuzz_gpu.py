"""Randomised differential test of the GPU path against the CPU oracle (run on a GPU box).

usage: python tools/fuzz_gpu.py [trials] [seed] [big|lds|origin]
Covers: alphabet sizes incl. powers of two (delimiter above / below / inside the alphabet),
runs, periodic text, repeated blocks, lengths around the internal size thresholds, batches;
every trial through one of the forward paths (the batch kernels, the device-wide path, or the
library's own choice) and one of the inverse routes (blocks over at most four symbols inside LDS,
or the table kernels for everything).
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from bwt_amd import api  # noqa: E402

orc = oracle.oracle()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 500
BIG_ONLY = len(sys.argv) > 3 and sys.argv[3] == "big"  # only long blocks and many-block batches
ORIGIN_ONLY = len(sys.argv) > 3 and sys.argv[3] == "origin"  # the origin-pointer format (sorted rotations), forward and back
LDS_ONLY = len(sys.argv) > 3 and sys.argv[3] == "lds"  # only the shapes the in-LDS kernels take: blocks of 2-256 KiB, batch path
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nrng = np.random.default_rng(rng.randrange(1 << 30))
prng = random.Random(12345)  # forward path per trial, drawn apart so that the inputs do not depend on it
SIZES = [1, 2, 3, 7, 16, 17, 18, 63, 64, 65, 129, 255, 256, 257, 1000, 4095, 4096, 4097, 4863, 4864, 4865, 5119, 5120,
         5121, 8191, 8192, 8193, 12345, 20000, 40000, 70000]


def make(n):
    kind = rng.choice(["uniform", "runs", "periodic", "tiled", "lowent", "allsame"])
    sigma = rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 64, 100, 255, 256])
    base = rng.randrange(0, 257 - sigma)
    if kind == "uniform":
        a = nrng.integers(0, sigma, size=n)
    elif kind == "runs":
        a = np.repeat(nrng.integers(0, sigma, size=n // 3 + 1), nrng.integers(1, 40, size=n // 3 + 1))[:n]
        if a.size < n:
            a = np.resize(a, n)
    elif kind == "periodic":
        p = rng.choice([1, 2, 3, 7, 64, 100])
        a = np.resize(nrng.integers(0, sigma, size=p), n)
    elif kind == "tiled":
        p = max(1, n // rng.choice([2, 3, 7]))
        a = np.resize(nrng.integers(0, sigma, size=p), n)
    elif kind == "lowent":
        a = nrng.integers(0, sigma, size=n)
        a[nrng.random(n) < 0.95] = 0
    else:
        a = np.zeros(n, dtype=np.int64)
    return (a + base).astype(np.uint8).tobytes(), kind, sigma, base


fails = 0
for t in range(trials):
    n = rng.choice(SIZES) if rng.random() < 0.7 else rng.randrange(1, 30000)
    big = rng.random() * (0.1 if BIG_ONLY else 1.0)
    if big < 0.05:    # one long block: several buckets, the 1024-thread kernel
        n = rng.randrange(100000, 600000)
    elif big < 0.10:  # more blocks than compute units: the 256-thread batch kernel
        n = rng.randrange(280, 600) * rng.choice([4096, 8192, 12000])
    if ORIGIN_ONLY:
        n = rng.choice(SIZES) if rng.random() < 0.6 else rng.randrange(1, 120000)
        data, kind, sigma, base = make(n)
        bs = rng.choice([n, max(1, n // 2), max(1, n // 3 + 1), 1 + rng.randrange(n), 4096, 7 * 64])
        origin = oracle.origin()
        cols, rows = [], []
        for off in range(0, n, bs):
            last, row = origin.forward(data[off:off + bs])
            cols.append(last)
            rows.append(row)
        got_last, got_rows = api.forward_origin_batch(data, bs)
        ok = got_last == b"".join(cols) and list(got_rows) == rows
        back_ok = ok and api.inverse_origin_batch(got_last, bs, got_rows) == data
        if not (ok and back_ok):
            fails += 1
            print("FAIL", t, "n", n, "bs", bs, kind, "sigma", sigma, "base", base, "fwd", ok, "inv", back_ok, flush=True)
            if fails > 10:
                break
        if t % 100 == 99:
            print("trial", t + 1, "fails", fails, flush=True)
        continue
    if LDS_ONLY:
        # blocks of 16 KiB .. 256 KiB (at most 64 KiB unless the alphabet is small: the text then fits the LDS both
        # ways), a handful of them plus a ragged tail, through the batch kernels
        small = rng.random() < 0.6
        bs = rng.choice([2048, 2049, 3000, 4096, 6000, 8192, 12000, 16384, 16385, 20000, 28671, 28672, 32768, 49151, 49152, 65535, 65536, 65537, 100000,
                         131072, 262143, 262144]) if small \
            else rng.choice([2048, 2500, 4096, 7000, 8192, 16384, 20000, 32768, 49151, 49152, 60000, 65535, 65536])
        n = bs * rng.randrange(1, 5) + rng.choice([0, 1, 17, bs // 2, bs - 1])
        n = min(n, 700000)
        data, kind, sigma, base = make(n)
        if small:
            a = np.frombuffer(data, dtype=np.uint8).astype(np.int64) - base
            sg = rng.choice([1, 2, 3, 4, 4, 4, 5])
            data = ((a % sg) * rng.choice([1, 3, 60]) + base // 4).astype(np.uint8).tobytes()
            sigma = sg
        os.environ["BWTGPU_FORWARD_PATH"] = "block"
        os.environ["BWTGPU_INVERSE_PATH"] = prng.choice(["auto", "packed", "general"])
        try:
            want = orc.forward_buffer(data, bs, 8)
        except ValueError:
            continue
        delim = want[0]
        got = api.forward_batch(data, bs, delim)
        ok = got == want[1:]
        back_ok = ok and api.inverse_batch(got, bs, delim) == data
        if not (ok and back_ok):
            fails += 1
            print("FAIL", t, "n", n, "bs", bs, kind, "sigma", sigma, "base", base, "delim", delim, "fwd", ok, "inv", back_ok, flush=True)
            if fails > 10:
                break
        if t % 20 == 19:
            print("trial", t + 1, "fails", fails, flush=True)
        continue
    data, kind, sigma, base = make(n)
    mode = rng.random()
    path = prng.choice(["auto", "auto", "block", "grid"])
    os.environ["BWTGPU_FORWARD_PATH"] = path
    os.environ["BWTGPU_INVERSE_PATH"] = prng.choice(["auto", "packed", "general"])  # in-LDS inverse of small alphabets (also for few blocks), or the table kernels only
    if 0.05 <= big < 0.10:
        bs = n // rng.randrange(270, 280)
        try:
            want = orc.forward_buffer(data, bs, 8)
        except ValueError:
            continue
        delim = want[0]
        got = api.forward_batch(data, bs, delim)
        ok = got == want[1:]
        back_ok = ok and api.inverse_batch(got, bs, delim) == data
    elif mode < 0.6:  # whole-buffer API with the reference's delimiter choice, random block size
        bs = rng.choice([n, max(1, n // 2), max(1, n // 3 + 1), 1 + rng.randrange(n), 4096, 8192])
        try:
            want = orc.forward_buffer(data, bs, 4)
        except ValueError:  # all 256 byte values occur: no delimiter
            continue
        delim = want[0]
        got = api.forward_batch(data, bs, delim)
        ok = got == want[1:]
        back_ok = ok and api.inverse_batch(got, bs, delim) == data
    else:  # single block with an arbitrary delimiter (may occur in the data, may be above or below everything)
        delim = rng.choice([0, 255, base, base + sigma - 1, min(255, base + sigma), max(0, base - 1), rng.randrange(256)])
        want = orc.forward(data, delim)
        got = api.bwt_forward(data, delim)
        ok = got == want
        back_ok = ok and api.bwt_inverse(want, delim) == orc.inverse(want, delim)
        bs = n
    if not (ok and back_ok):
        fails += 1
        print("FAIL", t, "n", n, "bs", bs, kind, "sigma", sigma, "base", base, "delim", delim, "path", path, "fwd", ok,
              "inv", back_ok, flush=True)
        if fails > 10:
            break
    if t % 100 == 99:
        print("trial", t + 1, "fails", fails, flush=True)
print("DONE trials", trials, "fails", fails)
sys.exit(1 if fails else 0)
