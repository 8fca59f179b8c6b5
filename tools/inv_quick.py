#!/usr/bin/env python3
"""Development aid: the small-alphabet inverse test alone, then timings of the inverse on 1 GiB of
ACGT text at 256 KiB and 64 KiB blocks with and without the in-LDS path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess

rc = subprocess.call([sys.executable, "-u", "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k", "small_alphabet or misuse",
                      "--timeout", "300", "--timeout-method", "thread"])
if rc:
    sys.exit(rc)
for path in ["general", "auto"]:
    env = dict(os.environ, BWTGPU_INVERSE_PATH=path)
    for wl in ["acgt-4g-256k", "random-1g-64k", "period7-1g-256k"]:
        out = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-other-workloads"],
                             env=env, capture_output=True, text=True)
        import json
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(path, wl, "value", round(d["value"], 2), "fwd", round(d.get("forward_ms", 0), 2), "inv", round(d.get("inverse_ms", 0), 2), flush=True)
        except Exception as e:
            print(path, wl, "failed", e, out.stdout[-500:], out.stderr[-1500:], flush=True)
            sys.exit(3)
