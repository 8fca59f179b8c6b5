#!/bin/bash
# Development aid: the inverse of 1 GiB of ACGT text at several block sizes, with and without the in-LDS path.
for bs in ${SIZES:-8192 16384 32768 65536 131072 262144}; do
  for path in general auto; do
    BWTGPU_INVERSE_PATH=$path timeout -k 10 120 python bench.py --workload acgt-4g-256k --bytes-per-gpu 1073741824 --block-size $bs --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-workloads 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$bs $path', 'fwd', round(d['forward_ms'], 2), 'inv', round(d['inverse_ms'], 2), d['roundtrip_verified'])" || exit 1
  done
done
