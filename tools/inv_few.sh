#!/bin/bash
# Development aid: few blocks of ACGT text at 256 KiB -- tiled table build + walks (auto) against the in-LDS inverse (packed).
for mib in 1 4 16 32; do
  for path in auto packed; do
    BWTGPU_INVERSE_PATH=$path timeout -k 10 120 python bench.py --workload acgt-4g-256k --bytes-per-gpu $((mib * 1048576)) --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-other-workloads 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mib MiB $path', 'fwd', round(d['forward_ms'], 3), 'inv', round(d['inverse_ms'], 3), d['roundtrip_verified'])" || exit 1
  done
done
