#!/bin/bash
# Runs bench.py over every workload (no CPU baseline) and prints a compact table.
# usage: tools/bench_all.sh [tag] [workloads...]
tag=${1:-run}; shift
wl=${@:-acgt-4g-256k random-1g-64k period7-1g-256k mixed-4g-1m}
mkdir -p gpurun_out
for w in $wl; do
  timeout -k 10 400 python -u bench.py --steps 2 --warmup 1 --workload $w --no-cpu-baseline --no-e2e --no-other-workloads > gpurun_out/bench_${tag}_$w.log 2>&1 || { echo "FAILED $w"; tail -5 gpurun_out/bench_${tag}_$w.log; exit 1; }
  tail -1 gpurun_out/bench_${tag}_$w.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-18s rt=%6.2f GB/s fwd=%8.1f ms inv=%7.1f ms ok=%s' % (d['config']['workload'], d['value'], d['forward_ms'], d['inverse_ms'], d['roundtrip_verified']), d.get('forward_phase_share'))"
done
