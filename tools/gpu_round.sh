#!/bin/bash
# One GPU session: the whole GPU test suite, then the default bench line.  usage: tools/gpu_round.sh [tag]
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
