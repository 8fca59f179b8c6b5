timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "whole_block or periodic_runs or mixed or benchmark_shapes or synthetic or blocks_above" > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
for w in "period7-1g-256k" "mixed-4g-1m" "mixed-4g-1m --kind text --bytes-per-gpu 805306368"; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-e2e --no-other-workloads > gpurun_out/b_x.json 2>gpurun_out/b_x.err
python - <<PY
import json
d=json.load(open("gpurun_out/b_x.json")); print(d["config"]["workload"], round(d["value"],2), round(d["forward_ms"],2), round(d["inverse_ms"],2), d["roundtrip_verified"], {k:v for k,v in d["forward_phase_share"].items() if v>0.03})
PY
done
