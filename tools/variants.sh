#!/bin/bash
# Times library variants built into build/vars/ (libbwtgpu_<tag>.so) on one workload.  usage: tools/variants.sh <workload> <tag>...
wl=$1; shift
for t in "$@"; do
  BWTGPU_LIB=build/vars/libbwtgpu_$t.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-e2e --no-other-workloads --workload $wl > gpurun_out/var_$t.json 2> gpurun_out/var_$t.err || { echo $t failed; tail -3 gpurun_out/var_$t.err; continue; }
  python3 - $t <<'PY'
import json,sys
d=json.load(open("gpurun_out/var_%s.json"%sys.argv[1]))
print(sys.argv[1], d["config"]["workload"], "value %.2f fwd %.3f inv %.3f ok %s" % (d["value"], d["forward_ms"], d["inverse_ms"], d["roundtrip_verified"]), d["forward_phase_share"].get("inverse_t16"))
PY
done
