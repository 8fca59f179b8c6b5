#!/bin/bash
# A/B timing of library variants under build/x/ (forward ms and phase shares of the second launch).
# usage: tools/ab.sh "<bench args>" VARIANT...    (VARIANT "base" = the in-tree library)
args=$1; shift
for v in "$@"; do
  lib=$PWD/build/x/lib_$v.so; [ "$v" = base ] && lib=$PWD/bwt_amd/libbwtgpu.so
  BWTGPU_LIB=$lib BWTGPU_PHASES=${PHASES:-second} timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --no-other-workloads $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python3 - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % v).read().strip().splitlines()[0])
    print("%-10s fwd %8.2f ms inv %7.2f ms ok=%s %s" % (v, d["forward_ms"], d["inverse_ms"], d["roundtrip_verified"],
          {k: x for k, x in d["forward_phase_share"].items() if not isinstance(x, float) or x > 0.03}))
except Exception as e:
    print(v, "failed", e, open("gpurun_out/ab_%s.err" % v).read()[-300:])
PY
done
