#!/bin/bash
# Per-kernel times of one bench.py invocation (rocprofv3 --kernel-trace --stats).  usage: tools/kstats.sh <tag> [bench args...]
tag=$1; shift
out=gpurun_out/ks_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-workloads "$@" > $out.log 2>&1 || { echo failed; tail -5 $out.log; exit 1; }
f=$(ls -S $(find $out -name '*kernel_stats.csv') | head -1)  # the largest: the bench process itself, not a child
cp $f $out.csv
python3 - $out.csv $out.log <<'PY'
import csv, json, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bwtgpu" in r["Name"]:
        print("%-72s %3s x %9.3f ms" % (r["Name"].replace("bwtgpu::", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e6))
line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
d = json.loads(line)
print(d["config"]["workload"], "value", round(d["value"], 2), "fwd", round(d["forward_ms"], 2), "inv", round(d["inverse_ms"], 2),
      {k: v for k, v in d["forward_phase_share"].items() if not isinstance(v, float) or v > 0.03 or k.startswith("lds_") or k.startswith("inverse_")})
PY
