#!/bin/bash
# Produces the per-round profile artefacts on the GPU box (run through gpurun):
#   kernel stats (rocprofv3 --kernel-trace --stats) and HBM traffic (two separate --pmc runs).
# usage: tools/profile_round.sh <tag> [workload]     e.g. r02_v1 mixed-4g-1m
tag=${1:-rXX}
wl=${2:-acgt-4g-256k}
out=gpurun_out/prof_${tag}_$wl
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-workloads --workload $wl"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- $B > $out/stats.log 2>&1 || { echo stats failed; tail -5 $out/stats.log; exit 1; }
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-workloads --workload $wl"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch --output-format csv -- $B1 > $out/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write --output-format csv -- $B1 > $out/write.log 2>&1 || { echo write failed; exit 1; }
python - <<PY
import csv, glob, json, collections, os
out = "$out"
stats = max(glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getsize)
open(out + "/kernel_stats.csv", "w").write(open(stats).read())
res = collections.defaultdict(dict)
for name in ("fetch", "write"):
    f = max(glob.glob(out + "/%s/**/*counter_collection.csv" % name, recursive=True), key=os.path.getsize)
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "bwtgpu" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].replace("void ", "").split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, c), v in acc.items():
        res[k][c + "_KB"] = v
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(open(out + "/kernel_stats.csv").read()[:1500])
print(json.dumps(res, indent=1))
print(open(out + "/stats.log").read().strip().split("\n")[-1][:400])
PY
