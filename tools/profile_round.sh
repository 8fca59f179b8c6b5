#!/bin/bash
# Produces the per-round profile artefacts on the GPU box (run through gpurun):
#   kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic (two separate --pmc runs) and wave-level
#   SQ counters (two more --pmc runs).
# usage: tools/profile_round.sh <tag> [workload]     e.g. r02_v1 mixed-4g-1m
tag=${1:-rXX}
wl=${2:-acgt-4g-256k}
out=gpurun_out/prof_${tag}_$wl
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-workloads --workload $wl"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- $B > $out/stats.log 2>&1
# (a run through the device-wide path can end in a fault inside the profiler's own exit handler, after it has written
# its files: what counts is whether the files are there)
ls $out/stats/*/*kernel_stats.csv > /dev/null 2>&1 || { echo stats failed; tail -5 $out/stats.log; exit 1; }
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-workloads --workload $wl"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch --output-format csv -- $B1 > $out/fetch.log 2>&1
ls $out/fetch/*/*counter_collection.csv > /dev/null 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write --output-format csv -- $B1 > $out/write.log 2>&1
ls $out/write/*/*counter_collection.csv > /dev/null 2>&1 || { echo write failed; exit 1; }
# wave-level counters (two more passes of one step each): instruction mix, issue and LDS pipe activity
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace -d $out/sqa --output-format csv -- $B1 > $out/sqa.log 2>&1
ls $out/sqa/*/*counter_collection.csv > /dev/null 2>&1 || { echo sq pass a failed; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --kernel-trace -d $out/sqb --output-format csv -- $B1 > $out/sqb.log 2>&1
ls $out/sqb/*/*counter_collection.csv > /dev/null 2>&1 || { echo sq pass b failed; exit 1; }
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
sq = collections.defaultdict(dict)
for name in ("sqa", "sqb"):
    f = max(glob.glob(out + "/%s/**/*counter_collection.csv" % name, recursive=True), key=os.path.getsize)
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "bwtgpu" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].replace("void ", "").split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, c), v in acc.items():
        sq[k][c] = v
json.dump(sq, open(out + "/sq.json", "w"), indent=1)
print(open(out + "/kernel_stats.csv").read()[:1500])
print(json.dumps(res, indent=1))
print(open(out + "/stats.log").read().strip().split("\n")[-1][:400])
PY
