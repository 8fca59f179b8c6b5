#!/bin/bash
# HBM bytes per kernel (two separate rocprofv3 --pmc passes) of one bench.py step.  usage: tools/pmc_traffic.sh <tag> [bench args...]
tag=$1; shift
out=gpurun_out/pt_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-workloads"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch --output-format csv -- $B "$@" > $out/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write --output-format csv -- $B "$@" > $out/write.log 2>&1 || { echo write failed; exit 1; }
python3 - $out <<'PY'
import csv, glob, json, collections, os, sys
out = sys.argv[1]
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
for k, v in res.items():
    print("%-60s fetch %8.2f GB  write %8.2f GB" % (k.replace("bwtgpu::", ""), v.get("FETCH_SIZE_KB", 0) * 1024 / 1e9, v.get("WRITE_SIZE_KB", 0) * 1024 / 1e9))
PY
