#!/bin/bash
# Collects SQ counters for one bench.py step (two separate rocprofv3 --pmc runs).
# usage: tools/pmc_sq.sh [tag] [extra bench args...]
tag=${1:-sq}; shift
mkdir -p gpurun_out
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-workloads $@"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d gpurun_out/pmc_${tag}_a --output-format csv -- $B > gpurun_out/pmc_${tag}_a.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA -d gpurun_out/pmc_${tag}_b --output-format csv -- $B > gpurun_out/pmc_${tag}_b.log 2>&1 || exit 1
python - <<PY
import csv,glob,collections
for part in "ab":
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv" % part, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in acc.items():
            if "forward" in k or "segments" in k:
                print(k, {a: "%.4g"%b for a,b in v.items()})
PY
