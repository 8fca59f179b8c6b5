#!/bin/bash
# File-to-file throughput of the command-line tools on a large synthetic file in tmpfs
# (PCIe, file reads and writes included).  usage: tools/e2e_file.sh [kind] [MiB] [block_size]
kind=${1:-acgt}; mib=${2:-1024}; bs=${3:-262144}
d=$(mktemp -d /dev/shm/bwt_e2e.XXXXXX) || exit 1
trap 'rm -rf "$d"' EXIT
python - "$kind" "$mib" "$d/in" <<'PY' || exit 1
import sys
sys.path.insert(0, ".")
from bwt_amd import synth
kind, mib, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
with open(path, "wb") as f:
    for i in range(0, mib, 64):
        n = min(64, mib - i) << 20
        f.write(synth.generate(kind, synth.base_seed(3), i << 20, n).tobytes())
PY
t0=$(date +%s.%N)
./build/bwt "$d/in" "$d/fwd" "$bs" || exit 1
t1=$(date +%s.%N)
./build/inverse_bwt "$d/fwd" "$d/back" "$bs" || exit 1
t2=$(date +%s.%N)
cmp "$d/in" "$d/back" || { echo "ROUND TRIP MISMATCH"; exit 1; }
python - "$t0" "$t1" "$t2" "$mib" "$kind" "$bs" <<'PY'
import sys
t0, t1, t2, mib = float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
gb = mib * 1048576 / 1e9
print("%s %d MiB @ %s: bwt %.3f s, inverse_bwt %.3f s (process start and device init included), round trip %.2f GB/s, verified"
      % (sys.argv[5], mib, sys.argv[6], t1 - t0, t2 - t1, gb / (t2 - t0)))
PY
