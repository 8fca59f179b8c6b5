for k in ${KINDS_1M:-random255 acgt period7}; do
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --kind $k --block-size 1048576 --bytes-per-gpu 1073741824 --no-cpu-baseline --no-e2e --no-other-workloads 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$k', d['value'], d['forward_ms'], d['inverse_ms'], d['forward_phase_share'])"
done
