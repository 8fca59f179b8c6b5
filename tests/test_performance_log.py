"""The log build/performance prints keeps the reference's grammar: a log captured on an MI355X
(tests/golden/performance_two_files.log: `build/performance <alice29.txt + sum> 2`) is read by this
repository's reader and -- where the reference checkout is present -- by the reference's own plot
script (reference util/plot_performance.py:52-110), which must return every (file, block size) row
with the same numbers and must not trip over the extra "GPU ..." lines."""
import os
import sys

import pytest

from conftest import GOLDEN
from perf_log import parse_performance_log

LOG = os.path.join(GOLDEN, "performance_two_files.log")
SIZES_KB = [0.5, 1.0, 4.0, 16.0, 64.0, 256.0, 1024.0]
WANT = sorted((f, s) for f in ("alice29.txt", "sum") for s in SIZES_KB)


def test_own_reader_finds_every_row():
    rows = parse_performance_log(open(LOG, encoding="utf-8").read())
    assert sorted((r["filename"], r["block_size_kb"]) for r in rows) == WANT
    for r in rows:
        assert r["roundtrip_time_ms"] == pytest.approx(r["forward_time_ms"] + r["inverse_time_ms"], rel=1e-3)
        assert r["gpu_roundtrip_gbps"] > 0 and r["gpus_used"] == 1


REF_UTIL = "/root/reference/util"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_UTIL, "plot_performance.py")),
                    reason="the reference checkout is not on this machine")
def test_reference_plot_script_parses_the_log():
    sys.path.insert(0, REF_UTIL)
    try:
        import matplotlib

        matplotlib.use("Agg")
        import plot_performance
    finally:
        sys.path.remove(REF_UTIL)
    p = plot_performance.BenchmarkParser()
    p.parse_file(LOG, "C++")
    assert sorted((r["filename"], r["block_size_kb"]) for r in p.data) == WANT
    mine = {(r["filename"], r["block_size_kb"]): r for r in parse_performance_log(open(LOG, encoding="utf-8").read())}
    for r in p.data:
        m = mine[(r["filename"], r["block_size_kb"])]
        for k in ("forward_time_ms", "inverse_time_ms", "roundtrip_time_ms", "throughput_mb_s", "compression_ratio"):
            assert r[k] == pytest.approx(m[k]), (k, r, m)
    agg = plot_performance.aggregate_data(p.data)
    assert agg
