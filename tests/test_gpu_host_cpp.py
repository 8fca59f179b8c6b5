"""The C++ host layer (reference-compatible API + CLIs) on the GPU: runs the C++ test
programs and checks the command-line tools byte-for-byte against the golden vectors."""
import hashlib
import os
import subprocess

import pytest

from conftest import CORPUS, ROOT

pytestmark = pytest.mark.gpu
BUILD = os.path.join(ROOT, "build")


@pytest.fixture(scope="module", autouse=True)
def binaries():
    need = ["bwt", "inverse_bwt", "test_small", "test_medium", "performance"]
    if not all(os.path.exists(os.path.join(BUILD, n)) for n in need):
        subprocess.check_call(["make", "-C", ROOT, "-j8", "all"])
    yield


def run(args, **kw):
    return subprocess.run(args, cwd=ROOT, capture_output=True, text=True, timeout=600, **kw)


def test_cpp_known_answers():
    r = run([os.path.join(BUILD, "test_small")])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All tests passed!" in r.stdout and "[FAIL]" not in r.stdout


def test_cpp_file_round_trips():
    r = run([os.path.join(BUILD, "test_medium")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("[PASS]") == 55  # 11 files x 5 block sizes (kennedy.xls: expected rejection)


def test_cli_matches_golden(corpus_golden, tmp_path):
    rows = [r for r in corpus_golden["files"] if r["file"] in ("alice29.txt", "ptt5", "sum", "kennedy.xls")
            and r["block_size"] in (4096, 131072, 1048576)]
    assert rows
    for row in rows:
        src = os.path.join(CORPUS, row["file"])
        f, b = str(tmp_path / "f.bwt"), str(tmp_path / "f.out")
        r = run([os.path.join(BUILD, "bwt"), src, f, str(row["block_size"])])
        assert r.returncode == row["rc"], (row, r.stderr)
        if row["rc"]:
            assert "Cannot find a unique delimiter" in r.stderr
            continue
        assert hashlib.sha256(open(f, "rb").read()).hexdigest() == row["sha256"], row
        r = run([os.path.join(BUILD, "inverse_bwt"), f, b, str(row["block_size"])])
        assert r.returncode == 0, r.stderr
        assert open(b, "rb").read() == open(src, "rb").read()


def test_cli_contract(tmp_path):
    bwt, inv = os.path.join(BUILD, "bwt"), os.path.join(BUILD, "inverse_bwt")
    for exe in (bwt, inv):
        r = run([exe])
        assert r.returncode == 1 and "Usage:" in r.stderr
        r = run([exe, "a", "b", "0"])
        assert r.returncode == 1 and "Block size must be greater than 0" in r.stderr
        r = run([exe, str(tmp_path / "missing"), str(tmp_path / "o")])
        assert r.returncode == 1 and r.stderr.count("Could not open input file") == 1
        # the forward tool then reports what the reference reports (src/bwt.cpp:163-166): its
        # delimiter search could not read the file
        assert ("Cannot find a unique delimiter" in r.stderr) == (exe == bwt)
    # default block size is 128 bytes (reference src/bwt.cpp:249, src/inverse_bwt.cpp:111)
    src = os.path.join(CORPUS, "grammar.lsp")
    f, b = str(tmp_path / "g.bwt"), str(tmp_path / "g.out")
    assert run([bwt, src, f]).returncode == 0
    n = os.path.getsize(src)
    assert os.path.getsize(f) == n + 1 + (n + 127) // 128
    assert run([inv, f, b]).returncode == 0
    assert open(b, "rb").read() == open(src, "rb").read()
    # any block size the reference accepts works: beyond the file it means "one block"
    f2, b2 = str(tmp_path / "g2.bwt"), str(tmp_path / "g2.out")
    assert run([bwt, src, f, "1048576"]).returncode == 0
    assert run([bwt, src, f2, "5000000000"]).returncode == 0
    assert open(f2, "rb").read() == open(f, "rb").read()
    assert run([inv, f2, b2, "5000000000"]).returncode == 0
    assert open(b2, "rb").read() == open(src, "rb").read()
    # BWT_DEVICES asks for more GPUs than there may be: capped to what the library bound, same bytes
    f3 = str(tmp_path / "g3.bwt")
    r = subprocess.run([bwt, src, f3, "1048576"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, BWT_DEVICES="4"))
    assert r.returncode == 0, r.stderr
    assert open(f3, "rb").read() == open(f, "rb").read()
    # empty file -> 1-byte output (delimiter 0x00); inverting it gives an empty file
    e, ef, eb = str(tmp_path / "e"), str(tmp_path / "e.bwt"), str(tmp_path / "e.out")
    open(e, "wb").close()
    assert run([bwt, e, ef, "1024"]).returncode == 0
    assert open(ef, "rb").read() == b"\x00"
    assert run([inv, ef, eb, "1024"]).returncode == 0
    assert os.path.getsize(eb) == 0
    # an empty .bwt file is an error (reference src/inverse_bwt.cpp:78-82)
    r = run([inv, e, eb, "1024"])
    assert r.returncode == 1 and "Empty input file" in r.stderr


def test_two_slab_pipeline_depth(tmp_path):
    """A file larger than one slab exercises slot reuse and in-order collection."""
    import numpy as np

    from bwt_amd import synth
    import oracle

    data = synth.generate("acgt", 5, 0, 40 * (1 << 20) + 12345)  # 512-byte blocks: 16 MiB slabs -> 3 slabs
    src, f, b = str(tmp_path / "big"), str(tmp_path / "big.bwt"), str(tmp_path / "big.out")
    data.tofile(src)
    assert run([os.path.join(BUILD, "bwt"), src, f, "512"]).returncode == 0
    out = np.fromfile(f, dtype=np.uint8)
    assert out.size == data.size + 1 + (data.size + 511) // 512
    # spot-check blocks from every slab against the oracle
    orc = oracle.oracle()
    for blk in (0, 1, 32767, 32768, 40000, 70000, (data.size - 1) // 512):
        want = orc.forward(data[blk * 512:(blk + 1) * 512].tobytes(), int(out[0]))
        got = out[1 + blk * 513:1 + blk * 513 + len(want)].tobytes()
        assert got == want, blk
    assert run([os.path.join(BUILD, "inverse_bwt"), f, b, "512"]).returncode == 0
    assert np.array_equal(np.fromfile(b, dtype=np.uint8), data)
    # The same file driven by the calling thread alone (multi-slab files normally get a reading thread,
    # one submitting thread per GPU and a writing thread) and with more GPUs asked for than there are:
    # identical bytes both ways.
    # ... and by the two-pass route (delimiter pass, then reader / submitter / writer threads), which files too
    # long to stay on the devices and inputs that are not regular files take
    for env in ({"BWT_HOST_SERIAL": "1"}, {"BWT_DEVICES": "8"}, {"BWT_RESIDENT_MIB": "0"},
                {"BWT_RESIDENT_MIB": "0", "BWT_HOST_SERIAL": "1"}, {"BWT_SLAB_MIB": "1"}):
        f2, b2 = str(tmp_path / "big2.bwt"), str(tmp_path / "big2.out")
        e = dict(os.environ, **env)
        assert subprocess.run([os.path.join(BUILD, "bwt"), src, f2, "512"], cwd=ROOT, env=e, timeout=600).returncode == 0
        assert open(f2, "rb").read() == open(f, "rb").read()
        assert subprocess.run([os.path.join(BUILD, "inverse_bwt"), f2, b2, "512"], cwd=ROOT, env=e, timeout=600).returncode == 0
        assert np.array_equal(np.fromfile(b2, dtype=np.uint8), data)


def test_threaded_runner_reports_errors(tmp_path):
    """A multi-slab inverse whose second slab holds a block without delimiter: exit code 1 and a message,
    no hang (reader, submitters and writer all stop)."""
    import numpy as np

    from bwt_amd import synth

    data = synth.generate("acgt", 9, 0, 20 * (1 << 20))
    src, f, b = str(tmp_path / "d"), str(tmp_path / "d.bwt"), str(tmp_path / "d.out")
    data.tofile(src)
    assert run([os.path.join(BUILD, "bwt"), src, f, "512"]).returncode == 0
    raw = np.fromfile(f, dtype=np.uint8)
    delim = int(raw[0])
    blk = 33000  # in the second 16 MiB slab
    seg = raw[1 + blk * 513:1 + (blk + 1) * 513]
    seg[seg == delim] = ord("A")
    raw.tofile(f)
    r = run([os.path.join(BUILD, "inverse_bwt"), f, b, "512"])
    assert r.returncode == 1 and "delimiter" in r.stderr
    # the output was set to its full length up front: what is left is only what was written
    got = np.fromfile(b, dtype=np.uint8)
    assert got.size < data.size and np.array_equal(got, data[:got.size])
    # a device node as the output: nothing to reserve, plain writes
    assert run([os.path.join(BUILD, "inverse_bwt"), str(tmp_path / "d.good.bwt"), "/dev/null", "512"]).returncode == 1  # no such input
    assert run([os.path.join(BUILD, "bwt"), src, "/dev/null", "512"]).returncode == 0


def _sample_blocks_equal_oracle(data, out, bs, delim, blocks):
    import oracle

    orc = oracle.oracle()
    for blk in blocks:
        want = orc.forward(data[blk * bs:(blk + 1) * bs].tobytes(), delim)
        got = out[1 + blk * (bs + 1):1 + blk * (bs + 1) + len(want)].tobytes()
        assert got == want, (bs, blk)


# Block sizes on either side of every threshold at which the library changes kernels or workspace
# layout (one-wave blocks up to 1024 bytes, walks split from 1280 / 4096 rows, tiled table build
# from 16384 rows, device-wide forward path from 20000 bytes), with block counts just below 128 and
# just below 2048, where the batch geometry changes; the last block is ragged.
@pytest.mark.parametrize("bs", [1025, 1279, 1280, 2048, 3000, 4095, 16383, 19999, 20000])
@pytest.mark.parametrize("nblocks", [127, 2047])
def test_cli_round_trip_around_geometry_thresholds(tmp_path, bs, nblocks):
    import numpy as np

    from bwt_amd import synth

    total = (nblocks - 1) * bs + bs // 3 + 1
    data = synth.generate("random255", 77 + bs, 0, total)
    data[data > 200] = 65  # a skewed byte: longer doubling, and the delimiter is 201
    src, f, b = str(tmp_path / "in"), str(tmp_path / "f.bwt"), str(tmp_path / "out")
    data.tofile(src)
    r = run([os.path.join(BUILD, "bwt"), src, f, str(bs)])
    assert r.returncode == 0, r.stderr
    out = np.fromfile(f, dtype=np.uint8)
    assert out.size == total + 1 + nblocks and out[0] == 201
    _sample_blocks_equal_oracle(data, out, bs, 201, [0, 1, nblocks // 2, nblocks - 2, nblocks - 1])
    r = run([os.path.join(BUILD, "inverse_bwt"), f, b, str(bs)])
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(b, dtype=np.uint8), data)


# DNA files through the command-line tools: the inverse takes blocks over at most four symbols
# inside LDS when a block has 49152 .. 262145 rows (on either side the table kernels run), also for
# a slab pipeline whose last slab is short and for a file that mixes DNA with other bytes.
@pytest.mark.parametrize("bs", [49150, 49151, 65536, 262144, 262145])
def test_cli_dna_files(tmp_path, bs):
    import numpy as np

    from bwt_amd import synth

    nblocks = 40 if bs < 100000 else 18
    total = (nblocks - 1) * bs + bs // 2 + 3
    data = synth.generate("acgt", 1000 + bs, 0, total)
    if bs == 65536:  # a stretch of ordinary bytes in the middle: those blocks go to the table kernels
        data[5 * bs + 100:8 * bs - 100] = synth.generate("random255", 3, 0, 3 * bs - 200)
        data[data == 0] = 1
    src, f, b = str(tmp_path / "in"), str(tmp_path / "f.bwt"), str(tmp_path / "out")
    data.tofile(src)
    r = run([os.path.join(BUILD, "bwt"), src, f, str(bs)])
    assert r.returncode == 0, r.stderr
    out = np.fromfile(f, dtype=np.uint8)
    assert out.size == total + 1 + nblocks and out[0] == 0
    _sample_blocks_equal_oracle(data, out, bs, 0, [0, 6, nblocks - 1])
    for env in ({}, {"BWTGPU_INVERSE_PATH": "packed"}, {"BWTGPU_INVERSE_PATH": "general"}):
        r = subprocess.run([os.path.join(BUILD, "inverse_bwt"), f, b, str(bs)], cwd=ROOT, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.fromfile(b, dtype=np.uint8), data), env


@pytest.mark.parametrize("bs,slab", [(1279, 2048), (1280, 2048), (2048, 2048), (4095, 2048), (16383, 256),
                                     (19999, 256), (20000, 256)])
def test_pipeline_sized_for_a_full_slab_takes_any_shorter_batch(bs, slab):
    """A pipeline is created once for a full slab and then handed shorter batches (the last slab
    of a file, or a smaller file that reuses a cached pipeline): batches on either side of every
    block-count threshold must fit its workspace and give the oracle's bytes."""
    import ctypes

    import numpy as np

    import oracle
    from bwt_amd import api, synth

    L = api.lib()
    assert api.init(1) >= 1
    orc = oracle.oracle()
    data = synth.generate("random255", 5 + bs, 0, slab * bs)
    data[data > 180] = 66
    delim = 181
    fwd = L.bwtgpu_pipeline_create(0, 0, bs, slab, 1)
    inv = L.bwtgpu_pipeline_create(0, 1, bs, slab, 1)
    assert fwd and inv, api.last_error()
    try:
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        counts = sorted({1, 31, 32, 33, 127, 128, 129, slab - 1, slab} | ({821, 822, 1465, 2047} if slab >= 2048 else set()))
        for nb in counts:
            for ragged in (0, bs // 2):
                total = nb * bs - ragged
                if total <= 0:
                    continue
                ctypes.memmove(L.bwtgpu_pipeline_input(fwd, 0), data.ctypes.data, total)
                assert L.bwtgpu_pipeline_submit(fwd, 0, total, delim) == 0, (nb, api.last_error())
                assert L.bwtgpu_pipeline_collect(fwd, 0, ctypes.byref(out), ctypes.byref(n)) == 0, (nb, api.last_error())
                got = np.frombuffer(ctypes.string_at(out.value, n.value), dtype=np.uint8)
                assert got.size == total + nb
                for blk in {0, nb // 2, nb - 1}:
                    want = orc.forward(data[blk * bs:min((blk + 1) * bs, total)].tobytes(), delim)
                    assert got[blk * (bs + 1):blk * (bs + 1) + len(want)].tobytes() == want, (bs, nb, blk)
                ctypes.memmove(L.bwtgpu_pipeline_input(inv, 0), got.ctypes.data, got.size)
                assert L.bwtgpu_pipeline_submit(inv, 0, got.size, delim) == 0, (nb, api.last_error())
                assert L.bwtgpu_pipeline_collect(inv, 0, ctypes.byref(out), ctypes.byref(n)) == 0, (nb, api.last_error())
                assert n.value == total
                assert ctypes.string_at(out.value, total) == data[:total].tobytes(), (bs, nb)
    finally:
        L.bwtgpu_pipeline_destroy(fwd)
        L.bwtgpu_pipeline_destroy(inv)


def test_performance_tool(tmp_path):
    """build/performance on a two-file directory: the reference's log grammar (Test / Block Size /
    Mean / Throughput lines, aggregate table; reference tests/performance.cpp:131-259) plus the
    device-resident lines and the second aggregate table."""
    from perf_log import parse_performance_log

    d = tmp_path / "two"
    d.mkdir()
    for name in ("alice29.txt", "sum"):
        (d / name).write_bytes(open(os.path.join(CORPUS, name), "rb").read())
    r = run([os.path.join(BUILD, "performance"), str(d), "2"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = parse_performance_log(r.stdout)
    sizes = [0.5, 1.0, 4.0, 16.0, 64.0, 256.0, 1024.0]  # KB, as the reference's plot script keys them
    assert sorted((x["filename"], x["block_size_kb"]) for x in rows) == sorted(
        (f, s) for f in ("alice29.txt", "sum") for s in sizes)
    for x in rows:
        assert x["throughput_mb_s"] > 0 and x["forward_time_ms"] > 0 and x["inverse_time_ms"] > 0
        assert x["gpu_roundtrip_gbps"] > 0 and 0 < x["gpu_pct_hbm"] < 100 and x["gpus_used"] >= 1
    assert "Aggregate Statistics by Block Size (Entire Dataset)" in r.stdout
    assert "GPU Aggregate by Block Size" in r.stdout
    assert "FAILED" not in r.stdout
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):  # kept so that a captured log can be committed as the parser fixture
        open(os.path.join(out, "performance_two_files.log"), "w").write(r.stdout)


def test_compare_tool():
    """build/compare_bwt_performance (the shape of the reference's tests/compare_bwt_performance.cpp: five
    trials at 64 / 128 / 256 KiB, forward, inverse, round trip): nine SUMMARY lines in the reference's
    grammar (util/performance_comparison.cpp:297-312), every trial's two round trips verified by the tool."""
    r = run([os.path.join(BUILD, "compare_bwt_performance"), os.path.join(CORPUS, "alice29.txt")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l.split("|") for l in r.stdout.splitlines() if l.startswith("SUMMARY|")]
    assert [l[2] for l in lines] == ["forward", "inverse", "roundtrip"] * 3
    for l in lines:
        assert len(l) == 8 and l[1] == "alice29.txt" and l[6] in ("your_bwt", "bzip2")
        assert float(l[3]) > 0 and float(l[4]) > 0 and float(l[5]) > 0 and float(l[7]) >= 0
    assert r.stdout.count("(successful: 5)") == 3
    assert run([os.path.join(BUILD, "compare_bwt_performance"), "/nonexistent"]).returncode == 1


def test_origin_extractor_tools(tmp_path):
    """build/bzip2_bwt_extractor and build/bzip2_inverse_bwt_extractor (the command lines of the reference's
    tests/bzip2_bwt_extractor.cpp and tests/bzip2_inverse_bwt_extractor.cpp): file -> container -> file."""
    import oracle

    fwd, inv = os.path.join(BUILD, "bzip2_bwt_extractor"), os.path.join(BUILD, "bzip2_inverse_bwt_extractor")
    src = os.path.join(CORPUS, "alice29.txt")
    data = open(src, "rb").read()
    orc = oracle.origin()
    for args, bs in (([], 65536), (["20000"], 20000)):
        f, b = str(tmp_path / "c.bz2bwt"), str(tmp_path / "c.out")
        assert run([fwd, src, f] + args).returncode == 0
        raw = open(f, "rb").read()
        nb = (len(data) + bs - 1) // bs
        assert len(raw) == len(data) + 4 * nb
        last, orig = orc.forward(data[:bs])  # the first block against the rotation-sort oracle
        assert raw[0] == 0xFF and int.from_bytes(raw[1:4], "big") == orig and raw[4:4 + bs] == last
        assert run([inv, f, b] + args).returncode == 0
        assert open(b, "rb").read() == data
    assert run([fwd, src]).returncode == 1
    assert run([fwd, src, str(tmp_path / "x"), "0"]).returncode == 1
