"""Origin-pointer blocks (the container of the reference's bzip2 comparison tool, reference
util/bzip2_bwt_utils.cpp:137-253) inverted on the GPU, against the CPU restatement in
oracle/origin_oracle.c and against the texts the blocks were made from.

PARITY UNPINNED: the reference holds no vectors for this format and its forward half needs bzip2's
sources, which are absent; inputs are made by the oracle's textbook rotation sort."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle
from bwt_amd import api, synth
from conftest import CORPUS, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def init():
    assert api.init(1) >= 1


@pytest.fixture(scope="module")
def orc():
    return oracle.origin()


def make_batch(orc, data: bytes, bs: int):
    cols, rows = [], []
    for off in range(0, len(data), bs):
        last, orig = orc.forward(data[off:off + bs])
        cols.append(last)
        rows.append(orig)
    return b"".join(cols), rows


def text_bytes(n):
    t = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    return (t * (n // len(t) + 1))[:n]


CASES = [
    ("banana", b"banana", 6),
    ("one byte", b"x", 1),
    ("one-byte blocks", b"hello world", 1),
    ("two-byte blocks", b"hello world", 2),
    ("run", b"a" * 5000, 5000),
    ("run in blocks", b"a" * 5000, 1300),
    ("short text blocks", text_bytes(10000), 100),
    ("just below the split bound", text_bytes(3000), 1279),
    ("at the split bound", text_bytes(3000), 1280),
    ("4 KiB text blocks + tail", text_bytes(4096 * 5 + 777), 4096),
    ("text 64 KiB blocks", text_bytes(3 * 65536 + 12345), 65536),
    ("whole file as one block", text_bytes(152089), 1 << 20),
    ("acgt 256 KiB blocks", synth.generate("acgt", 7, 0, 2 * 262144 + 999).tobytes(), 262144),
    ("random 64 KiB blocks", synth.generate("random255", 3, 0, 3 * 65536 + 1).tobytes(), 65536),
    # a block that is a whole number of periods: equal rotations, several LF cycles
    ("period-7, blocks of whole periods", bytes(97 + i % 7 for i in range(7 * 3000)), 7 * 1000),
    ("period-7, one block", bytes(97 + i % 7 for i in range(7 * 9000)), 7 * 9000),
    ("period-7, ragged blocks", bytes(97 + i % 7 for i in range(50000)), 16384),
    ("text twice", text_bytes(20000) * 2, 40000),
    ("text three times, short blocks", text_bytes(400) * 3, 1200),
]


@pytest.mark.parametrize("name,data,bs", CASES, ids=[c[0] for c in CASES])
def test_inverse_matches_oracle_and_text(orc, name, data, bs):
    last, rows = make_batch(orc, data, bs)
    want = b"".join(orc.inverse(last[off:off + bs], rows[off // bs]) for off in range(0, len(last), bs))
    assert want == data  # the oracle inverts its own transform
    got = api.inverse_origin_batch(last, bs, rows)
    assert got == want


@pytest.mark.parametrize("name,data,bs", CASES, ids=[c[0] for c in CASES])
def test_forward_matches_oracle(orc, name, data, bs):
    """Rotation sort on the GPU against the oracle's (last column and origin row of every block), and back
    through the GPU inverse."""
    want_last, want_rows = make_batch(orc, data, bs)
    last, rows = api.forward_origin_batch(data, bs)
    assert last == want_last
    assert list(rows) == list(want_rows)
    assert api.inverse_origin_batch(last, bs, rows) == data


def test_forward_many_blocks_and_long_equal_rotations(orc):
    """More blocks than one workgroup's list holds still tied when the depth reaches the block length (every
    rotation of a block of one byte value equals every other), and many short blocks."""
    for data, bs in ((b"z" * 300000, 100000), (bytes(65 + i % 3 for i in range(3 * 40000)), 3 * 20000),
                     (synth.generate("acgt", 11, 0, 300 * 2000 + 55).tobytes(), 2000),
                     (synth.generate("random255", 5, 0, 1 << 20).tobytes(), 1 << 18)):
        want_last, want_rows = make_batch(orc, data, bs)
        last, rows = api.forward_origin_batch(data, bs)
        assert last == want_last and list(rows) == list(want_rows), (len(data), bs)


def test_forward_container_round_trip(orc, tmp_path):
    """Text -> container (0xFF, row, last column per block) -> text, the container parsed as the reference's
    reader does."""
    data = text_bytes(200000)
    last, rows = api.forward_origin_batch(data, 65536)
    raw = api.build_origin_container(last, 65536, rows)
    assert len(raw) == len(data) + 4 * len(rows)
    cols, got_rows = api.parse_origin_container(raw, 65536)
    assert cols.tobytes() == last and list(got_rows) == list(rows)
    assert api.inverse_origin_batch(cols, 65536, got_rows) == data


def test_many_blocks(orc):
    data = synth.generate("acgt", 11, 0, 300 * 2000 + 55).tobytes()
    last, rows = make_batch(orc, data, 2000)
    assert api.inverse_origin_batch(last, 2000, rows) == data
    data = synth.generate("random255", 5, 0, 2100 * 1500).tobytes()  # more than 2048 blocks: the other split bound
    last, rows = make_batch(orc, data, 1500)
    assert api.inverse_origin_batch(last, 1500, rows) == data


def test_device_entry_point(orc):
    import torch

    data = text_bytes(5 * 20000 + 321)
    bs = 20000
    last, rows = make_batch(orc, data, bs)
    L = api.lib()
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(last), dtype=torch.uint8).to(dev)
    d_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
    d_out = torch.zeros(len(last), dtype=torch.uint8, device=dev)
    d_lens = torch.zeros(len(rows), dtype=torch.int32, device=dev)
    ws = L.bwtgpu_inverse_origin_workspace_bytes(0, len(last), bs)
    assert ws > 0
    d_ws = torch.empty(ws, dtype=torch.uint8, device=dev)
    small = torch.empty(256, dtype=torch.uint8, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    rc = L.bwtgpu_inverse_origin_batch_device(0, vp(d_in), len(last), bs, vp(d_rows), vp(d_out), vp(d_lens), vp(small), 256, st)
    assert rc == api.ERR_WORKSPACE
    api.check(L.bwtgpu_inverse_origin_batch_device(0, vp(d_in), len(last), bs, vp(d_rows), vp(d_out), vp(d_lens),
                                                   vp(d_ws), ws, st))
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().tobytes() == data
    assert d_lens.cpu().tolist() == [bs] * 5 + [321]


def test_row_outside_the_block_is_refused(orc):
    last, rows = make_batch(orc, b"banana" * 3, 6)
    with pytest.raises(api.BwtGpuError) as e:
        api.inverse_origin_batch(last, 6, [rows[0], 6, rows[2]])
    assert e.value.code == api.ERR_INVALID
    big = text_bytes(5000)
    last, rows = make_batch(orc, big, 5000)
    with pytest.raises(api.BwtGpuError):
        api.inverse_origin_batch(last, 5000, [5000])


def test_container_file(orc, tmp_path):
    data = text_bytes(70000)
    for bs in (512, 30000):
        box = orc.forward_buffer(data, bs)
        assert orc.inverse_buffer(box, bs) == data
        src, dst = tmp_path / "c.bin", tmp_path / "t.bin"
        src.write_bytes(box)
        assert api.bzip2_format_inverse_process_file(str(src), str(dst), bs) == 0
        assert dst.read_bytes() == data
    # a wrong marker ends the reading; the blocks before it are kept (as the reference's loop does)
    box = orc.forward_buffer(data[:3000], 1000)
    bad = box[:2 * 1004] + b"\x00" + box[2 * 1004 + 1:]
    src.write_bytes(bad)
    assert api.bzip2_format_inverse_process_file(str(src), str(dst), 1000) == 0
    assert dst.read_bytes() == data[:2000] == orc.inverse_buffer(bad, 1000)


def test_cpp_mirror():
    exe = os.path.join(ROOT, "build", "test_bzip2_format")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "-j8", "all"])
    r = subprocess.run([exe], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "All tests passed!" in r.stdout and "[FAIL]" not in r.stdout
