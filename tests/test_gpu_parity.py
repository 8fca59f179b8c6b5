"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
committed golden vectors.  Bit-exact is the bar: this is byte and index work.

Run on an MI355X with ``python -m pytest tests -m gpu``.
"""
import hashlib
import os
import random

import numpy as np
import pytest

import oracle
from bwt_amd import api, synth
from conftest import CORPUS, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    return oracle.oracle()


@pytest.fixture(scope="module", autouse=True)
def gpu():
    assert api.init(1) >= 1
    yield


@pytest.fixture(autouse=True, params=["auto", "block", "grid", "lds"])
def forward_path(request):
    """Every test runs four times: with the library choosing between the batch kernels and the
    device-wide forward path by the shape of the input, with either one forced ("block": the general
    batch kernels only), and with the batch kernels forced and the text-in-LDS kernel in front of them
    ("lds": the library puts it there by itself, but small test inputs would take the device-wide path)."""
    old = os.environ.get("BWTGPU_FORWARD_PATH")
    old_inv = os.environ.get("BWTGPU_INVERSE_PATH")
    old_lds = os.environ.get("BWTGPU_FORWARD_LDS")
    os.environ["BWTGPU_FORWARD_PATH"] = "block" if request.param == "lds" else request.param
    if request.param == "block":
        os.environ["BWTGPU_FORWARD_LDS"] = "0"
    else:
        os.environ.pop("BWTGPU_FORWARD_LDS", None)
    # The inverse routes ride along: the "auto" runs let the in-LDS inverse of small-alphabet blocks also
    # take batches of few blocks (the library gives those to the tiled table build otherwise, so a test's
    # single blocks would never reach it), the "block" runs keep everything on the table kernels, the
    # "grid" runs leave the choice to the library.
    if request.param == "auto":
        os.environ["BWTGPU_INVERSE_PATH"] = "packed"
    elif request.param == "block":
        os.environ["BWTGPU_INVERSE_PATH"] = "general"
    else:
        os.environ.pop("BWTGPU_INVERSE_PATH", None)
    yield request.param
    if old is None:
        del os.environ["BWTGPU_FORWARD_PATH"]
    else:
        os.environ["BWTGPU_FORWARD_PATH"] = old
    if old_lds is None:
        os.environ.pop("BWTGPU_FORWARD_LDS", None)
    else:
        os.environ["BWTGPU_FORWARD_LDS"] = old_lds
    if old_inv is None:
        os.environ.pop("BWTGPU_INVERSE_PATH", None)
    else:
        os.environ["BWTGPU_INVERSE_PATH"] = old_inv


def test_known_answers(known_answers):
    for v in known_answers:
        data = bytes.fromhex(v["input_hex"])
        fwd = api.bwt_forward(data, v["delimiter"])
        assert fwd.hex() == v["forward_hex"], v["name"]
        assert api.bwt_inverse(fwd, v["delimiter"]).hex() == v["inverse_hex"], v["name"]


def test_synthetic_vectors(synthetic_golden):
    for row in synthetic_golden:
        data = synth.generate(row["kind"], row["seed"], row["offset"], row["length"]).tobytes()
        fwd = api.bwt_forward(data, row["delimiter"])
        assert hashlib.sha256(fwd).hexdigest() == row["forward_sha256"], row
        assert api.bwt_inverse(fwd, row["delimiter"]) == data, row


def test_random_blocks_against_oracle(orc):
    rng = random.Random(20251111)
    for trial in range(120):
        n = rng.choice([1, 2, 3, 5, 31, 64, 65, 127, 128, 129, 500, 1023, 1024, 1025, 4097, 20000])
        sigma = rng.choice([1, 2, 3, 4, 16, 100, 255])
        data = bytes(rng.randrange(sigma) for _ in range(n))
        delim = 255 if trial % 7 else rng.randrange(256)  # every 7th: a delimiter that may occur in the data
        want = orc.forward(data, delim)
        got = api.bwt_forward(data, delim)
        assert got == want, (trial, n, sigma, delim)
        assert api.bwt_inverse(want, delim) == orc.inverse(want, delim), (trial, n, sigma, delim)


def test_batches_with_ragged_tail(orc):
    rng = np.random.default_rng(5)
    for total, bs in [(1, 1), (10, 1), (10, 3), (4096, 1024), (4097, 1024), (5000, 512), (100000, 4096),
                      (300000, 65536), (70000, 70000), (70001, 70000)]:
        data = rng.integers(0, 200, size=total, dtype=np.uint8).tobytes()
        want = orc.forward_buffer(data, bs, 4)
        delim = want[0]
        got = api.forward_batch(data, bs, delim)
        assert got == want[1:], (total, bs)
        assert api.inverse_batch(got, bs, delim) == data, (total, bs)


def test_inverse_errors_and_misuse(orc):
    with pytest.raises(api.BwtGpuError) as e:
        api.bwt_inverse(b"abcdef", ord("~"))
    assert e.value.code == api.ERR_NO_DELIMITER
    # several delimiters: the walk starts at the last one and stops at the first it reaches
    for blk in [b"~~aabb", b"~~~", b"a~b~c~", b"~"]:
        assert api.bwt_inverse(blk, ord("~")) == orc.inverse(blk, ord("~")), blk
    # a long block with exactly one delimiter that is nobody's transform: its LF permutation has several
    # cycles, and the text is what the cycle through the delimiter row spells (the segmented walk's chain
    # of segments then does not pass through all of them)
    rng = random.Random(77)
    for n, sigma in [(5000, 4), (70000, 3), (300000, 200)]:
        blk = bytearray(rng.randrange(1, 1 + sigma) for _ in range(n))
        blk[rng.randrange(n)] = 0
        assert api.bwt_inverse(bytes(blk), 0) == orc.inverse(bytes(blk), 0), (n, sigma)
    with pytest.raises(api.BwtGpuError) as e:
        api.forward_batch(b"abc", 0, 0)
    assert e.value.code == api.ERR_INVALID
    # a block size beyond the input is the input's own length (any value the reference accepts works)
    assert api.forward_batch(b"abc", api.MAX_BLOCK_SIZE + 1, 0) == orc.forward(b"abc", 0)
    assert api.inverse_batch(orc.forward(b"abc", 0), 1 << 40, 0) == b"abc"
    L = api.lib()
    assert not L.bwtgpu_pipeline_create(0, 0, api.MAX_BLOCK_SIZE + 1, 1, 1)
    assert "exceeds the supported maximum" in api.last_error()


def test_small_alphabet_inverse(orc):
    """Blocks over at most four text symbols (inverted inside LDS from two bits per row): every
    alphabet size, the delimiter below / between / above the symbols, block lengths around the
    size limits and around multiples of 64 rows, the delimiter row at either end of the block,
    a symbol that first appears late (the block falls back to the general kernels), batches
    that mix such blocks with ordinary ones, and hand-made blocks with several LF cycles."""
    rng = np.random.default_rng(4242)
    cases = []
    for n in [49150, 49151, 49152, 49215, 49216, 49217, 65536, 131071, 262143, 262144]:
        for symbols, delim in [(b"ACGT", 0), (b"ACGT", ord("D")), (b"ACGT", 255), (b"ab", ord("a") + 3), (b"x", 7),
                               (bytes([0, 1, 2]), 3), (bytes([1, 254, 255]), 0)]:
            if n > 70000 and symbols != b"ACGT":
                continue
            idx = rng.integers(0, len(symbols), size=n)
            cases.append((np.frombuffer(symbols, np.uint8)[idx].tobytes(), delim))
    for data, delim in cases:
        fwd = orc.forward(data, delim)
        assert api.bwt_inverse(fwd, delim) == data, (len(data), delim)
    # delimiter row first / last: text that sorts entirely above or below the delimiter
    for data, delim in [(b"C" * 50000, ord("A")), (b"C" * 50000, ord("T")), (b"AC" * 26000, ord("B"))]:
        assert api.bwt_inverse(orc.forward(data, delim), delim) == data
    # a fifth symbol beyond the sampled prefix
    late = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100000)].tobytes())
    late[77777] = ord("N")
    assert api.bwt_inverse(orc.forward(bytes(late), 0), 0) == bytes(late)
    # mixed batch: ACGT, random bytes, two symbols, ragged ACGT tail
    bs = 50000
    parts = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=bs)], rng.integers(1, 200, size=bs, dtype=np.uint8),
             np.frombuffer(b"AT", np.uint8)[rng.integers(0, 2, size=bs)], np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=12345)]]
    data = np.concatenate(parts).astype(np.uint8).tobytes()
    want = orc.forward_buffer(data, bs, 4)
    assert api.inverse_batch(want[1:], bs, want[0]) == data
    # a ragged last block that is itself long enough for the in-LDS path (its own cut map and sizes)
    rag = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=2 * 70000 + 60001)].tobytes()
    rag_t = orc.forward_buffer(rag, 70000, 4)
    assert api.inverse_batch(rag_t[1:], 70000, rag_t[0]) == rag
    # After a batch in which every block was inverted inside LDS the library launches the table kernels
    # as small striding grids; the next batch may need them after all: same bytes either way.
    dna = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=6 * bs)].tobytes()
    dna_t = orc.forward_buffer(dna, bs, 4)
    for _ in range(2):
        assert api.inverse_batch(dna_t[1:], bs, dna_t[0]) == dna
        assert api.inverse_batch(want[1:], bs, want[0]) == data
        assert api.inverse_batch(want[1:], bs, want[0]) == data
    # several LF cycles, small alphabets
    prng = random.Random(9)
    for n, sigma in [(50000, 1), (60000, 2), (70000, 4), (262145, 4)]:
        blk = bytearray(prng.randrange(1, 1 + sigma) for _ in range(n))
        blk[prng.randrange(n)] = 0
        assert api.bwt_inverse(bytes(blk), 0) == orc.inverse(bytes(blk), 0), (n, sigma)
    # no delimiter / two delimiters in a small-alphabet block
    blk = bytearray(b"ACGT" * 13000)
    with pytest.raises(api.BwtGpuError):
        api.bwt_inverse(bytes(blk), 0)
    blk[5] = 0
    blk[47000] = 0
    assert api.bwt_inverse(bytes(blk), 0) == orc.inverse(bytes(blk), 0)


def test_text_in_lds_forward(orc, forward_path):
    """Blocks the text-in-LDS forward kernel takes (at most four text symbols up to 256 KiB, any bytes
    up to 64 KiB): every alphabet size, the delimiter below / between / above the symbols, lengths
    around the limits and around multiples of 16 positions, runs of the smallest and the largest
    symbol before the delimiter (equal windows that are told apart by position), repeats longer than
    one window and longer than the kernel compares (the block then falls back), a fifth symbol the
    sample misses, skewed keys, batches that mix all of it with a ragged tail."""
    if forward_path in ("grid", "block"):
        pytest.skip("the text-in-LDS kernel is not on this route")
    rng = np.random.default_rng(31337)
    cases = []

    def draw(symbols, n):
        return np.frombuffer(bytes(symbols), np.uint8)[rng.integers(0, len(symbols), size=n)]

    for n in [2048, 2049, 2063, 3000, 4096, 5000, 8191, 12345, 16384, 16385, 16399, 16400, 16401, 20000, 28671, 28672, 65535, 65536, 65537, 131072,
              262143, 262144]:
        for symbols, delim in [(b"ACGT", 0), (b"ACGT", ord("D")), (b"ACGT", ord("H")), (b"ACGT", 255), (b"ab", ord("a") + 3),
                               (b"ab", 0), (b"x", 7), (b"x", 200), (bytes([0, 1, 2]), 3), (bytes([1, 254, 255]), 0),
                               (bytes([1, 254, 255]), 100)]:
            if n > 70000 and symbols not in (b"ACGT", b"ab"):
                continue
            cases.append((draw(symbols, n), delim))
    # bytes: uniform, few values, with the delimiter just below / above / between
    for n in [2048, 2500, 4097, 10000, 16384, 30000, 65535, 65536]:
        cases.append((rng.integers(0, 255, size=n, dtype=np.uint8), 255))
        cases.append((rng.integers(1, 256, size=n, dtype=np.uint8), 0))
        cases.append((rng.integers(0, 100, size=n, dtype=np.uint8) * 2, 101))  # the delimiter between the byte values
        cases.append((rng.integers(0, 6, size=n, dtype=np.uint8) + 10, 200))    # six values: bytes, skewed keys
        cases.append((rng.integers(0, 6, size=n, dtype=np.uint8) + 10, 12))     # the delimiter occurs in the text
    # runs before the delimiter: the last windows hold virtual symbols
    for symbols, delim in [(b"ACGT", 0), (b"ACGT", 255), (b"ACGT", ord("D")), (b"ab", 0), (b"ab", 255)]:
        for run_sym in (min(symbols), max(symbols)):
            for run in (1, 15, 16, 17, 40, 300):
                a = draw(symbols, 40000).copy()
                a[-run:] = run_sym
                cases.append((a, delim))
    for delim in (0, 255, 128):
        for run_val, run in [(1, 3), (1, 9), (254, 5), (127, 4), (129, 4)]:
            a = rng.integers(1, 255, size=30000, dtype=np.uint8)
            a[a == 128] = 7
            a[-run:] = run_val
            cases.append((a, delim))
    # repeats: two copies of a stretch (ties beyond one window), long enough to exceed what is compared
    for rep in (20, 40, 100, 200, 2000):
        a = draw(b"ACGT", 100000).copy()
        a[70000:70000 + rep] = a[1000:1000 + rep]
        a[99990 - rep:99990] = a[1000:1000 + rep]
        cases.append((a, 0))
        b = rng.integers(0, 250, size=50000, dtype=np.uint8)
        b[30000:30000 + rep] = b[100:100 + rep]
        cases.append((b, 255))
    # a fifth symbol, rare
    late = draw(b"ACGT", 100000).copy()
    late[77777] = ord("N")
    cases.append((late, 0))
    # skewed: 90 % one symbol; a long run inside
    sk = draw(b"ACGT", 200000).copy()
    sk[rng.random(200000) < 0.9] = ord("A")
    cases.append((sk, 0))
    runny = draw(b"ACGT", 150000).copy()
    runny[50000:90000] = ord("C")
    cases.append((runny, 255))
    # mildly skewed symbol frequencies: the fullest bucket target does not fit, a smaller one does (or none does and
    # the block is left to the general kernels)
    for probs in [(0.3, 0.3, 0.2, 0.2), (0.35, 0.3, 0.2, 0.15), (0.4, 0.3, 0.2, 0.1), (0.5, 0.25, 0.15, 0.1)]:
        for n in (65536, 262144):
            cases.append((np.frombuffer(b"ACGT", np.uint8)[rng.choice(4, size=n, p=probs)], 0))
    for arr, delim in cases:
        data = arr.astype(np.uint8).tobytes()
        want = orc.forward(data, delim)
        got = api.bwt_forward(data, delim)
        assert got == want, (len(data), delim, sorted(set(data))[:6])
    # batches: more blocks than the kernel has workgroups would be too slow for the oracle; a mixed
    # batch with a ragged tail and blocks the kernel gives up
    for bs, total in [(32768, 9 * 32768 + 17000), (65536, 5 * 65536 + 16384), (262144, 2 * 262144 + 100000)]:
        parts = []
        left = total
        i = 0
        while left > 0:
            n = min(bs, left)
            kind = i % 4
            if kind == 0:
                parts.append(draw(b"ACGT", n))
            elif kind == 1:
                parts.append(rng.integers(0, 250, size=n, dtype=np.uint8))
            elif kind == 2:
                parts.append(np.tile(np.frombuffer(b"abcdefg", np.uint8), n // 7 + 1)[:n])
            else:
                parts.append(draw(b"AT", n))
            left -= n
            i += 1
        data = np.concatenate(parts).astype(np.uint8).tobytes()
        want = orc.forward_buffer(data, bs, 8)
        assert api.forward_batch(data, bs, want[0]) == want[1:], (bs, total)


def test_lds_table_inverse(orc):
    """Blocks of 48 K to 64 K + 1 rows over any alphabet (their LF table fits LDS as 16-bit entries):
    lengths around the limits, alphabets from 2 to 255 values, skewed ones (symbols with fewer than
    64 rows: the first-column lookup steps over several), text, the delimiter below / between / above,
    the delimiter row first / last, hand-made blocks with several LF cycles, none or two delimiters,
    and more blocks than compute units with a ragged tail."""
    rng = np.random.default_rng(1616)
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    cases = []
    for n in [49151, 49152, 49153, 50000, 65535, 65536]:
        cases.append((rng.integers(0, 255, size=n, dtype=np.uint8).tobytes(), 255))
        cases.append((rng.integers(1, 256, size=n, dtype=np.uint8).tobytes(), 0))
        cases.append(((rng.integers(0, 100, size=n, dtype=np.uint8) * 2).astype(np.uint8).tobytes(), 101))
        cases.append((rng.integers(0, 7, size=n, dtype=np.uint8).tobytes(), 200))
        sk = rng.integers(0, 250, size=n, dtype=np.uint8)
        sk[rng.random(n) < 0.95] = 65  # 249 rare symbols, a few dozen rows each
        cases.append((sk.tobytes(), 255))
        cases.append((text[1000:1000 + n], 0))
    cases += [(b"C" * 50000, ord("A")), (b"C" * 50000, ord("T")), (b"AC" * 26000, ord("B")),
              ((b"abcdefg" * 9000)[:60000], 0), (bytes(range(1, 256)) * 250, 0)]
    for data, delim in cases:
        fwd = orc.forward(data, delim)
        assert api.bwt_inverse(fwd, delim) == data, (len(data), delim)
    prng = random.Random(16)
    for n, sigma in [(50000, 200), (60000, 5), (65537, 254), (65537, 30)]:
        blk = bytearray(prng.randrange(1, 1 + sigma) for _ in range(n))
        blk[prng.randrange(n)] = 0
        assert api.bwt_inverse(bytes(blk), 0) == orc.inverse(bytes(blk), 0), (n, sigma)
    blk = bytearray(rng.integers(1, 200, size=60000, dtype=np.uint8).tobytes())
    with pytest.raises(api.BwtGpuError):
        api.bwt_inverse(bytes(blk), 0)
    blk[5] = 0
    blk[47000] = 0
    assert api.bwt_inverse(bytes(blk), 0) == orc.inverse(bytes(blk), 0)
    # more blocks than compute units, every 5th one DNA (the two-bit kernel's), a ragged tail
    bs = 65536
    for nblocks, tail in [(300, 50001), (270, 20000)]:
        total = (nblocks - 1) * bs + tail
        data = rng.integers(0, 251, size=total, dtype=np.uint8)
        for b in range(3, nblocks - 1, 5):
            data[b * bs:(b + 1) * bs] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=bs)]
        data = data.tobytes()
        want = orc.forward_buffer(data, bs, 8)
        assert api.inverse_batch(want[1:], bs, want[0]) == data, (nblocks, tail)


def test_small_alphabet_many_blocks(orc):
    """More DNA blocks than compute units (the in-LDS inverse works through a list of blocks per
    workgroup), every 37th block ordinary bytes (left to the table kernels, which skip the finished
    blocks), a ragged last block that is still long enough for the in-LDS path, and a second batch
    with a different last block through the same library state."""
    rng = np.random.default_rng(777)
    bs = 49152
    for nblocks, tail in [(300, 49100), (270, 20000)]:
        total = (nblocks - 1) * bs + tail
        data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=total)].copy()
        for b in range(5, nblocks - 1, 37):
            data[b * bs:(b + 1) * bs] = rng.integers(1, 250, size=bs, dtype=np.uint8)
        data = data.tobytes()
        want = orc.forward_buffer(data, bs, 8)
        assert api.forward_batch(data, bs, want[0]) == want[1:]
        assert api.inverse_batch(want[1:], bs, want[0]) == data, (nblocks, tail)


def test_find_unused_byte():
    assert api.find_unused_byte(b"") == 0
    assert api.find_unused_byte(bytes(range(256))) == -1
    assert api.find_unused_byte(bytes(range(255)) * 1000) == 255
    rng = np.random.default_rng(1)
    for n in [1, 15, 16, 17, 1000, 123457]:
        a = rng.integers(0, 256, size=n, dtype=np.uint8)
        a[a == 77] = 78
        a[a == 3] = 4
        present = set(a.tolist())
        want = next((v for v in range(256) if v not in present), -1)
        assert api.find_unused_byte(a[1:]) == next((v for v in range(256) if v not in set(a[1:].tolist())), -1)
        assert api.find_unused_byte(a) == want


def test_corpus_files(corpus_golden, tmp_path):
    """Whole-file forward bytes equal the reference's (SHA-256), every corpus file."""
    for row in corpus_golden["files"]:
        if row["block_size"] < 4096:
            continue  # tiny-block sweep is covered by test_small_block_sweep on two files
        src = os.path.join(CORPUS, row["file"])
        f, b = str(tmp_path / "f.bwt"), str(tmp_path / "f.out")
        rc = api.bwt_forward_process_file(src, f, row["block_size"])
        assert rc == row["rc"], row
        if rc:
            continue
        out = open(f, "rb").read()
        assert len(out) == row["output_bytes"] and out[0] == row["delimiter"]
        assert hashlib.sha256(out).hexdigest() == row["sha256"], (row["file"], row["block_size"])
        assert api.bwt_inverse_process_file(f, b, row["block_size"]) == 0
        assert open(b, "rb").read() == open(src, "rb").read()


def test_small_block_sweep(corpus_golden, tmp_path):
    for row in corpus_golden["files"]:
        if row["block_size"] >= 4096 or row["file"] not in ("fields.c", "sum"):
            continue
        src = os.path.join(CORPUS, row["file"])
        f = str(tmp_path / "f.bwt")
        assert api.bwt_forward_process_file(src, f, row["block_size"]) == 0
        assert hashlib.sha256(open(f, "rb").read()).hexdigest() == row["sha256"], row


def test_alice29_128k_full_vector():
    raw = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    want = open(os.path.join(GOLDEN, "alice29_128k.bwt"), "rb").read()
    assert api.forward_batch(raw, 131072, want[0]) == want[1:]
    assert api.inverse_batch(want[1:], 131072, want[0]) == raw


def test_file_edge_cases(corpus_golden, tmp_path):
    for e in corpus_golden["edge"]:
        raw = bytes.fromhex(e["input_hex"])
        src, f, b = str(tmp_path / "e.in"), str(tmp_path / "e.bwt"), str(tmp_path / "e.out")
        open(src, "wb").write(raw)
        assert api.bwt_forward_process_file(src, f, e["block_size"]) == e["rc"]
        assert open(f, "rb").read().hex() == e["output_hex"], e["name"]
        assert api.bwt_inverse_process_file(f, b, e["block_size"]) == 0
        assert open(b, "rb").read() == raw
    empty = str(tmp_path / "empty")
    open(empty, "wb").close()
    assert api.bwt_inverse_process_file(empty, str(tmp_path / "x"), 1024) == 1  # reference src/inverse_bwt.cpp:78-82


@pytest.mark.parametrize("kind,bs", [("random255", 65536), ("acgt", 262144), ("period7", 262144),
                                     ("period1", 65536), ("period64", 131072), ("acgt", 1048576)])
def test_benchmark_shapes_against_oracle(orc, kind, bs):
    """A few blocks of each benchmark configuration, byte-for-byte against the oracle."""
    total = 2 * bs + 12345
    data = synth.generate(kind, synth.base_seed(2), 1000, total).tobytes()
    want = orc.forward_buffer(data, bs, 8)
    delim = want[0]
    got = api.forward_batch(data, bs, delim)
    assert got == want[1:]
    assert api.inverse_batch(got, bs, delim) == data


def test_synth_device_matches_host_generator():
    """The device generator bench.py fills HBM with writes exactly the bytes of synth.generate:
    every kind, unaligned offsets and lengths, spans that cross mixed-corpus segment borders."""
    import ctypes

    import torch

    L = api.lib()
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    dev = torch.device("cuda", 0)
    d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    spans = [(0, 1), (3, 4099), (1 << 20, 65536), ((1 << 20) - 7, 100003), (5 * (1 << 20) - 123457, 3 * (1 << 20) + 11),
             ((1 << 32) - 1000, 70001), (7 * (1 << 30) + 13, 262144 + 5)]
    for kind, code in synth.KINDS.items():
        for seed in (synth.base_seed(2), synth.base_seed(5), 12345):
            for offset, length in spans:
                want = synth.generate(kind, seed, offset, length, text=text)
                d_out = torch.full((length + 64,), 0xEE, dtype=torch.uint8, device=dev)
                api.check(L.bwtgpu_synth_device(0, code, seed, offset, ctypes.c_void_p(d_out.data_ptr() + 32), length,
                                                ctypes.c_void_p(d_text.data_ptr()), d_text.numel(), None))
                torch.cuda.synchronize()
                got = d_out.cpu().numpy()
                assert np.array_equal(got[32:32 + length], want), (kind, seed, offset, length)
                assert (got[:32] == 0xEE).all() and (got[32 + length:] == 0xEE).all(), (kind, offset, length)


def test_mixed_corpus_against_reference(mixed_golden):
    """BASELINE's mixed corpus at 1 MiB blocks: every segment kind fills a block (uniform bytes, ACGT,
    period-7 and tiled text at 1 MiB), ragged tails, blocks that straddle two segments -- per-block
    SHA-256 of the forward bytes equal the reference's, and the inverse restores the input."""
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    for row in mixed_golden:
        data = synth.generate("mixed", row["seed"], row["offset"], row["length"], text=text).tobytes()
        assert hashlib.sha256(data).hexdigest() == row["input_sha256"]
        bs, delim = row["block_size"], row["delimiter"]
        assert api.find_unused_byte(data) == delim
        got = api.forward_batch(data, bs, delim)
        for i, want in enumerate(row["block_forward_sha256"]):
            blk = got[i * (bs + 1):(i + 1) * (bs + 1)]
            assert hashlib.sha256(blk).hexdigest() == want, (row["offset"], i)
        assert api.inverse_batch(got, bs, delim) == data, row["offset"]


def test_skewed_distributions_and_many_blocks(orc):
    """Inputs that stress the round-0 bucket partition (one dominant key range that does not fit an
    LDS bucket, sparse key space, runs of equal bytes) and a batch whose block count is not a
    multiple of 8 (the inverse spreads blocks over the 8 XCDs)."""
    rng = np.random.default_rng(77)
    cases = []
    # 90 % one byte, rest uniform over 200 values: most suffixes share their first symbols
    a = rng.integers(0, 200, size=300000, dtype=np.uint8)
    a[rng.random(300000) < 0.9] = 65
    cases.append((a.tobytes(), 131072))
    # long runs of equal bytes separated by random text
    b = np.concatenate([np.full(30000, 7, np.uint8), rng.integers(0, 250, size=50000, dtype=np.uint8),
                        np.full(70000, 9, np.uint8), rng.integers(30, 40, size=60000, dtype=np.uint8)])
    cases.append((b.tobytes(), 262144))
    # two symbols only, block of 1 MiB: 17-symbol keys, two partition passes
    c = rng.integers(0, 2, size=1048576 + 999, dtype=np.uint8) + 100
    cases.append((c.tobytes(), 1048576))
    # 37 blocks of 8 KiB + ragged tail
    d = rng.integers(0, 255, size=37 * 8192 + 123, dtype=np.uint8)
    cases.append((d.tobytes(), 8192))
    # more blocks than compute units: the 256-thread batch kernel (fewer blocks of this size get
    # the 1024-thread one)
    f = rng.integers(0, 4, size=300 * 16384 + 77, dtype=np.uint8) + 65
    cases.append((f.tobytes(), 16384))
    # text-like: words from a small vocabulary
    vocab = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(50)]
    e = b" ".join(vocab[int(i)] for i in rng.integers(0, 50, size=60000))
    cases.append((e, 262144))
    # round-0 buckets too long for the LDS sort, taken through the fused rerank piece by piece: a run of one key longer
    # than a piece among random bytes (few blocks: 1024-thread workgroups; many blocks: 256-thread ones, blocks below the
    # size from which such blocks are put off), several long buckets side by side (words over few first letters),
    # a long run at the block's very start and at its very end
    g = rng.integers(0, 250, size=24000, dtype=np.uint8)
    g[3000:3000 + 9000] = 32
    cases.append((g.tobytes(), 24000))
    cases.append((np.tile(g[:7900], 300).tobytes(), 7900))
    h = np.concatenate([np.full(6000, 1, np.uint8), rng.integers(2, 250, size=12000, dtype=np.uint8), np.full(6000, 251, np.uint8)])
    cases.append((h.tobytes(), 24000))
    cases.append((np.tile(h[:7000], 280).tobytes() + h.tobytes(), 7000))
    few = [bytes([97 + int(rng.integers(0, 2)), 97 + int(rng.integers(0, 2))]) + bytes(rng.integers(97, 123, size=int(rng.integers(1, 7)), dtype=np.uint8))
           for _ in range(400)]
    w = b"".join(few[int(i)] for i in rng.integers(0, 400, size=40000))
    cases.append((w[:200000], 65536))
    cases.append((w[:300 * 7500], 7500))
    for data, bs in cases:
        want = orc.forward_buffer(data, bs, 8)
        delim = want[0]
        got = api.forward_batch(data, bs, delim)
        assert got == want[1:], (len(data), bs)
        assert api.inverse_batch(got, bs, delim) == data, (len(data), bs)


def test_repeats_inside_blocks(orc):
    """Blocks of otherwise random symbols with stretches copied elsewhere (several lengths from a few symbols to
    thousands, overlapping copies, a copy that runs into the block's end): the few suffixes inside the repeats stay
    tied for many doubling rounds, which the general kernels go through by looking ranks up instead of completing
    the rank array.  Sizes above what the in-LDS kernel holds, so the general kernels see the blocks first, and
    sizes it holds, so they see them after it has given up."""
    rng = np.random.default_rng(909)
    cases = []
    for symbols, n in [(np.frombuffer(b"ACGT", np.uint8), 300000), (np.frombuffer(b"ACGT", np.uint8), 200000),
                       (np.arange(1, 251, dtype=np.uint8), 70000), (np.arange(1, 251, dtype=np.uint8), 50000),
                       (np.frombuffer(b"ACGTN", np.uint8), 150000)]:
        a = symbols[rng.integers(0, len(symbols), size=n)].copy()
        at = 1000
        for length in (7, 13, 30, 61, 150, 400, 1100, 3000):
            src = int(rng.integers(0, n - 4000))
            dst = int(rng.integers(0, n - 4000))
            a[dst:dst + length] = a[src:src + length]
            at += length
        a[5000:5600] = a[4800:5400]          # a copy that overlaps its source: period 200 for a while
        a[n - 700:] = a[20000:20700]         # ... and one that ends with the block
        cases.append((a, n))
    batch = np.concatenate([c[0][:70000] for c in cases] * 60)  # many blocks: the 256-thread kernels
    cases.append((batch, 70000))
    for arr, bs in cases:
        data = arr.tobytes()
        want = orc.forward_buffer(data, bs, 8)
        delim, want = want[0], want[1:]
        got = api.forward_batch(data, bs, delim)
        assert got == want, (len(data), bs)
        assert api.inverse_batch(got, bs, delim) == data


def test_periodic_runs(orc):
    """Long groups whose positions form an arithmetic progression are ordered in one step by the
    batch kernel: both directions, several periods in one block, periods that only qualify in a
    later round, more than 64 such groups at once, a delimiter that continues the period."""
    rng = np.random.default_rng(2024)
    unit100 = rng.integers(0, 256, size=100, dtype=np.uint8)
    unit100[unit100 == 255] = 7
    cases = [
        (np.tile(np.frombuffer(b"abcdefg", np.uint8), 40000), 280000, None),
        (np.tile(unit100, 3000), 300000, None),                                  # 100 groups, period > first depth
        (np.concatenate([np.tile(np.frombuffer(b"ab", np.uint8), 60000), np.frombuffer(b"z", np.uint8),
                         rng.integers(97, 123, size=30000, dtype=np.uint8),
                         np.tile(np.frombuffer(b"xyz", np.uint8), 30000), np.zeros(40000, np.uint8),
                         np.frombuffer(b"q", np.uint8), np.full(50000, 200, np.uint8)]), 1 << 20, None),
        (np.tile(np.frombuffer(b"ba", np.uint8), 70000), 140000, None),
        (np.tile(np.frombuffer(b"abc", np.uint8), 300 * 16384 // 3 + 10), 16384, None),  # more blocks than CUs: 256-thread kernel
        (np.tile(rng.integers(0, 250, size=5000, dtype=np.uint8), 300 * 32768 // 5000 + 2)[:300 * 32768 + 77], 32768,
         None),                                                                   # ... with a long period: short groups
        (np.tile(rng.integers(0, 200, size=1000, dtype=np.uint8), 300), 300000, None),   # period 1000: longer than the lane search
        (np.tile(np.frombuffer(open(os.path.join(CORPUS, "alice29.txt"), "rb").read(), np.uint8), 8)[:(1 << 20) + 50000],
         1 << 20, None),                                                          # text tiled with a long period: many short groups
        (np.concatenate([np.tile(rng.integers(0, 200, size=3000, dtype=np.uint8), 40),
                         np.tile(rng.integers(0, 200, size=5000, dtype=np.uint8), 30)]), 270000, None),  # two long periods
        (np.tile(rng.integers(0, 200, size=30000, dtype=np.uint8), 20), 1 << 20, None),  # 20 copies: groups beyond the short-group limit
        (np.tile(np.frombuffer(open(os.path.join(CORPUS, "alice29.txt"), "rb").read()[:70000], np.uint8), 3)[:170000],
         1 << 20, None),                                                          # two and a bit copies: groups of 2 and 3
        (np.full(50000, ord("a"), np.uint8), 50000, ord("a")),                   # the delimiter continues the run
        (np.tile(np.frombuffer(b"abc", np.uint8), 30000), 90000, ord("a")),
    ]
    for arr, bs, delim in cases:
        data = arr.tobytes()
        if delim is None:
            want = orc.forward_buffer(data, bs, 8)
            delim, want = want[0], want[1:]
        else:
            want = orc.forward(data, delim)
        got = api.forward_batch(data, bs, delim)
        assert got == want, (len(data), bs, delim)
        if data.count(bytes([delim])) == 0:
            assert api.inverse_batch(got, bs, delim) == data


def test_small_periodic_blocks_in_large_batches(orc):
    """Batches of more blocks than compute units take the periodic-run step from 8 KiB blocks (batches of few
    blocks from 32 KiB): periods 1, 7 and 64 at block sizes on both sides of that bound, runs with exceptions
    (a scanned page's zeros), periodic blocks among ordinary ones, a ragged tail."""
    rng = np.random.default_rng(777)
    page = np.frombuffer(open(os.path.join(CORPUS, "ptt5"), "rb").read(), np.uint8)
    unit64 = rng.integers(0, 255, size=64, dtype=np.uint8)
    for bs in (8191, 8192, 8193, 12000, 16384, 20000):
        nblocks = 300
        n = nblocks * bs + 1234
        mixed = rng.integers(0, 200, size=n, dtype=np.uint8)
        for b in range(0, nblocks, 3):  # every third block periodic, the others random bytes
            mixed[b * bs:(b + 1) * bs] = np.resize(np.frombuffer(b"abcdefg", np.uint8), bs)
        for arr in (np.full(n, 97, np.uint8), np.resize(np.frombuffer(b"abcdefg", np.uint8), n), np.resize(unit64, n),
                    np.resize(page, n), mixed):
            data = arr.tobytes()
            want = orc.forward_buffer(data, bs, 8)
            delim, want = want[0], want[1:]
            got = api.forward_batch(data, bs, delim)
            assert got == want, (bs, int(arr[0]), int(arr[1]))
            assert api.inverse_batch(got, bs, delim) == data


def test_whole_block_periods(orc):
    """Blocks whose text is periodic throughout are sorted through their last two periods only:
    periods from 1 to a third of the block, lengths that are no multiple of the period, delimiters
    below, between and above the text symbols, the period broken at the first or last byte (then the
    block is sorted as a whole), a prefix with a shorter period than the block, a delimiter that
    occurs in the text, and batches of such blocks."""
    rng = np.random.default_rng(4242)
    cases = []

    def unit(p, lo=1, hi=250):
        while True:
            u = rng.integers(lo, hi, size=p, dtype=np.uint8)
            # the unit must not itself be periodic with a shorter period (then that one is found)
            if all(not np.array_equal(u, np.roll(u, k)) for k in range(1, p) if p % k == 0):
                return u

    for p, length in [(1, 40000), (2, 33001), (3, 65537), (7, 262144), (7, 32768), (64, 131072), (100, 50001),
                      (255, 40000), (256, 40000), (257, 99999), (1000, 70000), (5000, 45000), (11000, 33000),
                      (10922, 32768), (10923, 32768), (152089 // 4, 300000)]:
        u = unit(p)
        data = np.tile(u, length // p + 2)[3:3 + length]
        for delim in (0, 255, int(u.min()) + 1 if int(u.min()) + 1 not in set(u.tolist()) else 251):
            if delim not in set(u.tolist()):
                cases.append((data, length, delim))
    u = unit(7)
    base = np.tile(u, 6000)[:40000]
    broken_first, broken_last, broken_mid = base.copy(), base.copy(), base.copy()
    broken_first[0] ^= 1
    broken_last[-1] ^= 1
    broken_mid[23456] ^= 1
    cases += [(broken_first, 40000, 0), (broken_last, 40000, 255), (broken_mid, 40000, 0)]
    # the first 16 bytes repeat after 7 although the block's period is 3000
    long_unit = np.concatenate([np.tile(u, 5)[:30], unit(2970)])
    cases.append((np.tile(long_unit, 20)[:50000], 50000, 0))
    # delimiter inside the text of a periodic block (API misuse): suffix order ties on the delimiter
    cases.append((np.tile(u, 6000)[:40000], 40000, int(u[2])))
    # batches: every block periodic (period 7 does not divide the block size, so the phase changes), ragged tail
    cases.append((np.tile(u, 40000)[:5 * 40000 + 12345], 40000, 0))
    cases.append((np.tile(unit(300), 3000)[:300 * 32768 // 4 + 99], 32768, 255))
    for data, bs, delim in cases:
        raw = data.tobytes()
        want = b"".join(orc.forward(raw[i:i + bs], delim) for i in range(0, len(raw), bs))
        got = api.forward_batch(raw, bs, delim)
        assert got == want, (len(raw), bs, delim)
        if raw.count(bytes([delim])) == 0:
            assert api.inverse_batch(got, bs, delim) == raw


def test_classifying_launch_and_both_lists(orc, forward_path):
    """The launch that looks for whole-block periods before anything is sorted (the library uses it for
    batches of blocks of at least 512 KiB on the 256-thread kernels, which no small test batch reaches):
    the threshold lowered and the 1024-thread variant ruled out, a batch of periodic and ordinary blocks
    with a ragged tail goes through it and through both of the lists it fills; and the mixed corpus'
    vectors the same way."""
    if forward_path != "block":
        pytest.skip("one route: the general batch kernels behind the classifying launch")
    rng = np.random.default_rng(99)
    old = {k: os.environ.get(k) for k in ("BWTGPU_CLASSIFY_MIN_BLOCK", "BWTGPU_NARROW")}
    os.environ["BWTGPU_CLASSIFY_MIN_BLOCK"] = "32768"
    os.environ["BWTGPU_NARROW"] = "1"
    try:
        bs = 40000
        unit7 = np.frombuffer(b"abcdefg", np.uint8)
        unit300 = rng.integers(1, 250, size=300, dtype=np.uint8)
        parts = [np.tile(unit7, bs // 7 + 1)[:bs], rng.integers(0, 250, size=bs, dtype=np.uint8),
                 np.tile(unit300, bs // 300 + 1)[:bs], np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=bs)],
                 np.full(bs, 65, np.uint8), rng.integers(97, 123, size=bs, dtype=np.uint8),
                 np.tile(unit7, 4000)[:17001]]
        data = np.concatenate(parts).astype(np.uint8).tobytes()
        want = orc.forward_buffer(data, bs, 8)
        got = api.forward_batch(data, bs, want[0])
        assert got == want[1:]
        assert api.inverse_batch(got, bs, want[0]) == data
        text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
        row = load_mixed()[0]
        data = synth.generate("mixed", row["seed"], row["offset"], row["length"], text=text).tobytes()
        got = api.forward_batch(data, row["block_size"], row["delimiter"])
        for i, sha in enumerate(row["block_forward_sha256"]):
            blk = got[i * (row["block_size"] + 1):(i + 1) * (row["block_size"] + 1)]
            assert hashlib.sha256(blk).hexdigest() == sha, i
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def load_mixed():
    import json

    with open(os.path.join(GOLDEN, "mixed.json")) as f:
        return json.load(f)


def test_device_wide_path_driven_from_the_host(orc, forward_path):
    """The device-wide forward path runs as one resident kernel by default; driven from the host (a launch
    per step, the count read back every round: what a device without cooperative launches falls back to) it
    gives the same bytes."""
    if forward_path != "grid":
        pytest.skip("one route")
    old = os.environ.get("BWTGPU_GRID_SYNC")
    os.environ["BWTGPU_GRID_SYNC"] = "host"
    try:
        rng = np.random.default_rng(8)
        text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
        ptt5 = open(os.path.join(CORPUS, "ptt5"), "rb").read()
        for data, bs in [(text, 131072), (text, 1 << 20), (ptt5[:200000], 65536), (rng.integers(0, 255, size=300000, dtype=np.uint8).tobytes(), 100000),
                         (synth.generate("acgt", 5, 0, 3 * 262144 + 77).tobytes(), 262144), ((b"abcdefg" * 30000)[:200001], 70000)]:
            want = orc.forward_buffer(data, bs, 8)
            assert api.forward_batch(data, bs, want[0]) == want[1:], (len(data), bs)
    finally:
        if old is None:
            os.environ.pop("BWTGPU_GRID_SYNC", None)
        else:
            os.environ["BWTGPU_GRID_SYNC"] = old


def test_blocks_above_two_mib(orc, forward_path):
    """Blocks too long for the one-workgroup kernels: device-wide forward path, 64-bit inverse
    table words, coarser segments.  Against the oracle at 3 and 5 MiB, by round trip at 48 MiB."""
    if forward_path != "auto":
        pytest.skip("long blocks have one route")
    cases = [("random255", 3 << 20, (3 << 20) + 4567), ("acgt", 3 << 20, (3 << 20) + 4567),
             ("period7", 2500000, 2500000), ("acgt", 5 << 20, 2 * (5 << 20) + 99)]
    for kind, bs, total in cases:
        data = synth.generate(kind, synth.base_seed(3), 11, total).tobytes()
        want = orc.forward_buffer(data, bs, 8)
        delim = want[0]
        got = api.forward_batch(data, bs, delim)
        assert got == want[1:], (kind, bs)
        assert api.inverse_batch(got, bs, delim) == data, (kind, bs)
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    data = (text * (48 * (1 << 20) // len(text) + 1))[:48 << 20]  # tiled text: long repeats
    delim = api.find_unused_byte(data)
    got = api.forward_batch(data, len(data), delim)
    assert len(got) == len(data) + 1 and got.count(bytes([delim])) == 1
    assert sorted(np.bincount(np.frombuffer(got, np.uint8), minlength=256).tolist()) == \
        sorted((np.bincount(np.frombuffer(data, np.uint8), minlength=256) + np.eye(256, dtype=np.int64)[delim]).astype(np.int64).tolist())
    assert api.inverse_batch(got, len(data), delim) == data


def test_slab_pipeline_entry_points(orc):
    """The pipeline calls the C++ host layer is built on, driven through ctypes: upload + byte
    presence, submit from the uploaded bytes, submit with its own upload, inverse, slot reuse."""
    import ctypes

    L = api.lib()
    rng = np.random.default_rng(11)
    bs = 4096
    data = rng.integers(0, 200, size=10 * bs + 321, dtype=np.uint8)
    want = orc.forward_buffer(data.tobytes(), bs, 4)
    fwd = L.bwtgpu_pipeline_create(0, 0, bs, 16, 2)
    inv = L.bwtgpu_pipeline_create(0, 1, bs, 16, 2)
    assert fwd and inv
    try:
        assert L.bwtgpu_pipeline_input_capacity(fwd) == 16 * bs
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        for slot, staged in [(0, True), (1, False), (0, False)]:
            ctypes.memmove(L.bwtgpu_pipeline_input(fwd, slot), data.ctypes.data, data.size)
            if staged:
                mask = (ctypes.c_uint32 * 8)()
                assert L.bwtgpu_pipeline_upload_presence(fwd, slot, data.size, mask) == 0
                present = set(data.tolist())
                assert [v for v in range(256) if (mask[v >> 5] >> (v & 31)) & 1] == sorted(present)
                assert L.bwtgpu_lowest_clear_bit(mask) == want[0]
            assert L.bwtgpu_pipeline_submit(fwd, slot, data.size, want[0]) == 0
            assert L.bwtgpu_pipeline_collect(fwd, slot, ctypes.byref(out), ctypes.byref(n)) == 0
            got = ctypes.string_at(out.value, n.value)
            assert got == want[1:], (slot, staged)
        # submit and collect in one call (the caller waits: the library may take a route that blocks)
        ctypes.memmove(L.bwtgpu_pipeline_input(fwd, 1), data.ctypes.data, data.size)
        assert L.bwtgpu_pipeline_run(fwd, 1, data.size, want[0], ctypes.byref(out), ctypes.byref(n)) == 0
        assert ctypes.string_at(out.value, n.value) == want[1:]
        ctypes.memmove(L.bwtgpu_pipeline_input(inv, 0), got, len(got))
        assert L.bwtgpu_pipeline_submit(inv, 0, len(got), want[0]) == 0
        assert L.bwtgpu_pipeline_submit(inv, 0, len(got), want[0]) != 0  # not collected yet
        assert L.bwtgpu_pipeline_collect(inv, 0, ctypes.byref(out), ctypes.byref(n)) == 0
        assert ctypes.string_at(out.value, n.value) == data.tobytes()
    finally:
        L.bwtgpu_pipeline_destroy(fwd)
        L.bwtgpu_pipeline_destroy(inv)


def test_resident_slabs_entry_points(orc):
    """A file of several slabs read and uploaded once: slabs staged into device memory with their byte
    values folded into one mask, then transformed from there -- against the oracle, slab by slab, with the
    slots reused and a ragged last slab; and the misuse the calls refuse."""
    import ctypes

    L = api.lib()
    rng = np.random.default_rng(23)
    bs, slab_blocks = 4096, 8
    cap = bs * slab_blocks
    data = rng.integers(1, 180, size=5 * cap - 777, dtype=np.uint8)  # five slabs, the last one short
    delim, want = 0, orc.forward_buffer(data.tobytes(), bs, 4)
    assert want[0] == delim
    p = L.bwtgpu_pipeline_create(0, 0, bs, slab_blocks, 2)
    q = L.bwtgpu_pipeline_create(0, 1, bs, slab_blocks, 1)
    assert p and q
    try:
        mask = (ctypes.c_uint32 * 8)()
        assert L.bwtgpu_pipeline_presence(p, mask) != 0            # nothing reserved yet
        assert L.bwtgpu_pipeline_reserve(q, 4) != 0                # an inverse pipeline
        assert L.bwtgpu_pipeline_reserve(p, 0) != 0
        for attempt in range(2):                                   # the second time the reservation is reused
            assert L.bwtgpu_pipeline_reserve(p, 5) == 0
            assert L.bwtgpu_pipeline_stage(p, 0, 5, 10) != 0       # slab index out of range
            for k in range(5):
                slot = k % 2
                assert L.bwtgpu_pipeline_stage_wait(p, slot) == 0
                part = data[k * cap:(k + 1) * cap]
                ctypes.memmove(L.bwtgpu_pipeline_input(p, slot), part.ctypes.data, part.size)
                assert L.bwtgpu_pipeline_stage(p, slot, k, part.size) == 0
            assert L.bwtgpu_pipeline_presence(p, mask) == 0
            assert [v for v in range(256) if (mask[v >> 5] >> (v & 31)) & 1] == sorted(set(data.tolist()))
            assert L.bwtgpu_lowest_clear_bit(mask) == delim
            out, n = ctypes.c_void_p(), ctypes.c_size_t()
            got = b""
            for k in range(0, 5, 2):                               # two slabs in flight
                for j in range(k, min(k + 2, 5)):
                    assert L.bwtgpu_pipeline_submit_resident(p, j % 2, j, delim) == 0
                for j in range(k, min(k + 2, 5)):
                    assert L.bwtgpu_pipeline_collect(p, j % 2, ctypes.byref(out), ctypes.byref(n)) == 0
                    got += ctypes.string_at(out.value, n.value)
            assert got == want[1:], attempt
    finally:
        L.bwtgpu_pipeline_destroy(p)
        L.bwtgpu_pipeline_destroy(q)


def test_concurrent_callers(orc):
    """The per-block functions are called from many worker threads in the reference
    (src/bwt.cpp:143-156): concurrent calls from several host threads give the same bytes."""
    import threading

    rng = np.random.default_rng(99)
    inputs = [rng.integers(0, 200, size=int(n), dtype=np.uint8).tobytes()
              for n in [100, 5000, 30000, 70000, 200000, 1000, 50000, 120000]]
    want = [orc.forward(d, 255) for d in inputs]
    got = [None] * len(inputs)
    back = [None] * len(inputs)

    def work(i):
        for _ in range(3):
            got[i] = api.bwt_forward(inputs[i], 255)
            back[i] = api.bwt_inverse(got[i], 255)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == want
    assert back == inputs
