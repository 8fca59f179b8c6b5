"""Pins the CPU oracle (oracle/bwt_oracle.c) against the reference's golden vectors.

The expected values in tests/golden/ were produced by the reference itself
(tests/golden/make_golden.py); the ten known answers are the ones the reference's
own unit test asserts (reference tests/test_small.cpp:80-92).  When the compiled
reference (oracle/_ref) is present it is also compared directly on random inputs.
"""
import hashlib
import os
import random

import pytest

import oracle
from bwt_amd import synth
from conftest import CORPUS, GOLDEN


@pytest.fixture(scope="module")
def orc():
    return oracle.oracle()


def test_known_answers(orc, known_answers):
    assert len(known_answers) >= 10
    for v in known_answers:
        data = bytes.fromhex(v["input_hex"])
        fwd = orc.forward(data, v["delimiter"])
        assert fwd.hex() == v["forward_hex"], v["name"]
        assert len(fwd) == len(data) + 1
        assert orc.inverse(fwd, v["delimiter"]).hex() == v["inverse_hex"], v["name"]


def test_inverse_without_delimiter_is_an_error(orc):
    with pytest.raises(ValueError):
        orc.inverse(b"abc", ord("~"))


def test_find_unique_byte(orc):
    assert orc.find_unique_byte(b"") == 0
    assert orc.find_unique_byte(bytes(range(256))) == -1
    assert orc.find_unique_byte(bytes(range(255))) == 255
    assert orc.find_unique_byte(bytes([0, 1, 2, 4])) == 3


def test_corpus_forward_sha256(orc, corpus_golden):
    """Whole-file forward bytes (reference src/bwt.cpp:159-237 layout) for every corpus file."""
    cache = {}
    checked = 0
    for row in corpus_golden["files"]:
        if row["block_size"] < 4096 and row["input_bytes"] > 200000:
            continue  # many tiny blocks: slow through ctypes, covered by the smaller files
        raw = cache.setdefault(row["file"], open(os.path.join(CORPUS, row["file"]), "rb").read())
        if row["rc"] != 0:
            with pytest.raises(ValueError):
                orc.forward_buffer(raw, row["block_size"], 4)
            continue
        out = orc.forward_buffer(raw, row["block_size"], 4)
        assert out[0] == row["delimiter"]
        nblocks = (len(raw) + row["block_size"] - 1) // row["block_size"]
        assert len(out) == len(raw) + 1 + nblocks == row["output_bytes"]  # reference tests/test_medium.cpp:60-72
        assert hashlib.sha256(out).hexdigest() == row["sha256"], (row["file"], row["block_size"])
        if row["block_size"] in (4096, 131072):
            assert orc.inverse_buffer(out, row["block_size"]) == raw
        checked += 1
    assert checked >= 50


def test_alice29_128k_full_vector(orc):
    raw = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    want = open(os.path.join(GOLDEN, "alice29_128k.bwt"), "rb").read()
    assert orc.forward_buffer(raw, 131072, 2) == want
    assert orc.inverse_buffer(want, 131072) == raw


def test_file_edge_cases(orc, corpus_golden):
    for e in corpus_golden["edge"]:
        raw = bytes.fromhex(e["input_hex"])
        out = orc.forward_buffer(raw, e["block_size"], 2)
        assert out.hex() == e["output_hex"], e["name"]
        assert orc.inverse_buffer(out, e["block_size"]) == raw


def test_empty_bwt_file_is_an_error(orc):
    with pytest.raises(ValueError):
        orc.inverse_buffer(b"", 1024)  # reference src/inverse_bwt.cpp:78-82


def test_synthetic_vectors(orc, synthetic_golden):
    for row in synthetic_golden:
        data = synth.generate(row["kind"], row["seed"], row["offset"], row["length"]).tobytes()
        fwd = orc.forward(data, row["delimiter"])
        assert hashlib.sha256(fwd).hexdigest() == row["forward_sha256"], row
        if "forward_hex" in row:
            assert fwd.hex() == row["forward_hex"]
        assert orc.inverse(fwd, row["delimiter"]) == data


def test_mixed_corpus_vectors(orc, mixed_golden):
    """The mixed corpus at 1 MiB blocks: the oracle reproduces the reference's per-block bytes."""
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    row = mixed_golden[0]
    data = synth.generate("mixed", row["seed"], row["offset"], row["length"], text=text).tobytes()
    assert hashlib.sha256(data).hexdigest() == row["input_sha256"]
    bs = row["block_size"]
    out = orc.forward_buffer(data, bs, 8)
    assert out[0] == row["delimiter"]
    for i, want in enumerate(row["block_forward_sha256"]):
        assert hashlib.sha256(out[1 + i * (bs + 1):1 + (i + 1) * (bs + 1)]).hexdigest() == want, i


def test_file_drivers(orc, tmp_path):
    src = os.path.join(CORPUS, "fields.c")
    raw = open(src, "rb").read()
    f, b = str(tmp_path / "f.bwt"), str(tmp_path / "f.out")
    assert orc.forward_file(src, f, 4096, 3) == 0
    assert open(f, "rb").read() == orc.forward_buffer(raw, 4096)
    assert orc.inverse_file(f, b, 4096) == 0
    assert open(b, "rb").read() == raw
    assert orc.forward_file(os.path.join(CORPUS, "kennedy.xls"), f, 4096, 3) == 1
    assert orc.forward_file(str(tmp_path / "missing"), f, 4096, 1) == 1


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_equals_compiled_reference(orc, tmp_path):
    ref = oracle.reference()
    rng = random.Random(1234)
    for trial in range(200):
        n = rng.choice([0, 1, 2, 3, 7, 64, 100, 1000, 5000])
        sigma = rng.choice([1, 2, 4, 26, 255])
        data = bytes(rng.randrange(sigma) for _ in range(n))
        delim = rng.choice([255, sigma, 0]) if trial % 5 else rng.randrange(256)  # sometimes a misuse delimiter
        fwd = ref.forward(data, delim)
        assert orc.forward(data, delim) == fwd
        assert orc.inverse(fwd, delim) == ref.inverse(fwd, delim)
    src = os.path.join(CORPUS, "sum")
    for bs in (700, 65536):
        a, b = str(tmp_path / "a"), str(tmp_path / "b")
        assert ref.forward_file(src, a, bs) == 0 and orc.forward_file(src, b, bs, 4) == 0
        assert open(a, "rb").read() == open(b, "rb").read()
        assert ref.find_unique_char(src) == orc.find_unique_byte(open(src, "rb").read())


def test_origin_pointer_oracle():
    """The CPU restatement of the origin-pointer format (reference util/bzip2_bwt_utils.cpp:137-253;
    parity unpinned: the reference holds no vectors for it): the textbook answer for "banana", the
    rank-doubling forward equal to the rotation-comparing one, and the inverse undoing both."""
    import random

    o = oracle.origin()
    assert o.forward(b"banana") == (b"nnbaaa", 3) == o.forward_slow(b"banana")
    assert o.inverse(b"nnbaaa", 3) == b"banana"
    rng = random.Random(5)
    for _ in range(200):
        n = rng.randint(1, 300)
        s = bytes(rng.randrange(97, 97 + rng.choice([1, 2, 4, 26])) for _ in range(n))
        if rng.random() < 0.3:  # a shorter string repeated: equal rotations
            p = rng.randint(1, max(1, n // 2))
            s = (s[:p] * (n // p + 1))[:n]
        a = o.forward(s)
        assert a == o.forward_slow(s)
        assert o.inverse(*a) == s
    data = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()[:50000]
    for bs in (1, 999, 50000):
        box = o.forward_buffer(data, bs)
        assert len(box) == len(data) + 4 * ((len(data) + bs - 1) // bs)
        assert o.inverse_buffer(box, bs) == data
    assert o.inverse_buffer(b"\x00abc", 3) == b""  # wrong marker: nothing is read
    with pytest.raises(ValueError):
        o.inverse(b"abc", 3)
