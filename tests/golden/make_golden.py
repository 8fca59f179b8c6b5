#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Run where the reference checkout is present (it is compiled, unmodified, into
oracle/_ref/libbwt_ref.so by ``make -C oracle``):

    python tests/golden/make_golden.py

Every expected output below is produced by the REFERENCE ITSELF (through
oracle/ref_shim.cpp), never by this repository's code.  The files written are
data only:

* known_answers.json   -- the ten (input, delimiter, forward) triples the reference's
                          own unit test asserts (tests/test_small.cpp:80-92), re-checked
                          against the compiled reference, plus edge / API-misuse cases.
* corpus_sha256.json   -- SHA-256 and size of the reference's whole-file forward output
                          for every Canterbury file x block size; rc for rejected files.
* alice29_128k.bwt     -- the reference's forward output for alice29.txt at 128 KiB blocks
                          (benchmark configuration 1), in full.
* synthetic.json       -- reference forward outputs (SHA-256; in full for short blocks) for
                          seeded synthetic blocks of every generator kind.
* mixed.json           -- the mixed corpus (1 MiB segments cycling random255 / ACGT / period-7 /
                          tiled alice29.txt) at 1 MiB blocks: per-block SHA-256 of the reference's
                          forward output for spans that give every segment kind a block of its own,
                          a ragged tail, and a span whose blocks straddle two segments.
                          ``python tests/golden/make_golden.py --only mixed`` regenerates just it.
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from bwt_amd import synth  # noqa: E402

CORPUS = os.path.join(ROOT, "data", "canterbury_corpus")
BLOCK_SIZES = [512, 1024, 4096, 16384, 65536, 131072, 262144, 1048576]

# (name, input, delimiter, expected forward) -- reference tests/test_small.cpp:80-92
KNOWN = [
    ("basic string", b"banana", "~", b"bnn~aaa"),
    ("hello", b"hello", "~", b"h~ello"),
    ("mississippi", b"mississippi", "~", b"ssmp~pissiii"),
    ("empty string", b"", "~", b"~"),
    ("single character", b"a", "~", b"~a"),
    ("repeated characters", b"aaaa", "~", b"~aaaa"),
    ("special characters", b"a!b@c#", "~", b"acb~!@#"),
    ("custom delimiter", b"test", "$", b"ttes$"),
    ("longer string", b"the quick brown fox jumps over the lazy dog", "~",
     b"kynxeserl i hhv ottu c uwd rfm ebp ~qjooozag"),
    ("string with newlines", b"line1\nline2\nline3", "~", b"12eeennnlll~\n\niii3"),
]

# Edge cases the reference has no assertion for; expected values come from running it.
EXTRA = [
    ("misuse: delimiter inside input", b"ab~ab", ord("~")),
    ("misuse: input is only delimiters", b"~~", ord("~")),
    ("unsigned byte order", bytes([0x61, 0x00, 0x62, 0xFF, 0x61]), 0x01),
    ("high delimiter", b"\x00\x01\x02\x00\x01", 0xFF),
    ("low delimiter", b"zyxzyx", 0x00),
    ("two symbols", b"abababababababab", ord("~")),
    ("run then change", b"a" * 33 + b"b", ord("~")),
    ("all 255 other values", bytes(range(1, 256)), 0x00),
    ("misuse: all 256 values in input", bytes(range(256)) * 2, 0x80),
]


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


# (offset, length) spans of the mixed stream, all at 1 MiB blocks: four whole segments + a ragged
# tail from a segment boundary (blocks = segments: random255, ACGT, period-7, text, random255 tail);
# the same from another place in the stream (other text phase); blocks that straddle two segments.
MIXED_SPANS = [(0, 4 * (1 << 20) + 300000), (37 * (1 << 20), 4 * (1 << 20) + 77), ((1 << 20) + 500000, 3 * (1 << 20) + 1234)]


def mixed_rows(ref):
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    seed = synth.base_seed(5)
    bs = 1 << 20
    rows = []
    for offset, length in MIXED_SPANS:
        data = synth.generate("mixed", seed, offset, length, text=text).tobytes()
        present = set(data)
        delim = next(v for v in range(256) if v not in present)
        blocks = []
        for b0 in range(0, length, bs):
            blk = data[b0:b0 + bs]
            fwd = ref.forward(blk, delim)
            assert ref.inverse(fwd, delim) == blk
            blocks.append(sha(fwd))
        rows.append({"kind": "mixed", "seed": seed, "offset": offset, "length": length, "block_size": bs,
                     "delimiter": delim, "input_sha256": sha(data), "block_forward_sha256": blocks})
    return rows


def main() -> None:
    ref = oracle.reference()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "mixed":
        with open(os.path.join(HERE, "mixed.json"), "w") as f:
            json.dump(mixed_rows(ref), f, indent=1)
        print("wrote mixed.json")
        return

    known = []
    for name, data, delim, expect in KNOWN:
        got = ref.forward(data, ord(delim))
        assert got == expect, (name, got, expect)
        assert ref.inverse(got, ord(delim)) == data
        known.append({"name": name, "input_hex": data.hex(), "delimiter": ord(delim),
                      "forward_hex": expect.hex(), "inverse_hex": data.hex(),
                      "source": "tests/test_small.cpp:80-92"})
    for name, data, delim in EXTRA:
        fwd = ref.forward(data, delim)
        inv = ref.inverse(fwd, delim)
        known.append({"name": name, "input_hex": data.hex(), "delimiter": delim,
                      "forward_hex": fwd.hex(), "inverse_hex": inv.hex(),
                      "source": "reference run"})
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)

    corpus = []
    with tempfile.TemporaryDirectory() as tmp:
        for fn in sorted(os.listdir(CORPUS)):
            src = os.path.join(CORPUS, fn)
            raw = open(src, "rb").read()
            for bs in BLOCK_SIZES:
                dst = os.path.join(tmp, "f.bwt")
                back = os.path.join(tmp, "f.out")
                if os.path.exists(dst):
                    os.remove(dst)
                rc = ref.forward_file(src, dst, bs)
                row = {"file": fn, "block_size": bs, "input_bytes": len(raw), "rc": rc}
                if rc == 0:
                    out = open(dst, "rb").read()
                    assert ref.inverse_file(dst, back, bs) == 0
                    assert open(back, "rb").read() == raw
                    row.update({"delimiter": out[0], "output_bytes": len(out), "sha256": sha(out)})
                    if fn == "alice29.txt" and bs == 131072:
                        with open(os.path.join(HERE, "alice29_128k.bwt"), "wb") as g:
                            g.write(out)
                corpus.append(row)
        # whole-file edge cases
        edge = []
        for name, raw, bs in [("empty file", b"", 1024),
                              ("one byte", b"x", 1024),
                              ("exact multiple", bytes(range(97, 123)) * 80 , 1040),
                              ("two full blocks", synth.generate("random255", 7, 0, 2048).tobytes(), 1024),
                              ("multiple plus one", synth.generate("acgt", 7, 0, 2049).tobytes(), 1024),
                              ("multiple minus one", synth.generate("acgt", 7, 0, 2047).tobytes(), 1024),
                              ("block size one", b"hello world", 1),
                              ("block larger than file", b"hello world", 4096)]:
            src = os.path.join(tmp, "e.in")
            dst = os.path.join(tmp, "e.bwt")
            open(src, "wb").write(raw)
            rc = ref.forward_file(src, dst, bs)
            out = open(dst, "rb").read()
            edge.append({"name": name, "block_size": bs, "input_hex": raw.hex(), "rc": rc,
                         "output_hex": out.hex()})
    with open(os.path.join(HERE, "corpus_sha256.json"), "w") as f:
        json.dump({"files": corpus, "edge": edge}, f, indent=1)

    synthetic = []
    for kind in ["random255", "acgt", "period7", "period1", "period64"]:
        for length in [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4096, 65536, 131072, 262144]:
            seed = synth.base_seed(length)
            data = synth.generate(kind, seed, 3, length).tobytes()
            delim = 0xFF if kind == "random255" or kind == "period64" else 0x00
            fwd = ref.forward(data, delim)
            row = {"kind": kind, "seed": seed, "offset": 3, "length": length, "delimiter": delim,
                   "forward_sha256": sha(fwd)}
            if length <= 257:
                row["forward_hex"] = fwd.hex()
            synthetic.append(row)
    with open(os.path.join(HERE, "synthetic.json"), "w") as f:
        json.dump(synthetic, f, indent=1)
    with open(os.path.join(HERE, "mixed.json"), "w") as f:
        json.dump(mixed_rows(ref), f, indent=1)
    print("wrote", len(known), "known answers,", len(corpus), "corpus rows,", len(synthetic), "synthetic rows")


if __name__ == "__main__":
    main()
