"""Deterministic synthetic inputs for the BWT codec benchmarks and tests (host side).

Every byte of a stream depends only on ``(kind, seed, absolute index)``, so any
slice can be produced independently: a rank generates exactly its shard, and
the device generator (``bwtgpu_synth`` in ``csrc/bwtgpu.hip``) produces the same
bytes as this numpy version (checked by ``tests/test_gpu_parity.py``).

Random words are the SplitMix64 output sequence: word ``k`` of a stream is
``mix(seed + (k + 1) * GOLDEN)``.

Kinds
-----
``random255``  bytes uniform over 0x00..0xFE (0xFF never occurs, so the on-disk
               format's "one unused byte value" delimiter always exists; the
               reference rejects inputs that use all 256 values,
               reference ``src/bwt.cpp:163-166``).  Four bytes per word:
               ``((w >> 16 t) & 0xFFFF) * 255 >> 16``.
``acgt``       ``"ACGT"[(w >> 2 t) & 3]``, 32 symbols per word.
``period7``    ``"abcdefg"[i % 7]``.
``period1``    ``"a"``.
``period64``   a 64-byte ``random255`` unit (stream positions 0..63) repeated.
``text``       a caller-supplied sample (alice29.txt) tiled: byte ``i`` is ``text[i % len(text)]``.
``mixed``      1 MiB segments cycling random255 / acgt / period7 / text, the text
               segments tiling a caller-supplied sample (alice29.txt).
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
KINDS = {"random255": 0, "acgt": 1, "period7": 2, "period1": 3, "period64": 4, "mixed": 5, "text": 6}
SEGMENT = 1 << 20


def base_seed(config_id: int) -> int:
    return 0x9E3779B97F4A7C15 ^ config_id


def _mix(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def _words(seed: int, first: int, count: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = np.arange(first + 1, first + 1 + count, dtype=np.uint64)
        return _mix(np.uint64(seed) + k * GOLDEN)


def _random255(seed: int, offset: int, length: int) -> np.ndarray:
    w0, w1 = offset // 4, (offset + length + 3) // 4
    w = _words(seed, w0, w1 - w0)
    lanes = np.empty((w.size, 4), dtype=np.uint64)
    for t in range(4):
        lanes[:, t] = (w >> np.uint64(16 * t)) & np.uint64(0xFFFF)
    b = ((lanes * np.uint64(255)) >> np.uint64(16)).astype(np.uint8).reshape(-1)
    s = offset - 4 * w0
    return b[s:s + length]


def _acgt(seed: int, offset: int, length: int) -> np.ndarray:
    w0, w1 = offset // 32, (offset + length + 31) // 32
    w = _words(seed, w0, w1 - w0)
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    idx = ((w[:, None] >> sh) & np.uint64(3)).astype(np.uint8).reshape(-1)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    s = offset - 32 * w0
    return lut[idx[s:s + length]]


def generate(kind: str, seed: int, offset: int, length: int, text: bytes | None = None) -> np.ndarray:
    """Bytes ``[offset, offset + length)`` of the stream ``(kind, seed)`` as a uint8 array."""
    if length == 0:
        return np.empty(0, dtype=np.uint8)
    if kind == "random255":
        return _random255(seed, offset, length)
    if kind == "acgt":
        return _acgt(seed, offset, length)
    if kind == "period7":
        i = np.arange(offset, offset + length, dtype=np.uint64)
        return (np.uint64(ord("a")) + i % np.uint64(7)).astype(np.uint8)
    if kind == "period1":
        return np.full(length, ord("a"), dtype=np.uint8)
    if kind == "period64":
        unit = _random255(seed, 0, 64)
        i = np.arange(offset, offset + length, dtype=np.uint64) % np.uint64(64)
        return unit[i.astype(np.int64)]
    if kind == "text":
        if not text:
            raise ValueError("kind 'text' needs a text sample")
        t = np.frombuffer(text, dtype=np.uint8)
        i = np.arange(offset, offset + length, dtype=np.uint64) % np.uint64(t.size)
        return t[i.astype(np.int64)]
    if kind == "mixed":
        if not text:
            raise ValueError("kind 'mixed' needs a text sample")
        t = np.frombuffer(text, dtype=np.uint8)
        out = np.empty(length, dtype=np.uint8)
        pos = offset
        while pos < offset + length:
            seg = pos // SEGMENT
            end = min((seg + 1) * SEGMENT, offset + length)
            n = end - pos
            which = seg % 4
            if which == 0:
                part = _random255(seed, pos, n)
            elif which == 1:
                part = _acgt(seed, pos, n)
            elif which == 2:
                part = generate("period7", seed, pos, n)
            else:
                i = np.arange(pos, end, dtype=np.uint64) % np.uint64(t.size)
                part = t[i.astype(np.int64)]
            out[pos - offset:end - offset] = part
            pos = end
        return out
    raise ValueError("unknown kind " + kind)
