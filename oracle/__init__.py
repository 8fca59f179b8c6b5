"""CPU checkers for the BWT block codec -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The shipped package ``bwt_amd`` never
does: its product path is the HIP library and fails loudly without it.

Two checkers live here (see ``oracle/Makefile``):

* ``liboracle.so``  -- ``bwt_oracle.c``, a plain-C restatement of the reference
  algorithm (prefix-doubling comparator sort, LF walk, file framing).
* ``_ref/libbwt_ref.so`` -- the unmodified reference compiled from its own
  sources (only buildable where the reference checkout is present; the built
  file travels with the tree).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libbwt_ref.so")

_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(verbose: bool = False) -> None:
    """Compile the checkers (``make -C oracle all``)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=out)


def _buf(data: bytes):
    return (ctypes.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")


class _Oracle:
    """ctypes view of liboracle.so (the C restatement)."""

    def __init__(self) -> None:
        if not os.path.exists(_ORACLE_SO):
            build()
        lib = ctypes.CDLL(_ORACLE_SO)
        lib.oracle_find_unique_byte.argtypes = [_u8p, ctypes.c_size_t]
        lib.oracle_find_unique_byte.restype = ctypes.c_int
        lib.oracle_bwt_forward.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_uint8, _u8p]
        lib.oracle_bwt_forward.restype = ctypes.c_int
        lib.oracle_bwt_inverse.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_uint8, _u8p]
        lib.oracle_bwt_inverse.restype = ctypes.c_long
        lib.oracle_forward_buffer.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_size_t, _u8p, ctypes.c_int]
        lib.oracle_forward_buffer.restype = ctypes.c_long
        lib.oracle_inverse_buffer.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_size_t, _u8p]
        lib.oracle_inverse_buffer.restype = ctypes.c_long
        lib.oracle_forward_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        lib.oracle_forward_file.restype = ctypes.c_int
        lib.oracle_inverse_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        lib.oracle_inverse_file.restype = ctypes.c_int
        self.lib = lib

    def find_unique_byte(self, data: bytes) -> int:
        return self.lib.oracle_find_unique_byte(_buf(data), len(data))

    def forward_rounds(self, data: bytes, delimiter: int):
        out = (ctypes.c_uint8 * (len(data) + 1))()
        rounds = self.lib.oracle_bwt_forward(_buf(data), len(data), delimiter, out)
        return bytes(out), rounds

    def forward(self, data: bytes, delimiter: int) -> bytes:
        return self.forward_rounds(data, delimiter)[0]

    def inverse(self, data: bytes, delimiter: int) -> bytes:
        out = (ctypes.c_uint8 * max(1, len(data)))()
        r = self.lib.oracle_bwt_inverse(_buf(data), len(data), delimiter, out)
        if r < 0:
            raise ValueError("block holds no delimiter")
        return bytes(out)[:r]

    def forward_buffer(self, data: bytes, block_size: int, nthreads: int = 1) -> bytes:
        nblocks = (len(data) + block_size - 1) // block_size
        out = (ctypes.c_uint8 * (len(data) + 1 + nblocks))()
        r = self.lib.oracle_forward_buffer(_buf(data), len(data), block_size, out, nthreads)
        if r < 0:
            raise ValueError("no unused byte value for the delimiter")
        return bytes(out)[:r]

    def inverse_buffer(self, data: bytes, block_size: int) -> bytes:
        out = (ctypes.c_uint8 * max(1, len(data)))()
        r = self.lib.oracle_inverse_buffer(_buf(data), len(data), block_size, out)
        if r < 0:
            raise ValueError("empty input" if r == -1 else "chunk without delimiter")
        return bytes(out)[:r]

    def forward_file(self, src: str, dst: str, block_size: int, nthreads: int = 1) -> int:
        return self.lib.oracle_forward_file(src.encode(), dst.encode(), block_size, nthreads)

    def inverse_file(self, src: str, dst: str, block_size: int) -> int:
        return self.lib.oracle_inverse_file(src.encode(), dst.encode(), block_size)


class _Reference:
    """ctypes view of _ref/libbwt_ref.so (the reference itself, via ref_shim.cpp)."""

    def __init__(self) -> None:
        lib = ctypes.CDLL(_REF_SO)
        lib.ref_bwt_forward.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_uint8, _u8p]
        lib.ref_bwt_forward.restype = ctypes.c_size_t
        lib.ref_bwt_inverse.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_uint8, _u8p]
        lib.ref_bwt_inverse.restype = ctypes.c_size_t
        lib.ref_forward_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        lib.ref_forward_file.restype = ctypes.c_int
        lib.ref_inverse_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        lib.ref_inverse_file.restype = ctypes.c_int
        lib.ref_find_unique_char.argtypes = [ctypes.c_char_p]
        lib.ref_find_unique_char.restype = ctypes.c_int
        self.lib = lib

    def forward(self, data: bytes, delimiter: int) -> bytes:
        out = (ctypes.c_uint8 * (len(data) + 1))()
        n = self.lib.ref_bwt_forward(_buf(data), len(data), delimiter, out)
        return bytes(out)[:n]

    def inverse(self, data: bytes, delimiter: int) -> bytes:
        if bytes([delimiter]) not in data:
            raise ValueError("the reference has undefined behaviour without a delimiter")
        out = (ctypes.c_uint8 * max(1, len(data)))()
        n = self.lib.ref_bwt_inverse(_buf(data), len(data), delimiter, out)
        return bytes(out)[:n]

    def forward_file(self, src: str, dst: str, block_size: int) -> int:
        return self.lib.ref_forward_file(src.encode(), dst.encode(), block_size)

    def inverse_file(self, src: str, dst: str, block_size: int) -> int:
        return self.lib.ref_inverse_file(src.encode(), dst.encode(), block_size)

    def find_unique_char(self, path: str) -> int:
        return self.lib.ref_find_unique_char(path.encode())


_oracle = None
_reference = None


def oracle() -> _Oracle:
    global _oracle
    if _oracle is None:
        _oracle = _Oracle()
    return _oracle


def have_reference() -> bool:
    return os.path.exists(_REF_SO)


def reference() -> _Reference:
    global _reference
    if _reference is None:
        if not have_reference():
            raise FileNotFoundError(_REF_SO + " (build it where the reference checkout exists: make -C oracle ref)")
        _reference = _Reference()
    return _reference


class _OriginOracle:
    """ctypes view of liborigin_oracle.so: the origin-pointer block format of the reference's bzip2
    comparison tool (reference util/bzip2_bwt_utils.cpp:137-253).  Parity unpinned: the reference
    holds no vectors for it and its forward half needs bzip2's sources, which are absent."""

    def __init__(self) -> None:
        so = os.path.join(_HERE, "liborigin_oracle.so")
        if not os.path.exists(so):
            build()
        lib = ctypes.CDLL(so)
        lib.oracle_origin_forward.argtypes = [_u8p, ctypes.c_size_t, _u8p]
        lib.oracle_origin_forward.restype = ctypes.c_long
        lib.oracle_origin_inverse.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_size_t, _u8p]
        lib.oracle_origin_inverse.restype = ctypes.c_int
        lib.oracle_origin_inverse_buffer.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_size_t, _u8p]
        lib.oracle_origin_inverse_buffer.restype = ctypes.c_long
        self.lib = lib

    def forward_slow(self, data: bytes):
        """(last column, origin row) by comparing whole rotations: small blocks only."""
        out = (ctypes.c_uint8 * max(1, len(data)))()
        orig = self.lib.oracle_origin_forward(_buf(data), len(data), out)
        return bytes(out)[:len(data)], int(orig)

    @staticmethod
    def forward(data: bytes):
        """(last column, origin row) of one block: rotations ordered by rank doubling over cyclic
        positions (numpy), equal rotations by position -- the same result as forward_slow."""
        import numpy as np

        n = len(data)
        if n == 0:
            return b"", 0
        s = np.frombuffer(data, dtype=np.uint8)
        rank = s.astype(np.int64)
        idx = np.arange(n, dtype=np.int64)
        h = 1
        while True:
            key = rank * (int(rank.max()) + 2) + rank[(idx + h) % n]
            order = np.argsort(key, kind="stable")
            ks = key[order]
            new = np.empty(n, dtype=np.int64)
            new[order] = np.concatenate(([0], np.cumsum(ks[1:] != ks[:-1])))
            rank = new
            if int(rank.max()) == n - 1 or h >= n:
                break
            h *= 2
        order = np.argsort(rank, kind="stable")
        last = s[(order - 1) % n].tobytes()
        return last, int(np.nonzero(order == 0)[0][0])

    def forward_buffer(self, data: bytes, block_size: int) -> bytes:
        """The container: per block 0xFF, the origin row in three bytes big-endian, the last column."""
        out = bytearray()
        for off in range(0, len(data), block_size):
            last, orig = self.forward(data[off:off + block_size])
            out += bytes([0xFF, (orig >> 16) & 255, (orig >> 8) & 255, orig & 255]) + last
        return bytes(out)

    def inverse(self, last: bytes, orig: int) -> bytes:
        out = (ctypes.c_uint8 * max(1, len(last)))()
        if self.lib.oracle_origin_inverse(_buf(last), len(last), orig, out):
            raise ValueError("origin row outside the block")
        return bytes(out)[:len(last)]

    def inverse_buffer(self, data: bytes, block_size: int) -> bytes:
        out = (ctypes.c_uint8 * max(1, len(data)))()
        r = self.lib.oracle_origin_inverse_buffer(_buf(data), len(data), block_size, out)
        if r < 0:
            raise MemoryError
        return bytes(out)[:r]


_origin = None


def origin() -> _OriginOracle:
    global _origin
    if _origin is None:
        _origin = _OriginOracle()
    return _origin
