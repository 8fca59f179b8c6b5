"""ctypes binding of the C ABI in ``include/bwtgpu.h`` plus a Python mirror of the
reference's public interface (``bwt_forward`` / ``bwt_inverse`` /
``bwt_forward_process_file`` / ``bwt_inverse_process_file``; reference
``src/bwt.hpp:9,14`` and ``src/inverse_bwt.hpp:8,13``).

Everything computes on the GPU through ``libbwtgpu.so``.  There is no CPU
fallback: a missing library or a missing device raises.
"""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# BWTGPU_LIB: another build of the same library (A/B experiments); the default is the in-tree one
LIB_PATH = os.environ.get("BWTGPU_LIB") or os.path.join(_PKG, "libbwtgpu.so")

OK = 0
ERR_NO_DEVICE = -1
ERR_INVALID = -2
ERR_BLOCK_TOO_LARGE = -3
ERR_WORKSPACE = -4
ERR_HIP = -5
ERR_NO_DELIMITER = -6
ERR_NO_FREE_BYTE = -7
ERR_INTERNAL = -8
MAX_BLOCK_SIZE = 1 << 30
LEN_NO_DELIMITER = 0xFFFFFFFF

# every symbol include/bwtgpu.h declares
SYMBOLS = [
    "bwtgpu_init", "bwtgpu_shutdown", "bwtgpu_device_count", "bwtgpu_last_error",
    "bwtgpu_num_blocks", "bwtgpu_forward_output_bytes", "bwtgpu_forward_workspace_bytes",
    "bwtgpu_inverse_workspace_bytes", "bwtgpu_forward_batch_device", "bwtgpu_forward_batch",
    "bwtgpu_inverse_batch_device", "bwtgpu_inverse_batch", "bwtgpu_presence_device",
    "bwtgpu_find_unused_byte", "bwtgpu_lowest_clear_bit", "bwtgpu_synth_device",
    "bwtgpu_alloc_host", "bwtgpu_free_host",
    "bwtgpu_pipeline_create", "bwtgpu_pipeline_input", "bwtgpu_pipeline_input_capacity",
    "bwtgpu_pipeline_submit", "bwtgpu_pipeline_collect", "bwtgpu_pipeline_destroy",
    "bwtgpu_pipeline_upload_presence", "bwtgpu_forward_batch_static", "bwtgpu_time_resident",
    "bwtgpu_pipeline_reserve", "bwtgpu_pipeline_stage", "bwtgpu_pipeline_stage_wait", "bwtgpu_pipeline_presence",
    "bwtgpu_pipeline_submit_resident", "bwtgpu_pipeline_run",
    "bwtgpu_inverse_origin_workspace_bytes", "bwtgpu_inverse_origin_batch_device", "bwtgpu_inverse_origin_batch",
    "bwtgpu_forward_origin_workspace_bytes", "bwtgpu_forward_origin_batch_device", "bwtgpu_forward_origin_batch",
]
STATIC_SCRATCH_BYTES = 64 << 20


class BwtGpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("bwtgpu error %d: %s" % (code, message))
        self.code = code


_lib = None
_vp, _sz, _u8, _i, _u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint8, ctypes.c_int, ctypes.c_uint64


def lib() -> ctypes.CDLL:
    """The loaded libbwtgpu.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "bwt_amd/libbwtgpu.so is missing: build it with `python -m bwt_amd.build` "
            "(hipcc, gfx950). bwt_amd has no CPU path.")
    L = ctypes.CDLL(LIB_PATH)
    L.bwtgpu_init.argtypes, L.bwtgpu_init.restype = [_i], _i
    L.bwtgpu_shutdown.argtypes, L.bwtgpu_shutdown.restype = [], None
    L.bwtgpu_device_count.argtypes, L.bwtgpu_device_count.restype = [], _i
    L.bwtgpu_last_error.argtypes, L.bwtgpu_last_error.restype = [], ctypes.c_char_p
    L.bwtgpu_num_blocks.argtypes, L.bwtgpu_num_blocks.restype = [_sz, _sz], _sz
    L.bwtgpu_forward_output_bytes.argtypes, L.bwtgpu_forward_output_bytes.restype = [_sz, _sz], _sz
    L.bwtgpu_forward_workspace_bytes.argtypes, L.bwtgpu_forward_workspace_bytes.restype = [_i, _sz, _sz], _sz
    L.bwtgpu_inverse_workspace_bytes.argtypes, L.bwtgpu_inverse_workspace_bytes.restype = [_i, _sz, _sz], _sz
    L.bwtgpu_forward_batch_device.argtypes = [_i, _vp, _sz, _sz, _u8, _vp, _vp, _sz, _vp]
    L.bwtgpu_forward_batch_device.restype = _i
    L.bwtgpu_forward_batch.argtypes, L.bwtgpu_forward_batch.restype = [_i, _vp, _sz, _sz, _u8, _vp], _i
    L.bwtgpu_forward_batch_static.argtypes = [_i, _vp, _sz, _sz, _u8, _vp, _vp, _vp]
    L.bwtgpu_forward_batch_static.restype = _i
    L.bwtgpu_time_resident.argtypes = [_i, _vp, _sz, _sz, _u8, _i, ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double)]
    L.bwtgpu_time_resident.restype = _i
    L.bwtgpu_inverse_origin_workspace_bytes.argtypes = [_i, _sz, _sz]
    L.bwtgpu_inverse_origin_workspace_bytes.restype = _sz
    L.bwtgpu_inverse_origin_batch_device.argtypes = [_i, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp]
    L.bwtgpu_inverse_origin_batch_device.restype = _i
    L.bwtgpu_inverse_origin_batch.argtypes = [_i, _vp, _sz, _sz, _vp, _vp]
    L.bwtgpu_inverse_origin_batch.restype = _i
    L.bwtgpu_forward_origin_workspace_bytes.argtypes = [_i, _sz, _sz]
    L.bwtgpu_forward_origin_workspace_bytes.restype = _sz
    L.bwtgpu_forward_origin_batch_device.argtypes = [_i, _vp, _sz, _sz, _vp, _vp, _vp, _sz, _vp]
    L.bwtgpu_forward_origin_batch_device.restype = _i
    L.bwtgpu_forward_origin_batch.argtypes = [_i, _vp, _sz, _sz, _vp, _vp]
    L.bwtgpu_forward_origin_batch.restype = _i
    L.bwtgpu_inverse_batch_device.argtypes = [_i, _vp, _sz, _sz, _u8, _vp, _vp, _vp, _sz, _vp]
    L.bwtgpu_inverse_batch_device.restype = _i
    L.bwtgpu_inverse_batch.argtypes = [_i, _vp, _sz, _sz, _u8, _vp, ctypes.POINTER(_sz)]
    L.bwtgpu_inverse_batch.restype = _i
    L.bwtgpu_presence_device.argtypes, L.bwtgpu_presence_device.restype = [_i, _vp, _sz, _vp, _vp], _i
    L.bwtgpu_find_unused_byte.argtypes, L.bwtgpu_find_unused_byte.restype = [_vp, _sz, _vp], _i
    L.bwtgpu_lowest_clear_bit.argtypes, L.bwtgpu_lowest_clear_bit.restype = [_vp], _i
    L.bwtgpu_synth_device.argtypes = [_i, _i, _u64, _u64, _vp, _sz, _vp, _sz, _vp]
    L.bwtgpu_synth_device.restype = _i
    L.bwtgpu_alloc_host.argtypes, L.bwtgpu_alloc_host.restype = [_sz], _vp
    L.bwtgpu_free_host.argtypes, L.bwtgpu_free_host.restype = [_vp], None
    L.bwtgpu_pipeline_create.argtypes, L.bwtgpu_pipeline_create.restype = [_i, _i, _sz, _sz, _i], _vp
    L.bwtgpu_pipeline_input.argtypes, L.bwtgpu_pipeline_input.restype = [_vp, _i], _vp
    L.bwtgpu_pipeline_input_capacity.argtypes, L.bwtgpu_pipeline_input_capacity.restype = [_vp], _sz
    L.bwtgpu_pipeline_submit.argtypes, L.bwtgpu_pipeline_submit.restype = [_vp, _i, _sz, _u8], _i
    L.bwtgpu_pipeline_collect.argtypes = [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_sz)]
    L.bwtgpu_pipeline_collect.restype = _i
    L.bwtgpu_pipeline_upload_presence.argtypes = [_vp, _i, _sz, ctypes.POINTER(ctypes.c_uint32)]
    L.bwtgpu_pipeline_upload_presence.restype = _i
    L.bwtgpu_pipeline_run.argtypes = [_vp, _i, _sz, ctypes.c_uint8, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.bwtgpu_pipeline_run.restype = _i
    L.bwtgpu_pipeline_reserve.argtypes, L.bwtgpu_pipeline_reserve.restype = [_vp, _sz], _i
    L.bwtgpu_pipeline_stage.argtypes, L.bwtgpu_pipeline_stage.restype = [_vp, _i, _sz, _sz], _i
    L.bwtgpu_pipeline_stage_wait.argtypes, L.bwtgpu_pipeline_stage_wait.restype = [_vp, _i], _i
    L.bwtgpu_pipeline_presence.argtypes, L.bwtgpu_pipeline_presence.restype = [_vp, ctypes.POINTER(ctypes.c_uint32)], _i
    L.bwtgpu_pipeline_submit_resident.argtypes, L.bwtgpu_pipeline_submit_resident.restype = [_vp, _i, _sz, ctypes.c_uint8], _i
    L.bwtgpu_pipeline_destroy.argtypes, L.bwtgpu_pipeline_destroy.restype = [_vp], None
    _lib = L
    return L


def last_error() -> str:
    return lib().bwtgpu_last_error().decode(errors="replace")


def check(rc: int) -> int:
    if rc < 0:
        raise BwtGpuError(rc, last_error())
    return rc


def init(n_devices: int = 0) -> int:
    return check(lib().bwtgpu_init(n_devices))


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def _delim(delimiter) -> int:
    if isinstance(delimiter, (bytes, bytearray, str)):
        if len(delimiter) != 1:
            raise ValueError("delimiter must be a single byte")
        return delimiter[0] if not isinstance(delimiter, str) else ord(delimiter)
    return int(delimiter) & 0xFF


# ---- host-buffer batch calls ----------------------------------------------------
def forward_batch(data, block_size: int, delimiter, device: int = 0) -> bytes:
    """Blocks of `block_size` input bytes -> concatenated (len+1)-byte transformed blocks."""
    a = _as_u8(data)
    L = lib()
    out = np.empty(L.bwtgpu_forward_output_bytes(a.size, block_size) if block_size else 0, dtype=np.uint8)
    check(L.bwtgpu_forward_batch(device, a.ctypes.data, a.size, block_size, _delim(delimiter), out.ctypes.data))
    return out.tobytes()


def inverse_batch(data, block_size: int, delimiter, device: int = 0) -> bytes:
    a = _as_u8(data)
    out = np.empty(max(1, a.size), dtype=np.uint8)
    n = _sz(0)
    check(lib().bwtgpu_inverse_batch(device, a.ctypes.data, a.size, block_size, _delim(delimiter),
                                     out.ctypes.data, ctypes.byref(n)))
    return out[:n.value].tobytes()


def find_unused_byte(data) -> int:
    """Lowest byte value absent from `data`, or -1 (reference src/bwt.cpp:19-47)."""
    a = _as_u8(data)
    mask = (ctypes.c_uint32 * 8)()
    rc = lib().bwtgpu_find_unused_byte(a.ctypes.data if a.size else None, a.size, mask)
    if rc == ERR_NO_FREE_BYTE:
        return -1
    return check(rc)


# ---- mirror of the reference's public interface ---------------------------------
def bwt_forward(input, delimiter=b"~") -> bytes:
    """reference src/bwt.hpp:9 -- one block: returns len(input)+1 bytes."""
    a = _as_u8(input)
    d = _delim(delimiter)
    if a.size == 0:
        return bytes([d])
    return forward_batch(a, a.size, d)


def bwt_inverse(bwt_str, delimiter=b"~") -> bytes:
    """reference src/inverse_bwt.hpp:8 -- one block of len+1 bytes -> the text."""
    a = _as_u8(bwt_str)
    if a.size == 0:
        raise BwtGpuError(ERR_NO_DELIMITER, "empty block holds no delimiter")
    if a.size == 1:
        return inverse_batch(a, 1, delimiter)
    return inverse_batch(a, a.size - 1, delimiter)


def bwt_forward_process_file(input_file: str, output_file: str, block_size: int) -> int:
    """reference src/bwt.hpp:14 / src/bwt.cpp:159-237: 0 on success, 1 on error (message on stderr)."""
    try:
        with open(input_file, "rb") as f:
            raw = np.frombuffer(f.read(), dtype=np.uint8)
    except OSError:
        print("Error: Could not open input file " + input_file, file=sys.stderr)
        return 1
    if block_size <= 0:
        print("Error: Block size must be greater than 0", file=sys.stderr)
        return 1
    # transform first, create the output file only once there is something to write
    try:
        d = find_unused_byte(raw)
        if d < 0:
            print("Error: Cannot find a unique delimiter (all 256 byte values appear in file)", file=sys.stderr)
            return 1
        body = forward_batch(raw, block_size, d) if raw.size else b""
    except BwtGpuError as e:
        print("Error: " + str(e), file=sys.stderr)
        return 1
    try:
        out = open(output_file, "wb")
    except OSError:
        print("Error: Could not open output file " + output_file, file=sys.stderr)
        return 1
    with out:
        out.write(bytes([d]))
        out.write(body)
    return 0


def bwt_inverse_process_file(input_file: str, output_file: str, block_size: int) -> int:
    """reference src/inverse_bwt.hpp:13 / src/inverse_bwt.cpp:66-99."""
    try:
        with open(input_file, "rb") as f:
            raw = np.frombuffer(f.read(), dtype=np.uint8)
    except OSError:
        print("Error: Could not open input file " + input_file, file=sys.stderr)
        return 1
    try:
        out = open(output_file, "wb")
    except OSError:
        print("Error: Could not open output file " + output_file, file=sys.stderr)
        return 1
    with out:
        if raw.size == 0:
            print("Error: Empty input file", file=sys.stderr)
            return 1
        try:
            out.write(inverse_batch(raw[1:], block_size, int(raw[0])))
        except BwtGpuError as e:
            print("Error: " + str(e), file=sys.stderr)
            return 1
    return 0


# ---- origin-pointer blocks: the container of the reference's bzip2 comparison tool ------------
def inverse_origin_batch(last_columns, block_size: int, orig_ptrs, device: int = 0) -> bytes:
    """Last columns of sorted rotations (block i at i*block_size) + the row of each block's text
    -> the texts (reference util/bzip2_bwt_utils.cpp:200-253, for a whole batch)."""
    a = _as_u8(last_columns)
    p = np.ascontiguousarray(orig_ptrs, dtype=np.uint32)
    if block_size <= 0:
        raise BwtGpuError(ERR_INVALID, "block_size must be greater than 0")
    if p.size != (a.size + block_size - 1) // block_size:
        raise BwtGpuError(ERR_INVALID, "one origin row per block expected")
    out = np.empty(max(1, a.size), dtype=np.uint8)
    check(lib().bwtgpu_inverse_origin_batch(device, a.ctypes.data if a.size else None, a.size, block_size,
                                            p.ctypes.data if p.size else None, out.ctypes.data))
    return out[:a.size].tobytes()


def forward_origin_batch(data, block_size: int, device: int = 0):
    """Texts (block i at i*block_size) -> (last columns of their sorted rotations back to back, the row of
    each block's text): what the reference's process_block produces per block
    (util/bzip2_bwt_utils.cpp:104-136)."""
    a = _as_u8(data)
    if block_size <= 0:
        raise BwtGpuError(ERR_INVALID, "block_size must be greater than 0")
    nb = (a.size + block_size - 1) // block_size
    out = np.empty(max(1, a.size), dtype=np.uint8)
    rows = np.zeros(max(1, nb), dtype=np.uint32)
    check(lib().bwtgpu_forward_origin_batch(device, a.ctypes.data if a.size else None, a.size, block_size,
                                            out.ctypes.data, rows.ctypes.data))
    return out[:a.size].tobytes(), rows[:nb]


def build_origin_container(last_columns, block_size: int, rows) -> bytes:
    """(last columns, origin rows) -> the container parse_origin_container reads (reference
    util/bzip2_bwt_utils.cpp:137-155)."""
    a = _as_u8(last_columns)
    parts = []
    for i, row in enumerate(rows):
        row = int(row)
        parts.append(bytes([0xFF, (row >> 16) & 0xFF, (row >> 8) & 0xFF, row & 0xFF]))
        parts.append(a[i * block_size:(i + 1) * block_size].tobytes())
    return b"".join(parts)


def parse_origin_container(raw, block_size: int):
    """Splits the container (per block: 0xFF, origin row as three bytes big-endian, the last column;
    reference util/bzip2_bwt_utils.cpp:157-198) into (last columns back to back, origin rows).  Like
    the reference's reader it stops, keeping the blocks before, at a wrong marker, a cut header, or
    a row outside its block."""
    a = _as_u8(raw)
    cols, rows, r = [], [], 0
    while r < a.size:
        if a[r] != 0xFF or a.size - r < 4:
            break
        orig = (int(a[r + 1]) << 16) | (int(a[r + 2]) << 8) | int(a[r + 3])
        n = min(block_size, a.size - r - 4)
        if n == 0 or orig >= n:
            break
        cols.append(a[r + 4:r + 4 + n])
        rows.append(orig)
        r += 4 + n
    last = np.concatenate(cols) if cols else np.empty(0, dtype=np.uint8)
    return last, np.asarray(rows, dtype=np.uint32)


def bzip2_format_inverse_process_file(input_file: str, output_file: str, block_size: int) -> int:
    """reference util/bzip2_bwt_utils.cpp:316-349 (Bzip2BWTProcessor::process_file_inverse)."""
    try:
        with open(input_file, "rb") as f:
            raw = np.frombuffer(f.read(), dtype=np.uint8)
    except OSError:
        print("Error: Cannot open input file: " + input_file, file=sys.stderr)
        return 1
    try:
        out = open(output_file, "wb")
    except OSError:
        print("Error: Cannot open output file: " + output_file, file=sys.stderr)
        return 1
    with out:
        last, rows = parse_origin_container(raw, block_size)
        try:
            out.write(inverse_origin_batch(last, block_size, rows) if last.size else b"")
        except BwtGpuError as e:
            print("Error: " + str(e), file=sys.stderr)
            return 1
    return 0
