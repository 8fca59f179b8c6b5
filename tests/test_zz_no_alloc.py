"""bwtgpu_forward_batch_static -- the forward entry point that allocates nothing (fixed scratch
array inside the library) -- gives the bytes of bwtgpu_forward_batch_device, call after call on
the same GPU: the second call is smaller than the first, so whatever the first left in the shared
scratch must not matter.  Bit-exact is the bar (byte and index work)."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from bwt_amd import api, synth
from conftest import CORPUS

pytestmark = pytest.mark.gpu


def _both(L, torch, data: np.ndarray, bs: int, delim: int):
    """(bytes from the static entry point, its status word, bytes from the workspace entry point)."""
    dev = torch.device("cuda", 0)
    n = int(data.size)
    nb = L.bwtgpu_num_blocks(n, bs)
    d_in = torch.from_numpy(data.copy()).to(dev)
    d_a = torch.full((n + nb + 64,), 0xA5, dtype=torch.uint8, device=dev)
    d_b = torch.full((n + nb + 64,), 0x5A, dtype=torch.uint8, device=dev)
    d_status = torch.full((1,), 0x7FFFFFFF, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    vp = ctypes.c_void_p
    api.check(L.bwtgpu_forward_batch_static(0, vp(d_in.data_ptr()), n, bs, delim, vp(d_a.data_ptr()),
                                            vp(d_status.data_ptr()), stream))
    ws = L.bwtgpu_forward_workspace_bytes(0, n, bs)
    d_ws = torch.empty(max(ws, 256), dtype=torch.uint8, device=dev)
    api.check(L.bwtgpu_forward_batch_device(0, vp(d_in.data_ptr()), n, bs, delim, vp(d_b.data_ptr()),
                                            vp(d_ws.data_ptr()), d_ws.numel(), stream))
    torch.cuda.synchronize()
    a, b = d_a.cpu().numpy(), d_b.cpu().numpy()
    assert (a[n + nb:] == 0xA5).all() and (b[n + nb:] == 0x5A).all()  # nothing written past the output
    return a[:n + nb].tobytes(), int(d_status.item()), b[:n + nb].tobytes()


def test_static_scratch_forward_matches_workspace_forward():
    import torch

    L = api.lib()
    assert api.init(1) >= 1
    orc = oracle.oracle()
    text = np.frombuffer(open(os.path.join(CORPUS, "alice29.txt"), "rb").read(), dtype=np.uint8)
    # two calls in a row, the second with a smaller input; then shapes that use the other kernels
    # behind the entry point (periodic blocks are put off to the second launch, one-wave blocks,
    # more blocks than the scratch has slices for, a single block, blocks of 1 MiB)
    cases = [
        (synth.generate("acgt", synth.base_seed(3), 0, 40 * 65536 + 777), 65536),
        (synth.generate("acgt", synth.base_seed(3), 5, 3 * 8192 + 11), 8192),
        (synth.generate("period7", 1, 0, 5 * 262144 + 5), 262144),
        (text, 131072),
        (synth.generate("random255", 9, 0, 3000 * 512 + 100), 512),
        (synth.generate("random255", 9, 0, 400 * 16384 + 1), 16384),
        (synth.generate("acgt", 4, 0, 100000), 1 << 30),
        (np.tile(text, 15)[:2 * (1 << 20) + 4321], 1 << 20),
        (synth.generate("random255", 2, 0, 70000), 70000),
    ]
    for i, (data, bs) in enumerate(cases):
        present = set(np.unique(data).tolist())
        delim = next(v for v in range(256) if v not in present)
        got, status, want = _both(L, torch, data, bs, delim)
        assert status == 0, (i, status)
        assert got == want, (i, data.size, bs)
        if data.size <= 1 << 20:  # and against the CPU oracle where that takes no time
            assert got == orc.forward_buffer(data.tobytes(), min(bs, data.size), 8)[1:], i


def test_static_scratch_limits_are_reported():
    import torch

    L = api.lib()
    assert api.init(1) >= 1
    dev = torch.device("cuda", 0)
    vp = ctypes.c_void_p
    d_in = torch.zeros(3 << 20, dtype=torch.uint8, device=dev)
    d_out = torch.zeros((3 << 20) + 8, dtype=torch.uint8, device=dev)
    # a block beyond 2 MiB - 1 has no slice in the fixed scratch
    rc = L.bwtgpu_forward_batch_static(0, vp(d_in.data_ptr()), 3 << 20, 3 << 20, 255, vp(d_out.data_ptr()), None, None)
    assert rc == api.ERR_BLOCK_TOO_LARGE and "static scratch" in api.last_error()
    # a batch whose block list alone (4 bytes per block) outgrows the array
    d_many = torch.zeros(17_000_000, dtype=torch.uint8, device=dev)
    d_many_out = torch.zeros(34_000_000, dtype=torch.uint8, device=dev)
    rc = L.bwtgpu_forward_batch_static(0, vp(d_many.data_ptr()), 17_000_000, 1, 255, vp(d_many_out.data_ptr()), None, None)
    assert rc == api.ERR_WORKSPACE and "exceeds the static scratch" in api.last_error()
    # block_size 0 and null pointers are refused like everywhere else
    assert L.bwtgpu_forward_batch_static(0, vp(d_in.data_ptr()), 100, 0, 255, vp(d_out.data_ptr()), None, None) == api.ERR_INVALID
    assert L.bwtgpu_forward_batch_static(0, None, 100, 10, 255, vp(d_out.data_ptr()), None, None) == api.ERR_INVALID
    # an empty batch is fine and writes a zero status
    d_status = torch.full((1,), 5, dtype=torch.int32, device=dev)
    assert L.bwtgpu_forward_batch_static(0, None, 0, 10, 255, None, vp(d_status.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert int(d_status.item()) == 0
    torch.cuda.synchronize()


_FIRST_CALL = r"""
import ctypes, sys
import numpy as np
import torch
torch.cuda.is_available()
sys.path.insert(0, {root!r})
import oracle
from bwt_amd import api, synth
L = api.lib()
dev = torch.device("cuda", 0)
data = synth.generate("acgt", synth.base_seed(3), 0, 5 * 65536 + 777)
n, bs, delim = int(data.size), 65536, 0
nb = L.bwtgpu_num_blocks(n, bs)
d_in = torch.from_numpy(data.copy()).to(dev)
d_out = torch.zeros(n + nb, dtype=torch.uint8, device=dev)
d_status = torch.full((1,), 0x7FFFFFFF, dtype=torch.int32, device=dev)
vp = ctypes.c_void_p
stream = torch.cuda.current_stream().cuda_stream
# before bwtgpu_init the entry point refuses: it never binds a device by itself
rc = L.bwtgpu_forward_batch_static(0, vp(d_in.data_ptr()), n, bs, delim, vp(d_out.data_ptr()), vp(d_status.data_ptr()), stream)
assert rc == api.ERR_NO_DEVICE, rc
assert L.bwtgpu_init(1) >= 1
# ... and right after it, as the library's first call that touches the device, it works
rc = L.bwtgpu_forward_batch_static(0, vp(d_in.data_ptr()), n, bs, delim, vp(d_out.data_ptr()), vp(d_status.data_ptr()), stream)
assert rc == 0, api.last_error()
torch.cuda.synchronize()
assert int(d_status.item()) == 0
want = oracle.oracle().forward_buffer(data.tobytes(), bs, 4)
assert want[0] == delim and d_out.cpu().numpy().tobytes() == want[1:]
print("first-call ok")
"""


def test_static_entry_point_is_usable_as_the_first_call_after_init():
    """A fresh process: the static entry point refuses before bwtgpu_init (it binds no device itself, which
    would create a stream and pinned memory), and works as the very first device call after it."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _FIRST_CALL.format(root=root)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "first-call ok" in out.stdout, out.stdout + out.stderr
