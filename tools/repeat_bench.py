"""Forward time of 1 GiB of uniform four-symbol text in 256 KiB blocks in which every block repeats one stretch of R symbols
(a copy of positions 1000.. at position 100000), and how many blocks the in-LDS forward kernel finished or gave up; run on
a GPU box: python tools/repeat_bench.py (BWTGPU_FORWARD_LDS=0 for the general kernels alone)."""
import sys, time, torch
sys.path.insert(0, '.')
from bwt_amd import api
L = api.lib(); api.init(1)
bs = 262144
n = 1 << 30
torch.manual_seed(7)
sym = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device='cuda')
for R in (0, 100, 200, 1000, 5000):
    x = sym[torch.randint(0, 4, (n,), device='cuda')]
    if R:
        v = x.view(-1, bs)
        v[:, 100000:100000 + R] = v[:, 1000:1000 + R]
    nb = n // bs
    out = torch.empty(n + nb, dtype=torch.uint8, device='cuda')
    wsb = L.bwtgpu_forward_workspace_bytes(0, n, bs); ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        api.check(L.bwtgpu_forward_batch_device(0, x.data_ptr(), n, bs, 0, out.data_ptr(), ws.data_ptr(), wsb, 0))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    lds = ws[64 + 8 * 32:64 + 8 * 40].cpu().numpy().view('uint64')
    back = torch.empty(n, dtype=torch.uint8, device='cuda'); lens = torch.empty(nb, dtype=torch.int32, device='cuda')
    wib = L.bwtgpu_inverse_workspace_bytes(0, n + nb, bs); wi = torch.empty(wib, dtype=torch.uint8, device='cuda')
    api.check(L.bwtgpu_inverse_batch_device(0, out.data_ptr(), n + nb, bs, 0, back.data_ptr(), lens.data_ptr(), wi.data_ptr(), wib, 0))
    torch.cuda.synchronize()
    print('repeat of', R, 'symbols per block: fwd %.2f ms, round trip ok %s; in-LDS blocks done / given up: %d / %d' % (dt * 1e3, bool(torch.equal(x, back)), lds[6], lds[7]), flush=True)
    del x, out, ws, back, wi
