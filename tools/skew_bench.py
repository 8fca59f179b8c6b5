"""Forward and inverse time of 1 GiB of four-symbol text in 256 KiB blocks for several symbol distributions, and how
many blocks the in-LDS forward kernel finished or gave up (run on a GPU box: python tools/skew_bench.py)."""
import sys, time, torch
sys.path.insert(0, '.')
from bwt_amd import api
L = api.lib(); api.init(1)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = 1 << 30
torch.manual_seed(5)
sym = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device='cuda')
for probs in [(0.25, 0.25, 0.25, 0.25), (0.3, 0.3, 0.2, 0.2), (0.35, 0.3, 0.2, 0.15), (0.4, 0.3, 0.2, 0.1), (0.5, 0.25, 0.15, 0.1)]:
    cum = torch.tensor(probs, device='cuda').cumsum(0)[:3].contiguous()
    x = torch.empty(n, dtype=torch.uint8, device='cuda')
    for o in range(0, n, 1 << 26):
        r = torch.rand(1 << 26, device='cuda')
        x[o:o + (1 << 26)] = sym[torch.bucketize(r, cum)]
    del r
    nb = n // bs
    out = torch.empty(n + nb, dtype=torch.uint8, device='cuda')
    back = torch.empty(n, dtype=torch.uint8, device='cuda')
    lens = torch.empty(nb, dtype=torch.int32, device='cuda')
    wsb = L.bwtgpu_forward_workspace_bytes(0, n, bs); ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    wib = L.bwtgpu_inverse_workspace_bytes(0, n + nb, bs); wi = torch.empty(wib, dtype=torch.uint8, device='cuda')
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        api.check(L.bwtgpu_forward_batch_device(0, x.data_ptr(), n, bs, 0, out.data_ptr(), ws.data_ptr(), wsb, 0))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        t = time.perf_counter()
        api.check(L.bwtgpu_inverse_batch_device(0, out.data_ptr(), n + nb, bs, 0, back.data_ptr(), lens.data_ptr(), wi.data_ptr(), wib, 0))
        torch.cuda.synchronize(); di = time.perf_counter() - t
    lds = ws[64 + 8 * 32:64 + 8 * 40].cpu().numpy().view('uint64')
    print(probs, 'fwd %.2f ms inv %.2f ms ok %s; in-LDS blocks done / given up: %d / %d' % (dt * 1e3, di * 1e3, bool(torch.equal(x, back)), lds[6], lds[7]), flush=True)
    del x, out, back, ws, wi
