"""Forward and inverse time of 512 MiB of DNA-like text: symbol frequencies 0.3 / 0.2 / 0.2 / 0.3, per block twenty dispersed
repeats (lengths geometric, mean 300), ten tandem repeats or homopolymer runs (unit 1-6, 20-200 symbols), a run of 100 N in
every hundredth block; how many blocks the in-LDS forward kernel finished or gave up and where the general kernels' time went.
Run on a GPU box: python tools/dna_like_bench.py [block size]."""
import sys, time, torch
import numpy as np
sys.path.insert(0, '.')
from bwt_amd import api
L = api.lib(); api.init(1)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = 1 << 29
rng = np.random.default_rng(11)
nbk = n // bs
# one "genome" template of 64 blocks, tiled with fresh point mutations would create inter-block similarity only: build per block
probs = np.array([0.3, 0.2, 0.2, 0.3])
base = np.frombuffer(b"ACGT", np.uint8)[rng.choice(4, size=n, p=probs)].copy()
v = base.reshape(nbk, bs)
for b in range(nbk):
    row = v[b]
    for _ in range(20):  # dispersed repeats
        ln = int(min(bs // 8, rng.geometric(1 / 300.0)))
        s, d = int(rng.integers(0, bs - ln)), int(rng.integers(0, bs - ln))
        row[d:d + ln] = row[s:s + ln]
    for _ in range(10):  # tandem repeats and homopolymer runs
        unit = int(rng.integers(1, 7)); ln = int(rng.integers(20, 200)); d = int(rng.integers(0, bs - ln))
        row[d:d + ln] = np.resize(row[d:d + unit], ln)
    if b % 100 == 0:
        d = int(rng.integers(0, bs - 100)); row[d:d + 100] = ord('N')
x = torch.from_numpy(base).cuda()
out = torch.empty(n + nbk, dtype=torch.uint8, device='cuda')
back = torch.empty(n, dtype=torch.uint8, device='cuda'); lens = torch.empty(nbk, dtype=torch.int32, device='cuda')
wsb = L.bwtgpu_forward_workspace_bytes(0, n, bs); ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
wib = L.bwtgpu_inverse_workspace_bytes(0, n + nbk, bs); wi = torch.empty(wib, dtype=torch.uint8, device='cuda')
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    api.check(L.bwtgpu_forward_batch_device(0, x.data_ptr(), n, bs, 0, out.data_ptr(), ws.data_ptr(), wsb, 0))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter()
    api.check(L.bwtgpu_inverse_batch_device(0, out.data_ptr(), n + nbk, bs, 0, back.data_ptr(), lens.data_ptr(), wi.data_ptr(), wib, 0))
    torch.cuda.synchronize(); di = time.perf_counter() - t
lds = ws[64 + 8 * 32:64 + 8 * 40].cpu().numpy().view('uint64')
ph = ws[64:64 + 184].cpu().numpy().view('uint64')
names = ["census", "keys0_count", "partition0", "rerank0", "keys", "sort", "rerank", "rank_rebuild", "blocks", "later_rounds", "bucket_rerank0_lds", "bucket_sort0_hbm", "bucket_sort0_lds", "sort_lds", "rerank_lds", "periodic_runs", "periodic_vote", "periodic_breaks", "periodic_settle", "block_period_search", "block_period_expand"]
tot = float(sum(ph[:8]) + sum(ph[10:21])) or 1.0
print('DNA-like 512 MiB, bs', bs, ': fwd %.2f ms (%.1f per GiB), inv %.2f ms (%.1f per GiB), ok %s; in-LDS done / given up %d / %d;' % (dt * 1e3, dt * 2e3, di * 1e3, di * 2e3, bool(torch.equal(x, back)), lds[6], lds[7]),
      {k: round(float(v) / tot, 3) for k, v in zip(names, ph) if float(v) / tot > 0.03 and k not in ('blocks', 'later_rounds')}, 'general blocks', int(ph[8]), 'rounds per block %.1f' % (float(ph[9]) / max(1.0, float(ph[8]))))
