#!/usr/bin/env python3
"""Round-trip BWT throughput on MI355X: forward + inverse over a synthetic shard that is
already resident in HBM, one process per GPU, blocks sharded across ranks (no collective
on the data path -- blocks are independent).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one forward pass plus one inverse pass of the hot path over the rank's whole
shard.  Rank 0 prints ONE JSON line.  `value` is input bytes of all ranks per second of
(forward + inverse), GB = 1e9 bytes.  `roofline` prices the dominant kernel (the forward
suffix-sort kernel, bwt_forward_lds_kernel on the headline workload) at its algorithmic bytes (read N, write N + nblocks per launch) over
its HIP-event duration; `cpu_baseline` times the reference CPU implementation on a bounded
sample of the same workload on this host's cores.  Beside the headline the line carries
  other_workloads   the other benchmark configurations (each with its own roofline entries, and the
                    Canterbury alice29.txt case with the CPU reference timed on the whole file),
  kernels           per-kernel time / algorithmic bytes / counter traffic of this workload as
                    profiled with rocprofv3 (read from profiles/kernels.json, which names its source),
  e2e               file to file through the command-line tools in tmpfs (PCIe, file I/O, process
                    start included) -- a separate figure, never `value`.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = 1 << 30
MIB = 1 << 20
HBM_PEAK = 8.0e12  # bytes/s, MI355X HBM3E

# name -> (kind, block_size, bytes per GPU, config id for the seed, description)
WORKLOADS = {
    "acgt-4g-256k": ("acgt", 256 * 1024, 4 * GIB, 3, "4 GiB synthetic ACGT text, 256 KiB blocks (BASELINE configs[2])"),
    "random-1g-64k": ("random255", 64 * 1024, 1 * GIB, 2, "1 GiB uniform bytes over 0x00-0xFE, 64 KiB blocks (BASELINE configs[1])"),
    "period7-1g-256k": ("period7", 256 * 1024, 1 * GIB, 4, "1 GiB period-7 text, 256 KiB blocks (BASELINE configs[3])"),
    "mixed-4g-1m": ("mixed", 1024 * 1024, 4 * GIB, 5, "4 GiB-per-GPU slice of the mixed corpus, 1 MiB blocks (BASELINE configs[4], sharded)"),
    # sub-cases of configs[3]: period 1 and a random unit of period 64 (same seed family as period-7)
    "period1-1g-256k": ("period1", 256 * 1024, 1 * GIB, 4, "1 GiB of one repeated byte, 256 KiB blocks (sub-case of BASELINE configs[3])"),
    "period64-1g-256k": ("period64", 256 * 1024, 1 * GIB, 4, "1 GiB of a random 64-byte unit repeated, 256 KiB blocks (sub-case of BASELINE configs[3])"),
    # not a BASELINE config: the shape of a single file (few large blocks), which takes the device-wide
    # forward path and the tiled inverse table build instead of the one-workgroup-per-block kernels
    "acgt-16m-1m": ("acgt", 1024 * 1024, 16 * MIB, 3, "16 MiB ACGT text, 1 MiB blocks: few large blocks (one file)"),
    # not a BASELINE config: natural text (alice29.txt repeated; a block is shorter than the file, so no block holds a
    # repeat of its own), which none of the in-LDS forward kernels takes -- the general kernels' rate on real text
    "text-1g-64k": ("text", 64 * 1024, 1 * GIB, 1, "1 GiB of alice29.txt repeated, 64 KiB blocks (natural text through the general kernels)"),
    # BASELINE configs[0]: the reference's own CPU-runnable case (bytes per GPU = the file's length)
    "alice29-128k": ("file:alice29.txt", 128 * 1024, 0, 1, "Canterbury alice29.txt, 128 KiB blocks (BASELINE configs[0])"),
}
CORPUS = os.path.join(ROOT, "data", "canterbury_corpus")
DEFAULT_WORKLOAD = "acgt-4g-256k"
METRIC = "round-trip BWT GB/s at 256KB blocks, 1/8xMI355X; % of HBM roofline"


def cpu_baseline_worker(kind: str, block_size: int, sample: int, seed: int, ncores: int) -> None:
    """Child process: time the CPU reference on `sample` bytes, file to file (as the
    reference's own benchmark does, reference tests/performance.cpp:84-117)."""
    avail = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, set(avail[:ncores]))  # hardware_concurrency() follows the affinity mask
    import oracle
    from bwt_amd import synth

    import numpy as np

    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    if kind.startswith("file:"):
        data = np.fromfile(os.path.join(CORPUS, kind[5:]), dtype=np.uint8)
        what = "the whole file (%d bytes), file to file in tmpfs" % data.size
    else:
        data = synth.generate(kind, seed, 0, sample, text=text)
        what = "first %d MiB of the workload, file to file in tmpfs" % (sample // MIB)
    sample = int(data.size)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm) as tmp:
        src, fwd, back = (os.path.join(tmp, n) for n in ("in", "fwd", "back"))
        data.tofile(src)
        if oracle.have_reference():
            impl, kind_name = oracle.reference(), "reference"
            run_f = lambda: impl.forward_file(src, fwd, block_size)
        else:
            impl, kind_name = oracle.oracle(), "port"
            run_f = lambda: impl.forward_file(src, fwd, block_size, ncores)
        t0 = time.perf_counter()
        rc1 = run_f()
        t1 = time.perf_counter()
        rc2 = impl.inverse_file(fwd, back, block_size)
        t2 = time.perf_counter()
        ok = rc1 == 0 and rc2 == 0 and open(back, "rb").read() == data.tobytes()
        import hashlib

        digest = hashlib.sha256(open(fwd, "rb").read()[1:]).hexdigest() if rc1 == 0 else ""
    print(json.dumps({"value": sample / (t2 - t0) / 1e9, "unit": "GB/s", "cores": len(avail[:ncores]),
                      "kind": kind_name, "forward_s": t1 - t0, "inverse_s": t2 - t1, "roundtrip_ok": ok,
                      "forward_sha256": digest,
                      "sample": what}))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--bytes-per-gpu", type=int, default=0, help="override the shard size (bytes)")
    ap.add_argument("--block-size", type=int, default=0, help="override the block size (exploration only)")
    ap.add_argument("--kind", default="", help="override the generator kind (exploration only)")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--other-cpu-sample-mib", type=int, default=16, help="CPU reference sample of each of the other workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file run through the command-line tools")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the one-step runs of the other benchmark configurations")
    ap.add_argument("--cpu-baseline-worker", nargs=5, metavar=("KIND", "BS", "SAMPLE", "SEED", "CORES"))
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        k, bs, sample, seed, cores = args.cpu_baseline_worker
        cpu_baseline_worker(k, int(bs), int(sample), int(seed), int(cores))
        return

    import torch
    import torch.distributed as dist

    from bwt_amd import api, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    kind, bs, nbytes, cfg, desc = WORKLOADS[args.workload]
    if args.bytes_per_gpu:
        nbytes = args.bytes_per_gpu
    if args.block_size:
        bs = args.block_size
    if args.kind:
        kind = args.kind
    L = api.lib()
    api.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    text = open(os.path.join(CORPUS, "alice29.txt"), "rb").read()
    d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)

    def dptr(t):
        return ctypes.c_void_p(t.data_ptr())

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(kind, bs, nbytes, cfg, steps, warmup):
        """One workload: generate the shard in HBM, time `steps` forward+inverse passes, verify."""
        seed = synth.base_seed(cfg)
        file_bytes = None
        if kind.startswith("file:"):  # a corpus file: every rank transforms the whole file
            file_bytes = open(os.path.join(CORPUS, kind[5:]), "rb").read()
            nbytes = len(file_bytes)
        nblocks = L.bwtgpu_num_blocks(nbytes, bs)
        fwd_bytes = nbytes + nblocks
        # ---- shard `rank` of the global synthetic stream, generated in HBM ---------------------
        d_in = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d_fwd = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
        d_back = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d_lens = torch.empty(nblocks, dtype=torch.int32, device=dev)
        ws_f = L.bwtgpu_forward_workspace_bytes(local_rank, nbytes, bs)
        ws_i = L.bwtgpu_inverse_workspace_bytes(local_rank, fwd_bytes, bs)
        d_ws_f = torch.empty(ws_f, dtype=torch.uint8, device=dev)
        d_ws_i = torch.empty(ws_i, dtype=torch.uint8, device=dev)
        if file_bytes is not None:
            d_in.copy_(torch.frombuffer(bytearray(file_bytes), dtype=torch.uint8))
        else:
            api.check(L.bwtgpu_synth_device(local_rank, synth.KINDS[kind], seed, rank * nbytes, dptr(d_in), nbytes,
                                            dptr(d_text), d_text.numel(), stream))
        # delimiter: lowest byte value absent from the WHOLE dataset = OR of the shards' presence masks
        d_mask = torch.zeros(8, dtype=torch.int32, device=dev)
        api.check(L.bwtgpu_presence_device(local_rank, dptr(d_in), nbytes, dptr(d_mask), stream))
        flags = shard.mask_words_to_flags([v & 0xFFFFFFFF for v in d_mask.cpu().tolist()])
        delim = shard.agree_on_delimiter(flags, device=dev)  # 256 flags, set-up only: not on the timed path
        if delim < 0:
            raise SystemExit("the workload uses all 256 byte values: the format has no delimiter for it")

        def forward():
            api.check(L.bwtgpu_forward_batch_device(local_rank, dptr(d_in), nbytes, bs, delim, dptr(d_fwd),
                                                    dptr(d_ws_f), d_ws_f.numel(), stream))

        def inverse():
            api.check(L.bwtgpu_inverse_batch_device(local_rank, dptr(d_fwd), fwd_bytes, bs, delim, dptr(d_back),
                                                    dptr(d_lens), dptr(d_ws_i), d_ws_i.numel(), stream))

        for _ in range(warmup):
            forward()
            inverse()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        sync_all()
        t0 = time.perf_counter()
        for s in range(steps):
            ev[s][0].record()
            forward()
            ev[s][1].record()
            inverse()
            ev[s][2].record()
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
        inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
        # ---- checks outside the timed region ---------------------------------------------------
        roundtrip_ok = bool(torch.equal(d_in, d_back))
        lens_ok = (bool((d_lens[:-1] == bs).all().item())
                   and int(d_lens[-1].item()) == nbytes - (nblocks - 1) * bs)
        status = d_ws_f[:8].cpu().numpy().view("uint32")  # [1] = forward kernel error flags
        ph = d_ws_f[64:64 + 184].cpu().numpy().view("uint64").astype(float)
        names = ["census", "keys0_count", "partition0", "rerank0", "keys", "sort", "rerank", "rank_rebuild",
                 None, None, "bucket_rerank0_lds", "bucket_sort0_hbm", "bucket_sort0_lds",
                 "sort_lds", "rerank_lds", "periodic_runs", "periodic_vote", "periodic_breaks", "periodic_settle",
                 "block_period_search", "block_period_expand"]
        tot = max(1.0, float(sum(v for k, v in zip(names, ph) if k)))
        phase_share = {k: round(float(v) / tot, 4) for k, v in zip(names, ph) if k}
        phase_share["later_rounds_per_block"] = round(float(ph[9]) / max(1.0, float(ph[8])), 3)
        phase_share["short_stretch_given_up"] = [int(ph[21]), int(ph[22])]
        # the text-in-LDS forward kernel's own clocks (u64 words 32.. of the header's clock area)
        lds = d_ws_f[64 + 8 * 32:64 + 8 * 40].cpu().numpy().view("uint64").astype(float)
        lds_tot = max(1.0, float(lds[:6].sum()))
        for k, v in zip(["sample", "pack", "count", "assign", "partition", "sort"], lds[:6]):
            phase_share["lds_" + k] = round(float(v) / lds_tot, 4)
        phase_share["lds_blocks_done_given_up"] = [int(lds[6]), int(lds[7])]
        t16 = d_ws_i[256:256 + 64].cpu().numpy().view("uint64").astype(float)  # the in-LDS table inverse's clocks
        if t16[6]:
            t16_tot = max(1.0, float(t16[:6].sum() + t16[7]))
            phase_share["inverse_t16"] = dict({k: round(float(v) / t16_tot, 4) for k, v in zip(["count", "scan", "table", "walk1", "order", "walk2"], t16[:6])},
                                              other=round(float(t16[7]) / t16_tot, 4), blocks=int(t16[6]))
        inv_clk = d_ws_i[64:64 + 128].cpu().numpy().view("uint64")  # debug builds of the library only
        if inv_clk.any():
            phase_share["inverse_stitch_clocks"] = [int(v) for v in inv_clk]
        ok = torch.tensor([int(roundtrip_ok and lens_ok and int(status[1]) == 0)], device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return {"elapsed": elapsed, "fwd_ms": fwd_ms, "inv_ms": inv_ms, "ok": bool(ok.item()),
                "status": int(status[1]), "phase_share": phase_share, "delim": delim, "nblocks": nblocks,
                "fwd_bytes": fwd_bytes, "d_fwd": d_fwd, "seed": seed, "nbytes": nbytes, "d_in": d_in}

    def roofline_entries(name, nbytes, nblocks, fwd_ms, inv_ms):
        """Both directions of one workload priced at their algorithmic bytes (forward: read N, write
        N + nblocks; inverse the other way round) over the HIP-event time of the pass on its stream."""
        algo = 2 * nbytes + nblocks
        out = {}
        for key, ms, what in (("roofline", fwd_ms, "forward pass (all its launches: bwt_forward_lds_kernel for blocks whose text fits the LDS, bwt_forward_kernel x2 behind it for the rest; the device-wide resident kernel for few large blocks)"),
                              ("roofline_inverse", inv_ms, "inverse pass (all its launches: bwt_inverse_packed_kernel for blocks over at most four symbols and bwt_inverse_t16_kernel for blocks of at most 64 KiB, both inside the LDS; table build, walk, stitch, place for the others)")):
            out[key] = {"bound": "hbm", "kernel": what, "achieved": algo / (ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": algo / (ms * 1e-3) / HBM_PEAK,
                        "traffic": None, "algorithmic_bytes_per_launch": algo, "avg_launch_ms": ms}
        t = traffic_table.get(name)
        if isinstance(t, dict):
            out["roofline"]["traffic"] = t.get("forward")
            out["roofline_inverse"]["traffic"] = t.get("inverse")
        elif t is not None:
            out["roofline"]["traffic"] = t
        return out

    def cpu_reference(kind, bs, sample, seed, d_fwd):
        """The CPU reference on a bounded sample, in a child process; its forward bytes must equal the GPU's."""
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", kind,
                                  str(bs), str(sample), str(seed), "16"], capture_output=True, text=True,
                                 timeout=900, check=True).stdout.strip().splitlines()[-1]
            cpu = json.loads(out)
            import hashlib

            nb_s = (sample + bs - 1) // bs
            gpu_prefix = d_fwd[:sample + nb_s].cpu().numpy().tobytes()
            cpu["gpu_forward_matches"] = (hashlib.sha256(gpu_prefix).hexdigest() == cpu.pop("forward_sha256"))
            return cpu
        except Exception as exc:  # the baseline is a reported number, never a reason to lose the GPU line
            return {"error": str(exc)[:200]}

    def e2e_files(d_in, nbytes, bs):
        """File to file through the command-line tools in tmpfs, like the reference's own benchmark
        (reference tests/performance.cpp:84-117): process start, device initialisation, file reads and
        writes and PCIe both ways are all inside the times.  Never part of `value`."""
        tools = [os.path.join(ROOT, "build", t) for t in ("bwt", "inverse_bwt")]
        if not all(os.path.exists(t) for t in tools) or not os.path.isdir("/dev/shm"):
            return {"error": "build/bwt or build/inverse_bwt or /dev/shm missing"}
        n = min(nbytes, GIB)
        try:
            with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
                src, fwd, back = (os.path.join(tmp, x) for x in ("in", "fwd", "back"))
                host = d_in[:n].cpu().numpy()
                host.tofile(src)
                t0 = time.perf_counter()
                rc1 = subprocess.run([tools[0], src, fwd, str(bs)], capture_output=True, timeout=300).returncode
                t1 = time.perf_counter()
                rc2 = subprocess.run([tools[1], fwd, back, str(bs)], capture_output=True, timeout=300).returncode
                t2 = time.perf_counter()
                import numpy as np

                same = rc1 == 0 and rc2 == 0 and bool(np.array_equal(np.fromfile(back, dtype=np.uint8), host))
            return {"value": n / (t2 - t0) / 1e9, "unit": "GB/s", "bytes": n, "block_size": bs,
                    "bwt_s": t1 - t0, "inverse_bwt_s": t2 - t1, "roundtrip_verified": same,
                    "what": "build/bwt then build/inverse_bwt on the first %d MiB of the workload as a file in tmpfs; "
                            "process start, device initialisation, file I/O and PCIe included" % (n // MIB)}
        except Exception as exc:
            return {"error": str(exc)[:200]}

    def load_json(name):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            return {}

    traffic_table = load_json("traffic.json")
    kernels_table = load_json("kernels.json")

    r = measure(kind, bs, nbytes, cfg, args.steps, args.warmup)
    elapsed, fwd_ms, inv_ms, all_ok = r["elapsed"], r["fwd_ms"], r["inv_ms"], r["ok"]
    nblocks, fwd_bytes, delim, d_fwd, seed = r["nblocks"], r["fwd_bytes"], r["delim"], r["d_fwd"], r["seed"]
    phase_share, status1, nbytes = r["phase_share"], r["status"], r["nbytes"]

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sample = min(nbytes, args.cpu_sample_mib * MIB)
            if sample > bs:
                sample -= sample % bs
            cpu = cpu_reference(kind, bs, sample, seed, d_fwd)
        e2e = None
        if world == 1 and not args.no_e2e:
            e2e = e2e_files(r["d_in"], nbytes, bs)
        total_in = nbytes * world
        step_s = elapsed / args.steps
        line = {
            "metric": METRIC,
            "value": total_in / step_s / 1e9,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.workload, "description": desc, "block_size": bs,
                       "bytes_per_gpu": nbytes, "blocks_per_gpu": nblocks, "delimiter": delim,
                       "parallelism": "block-sharded x%d, no collective" % world},
        }
        line.update(roofline_entries(args.workload, nbytes, nblocks, fwd_ms, inv_ms))
        line.update({
            "cpu_baseline": cpu,
            "forward_ms": fwd_ms,
            "inverse_ms": inv_ms,
            "forward_GBps": nbytes / (fwd_ms * 1e-3) / 1e9,
            "inverse_GBps": nbytes / (inv_ms * 1e-3) / 1e9,
            "roundtrip_pct_of_hbm_roofline": 100.0 * (4 * nbytes + 2 * nblocks) * world / step_s / (HBM_PEAK * world),
            "roundtrip_verified": all_ok,
            "kernel_status": status1,
            "forward_phase_share": phase_share,
            "kernels": kernels_table.get(args.workload),
            "e2e": e2e,
        })
    # ---- the other benchmark configurations, a few timed steps each (same code path) ------------
    others = []
    if not args.no_other_workloads and not (args.block_size or args.kind or args.bytes_per_gpu):
        del r, d_fwd
        torch.cuda.empty_cache()
        for name in sorted(WORKLOADS):
            if name == args.workload:
                continue
            k2, bs2, nb2, cfg2, desc2 = WORKLOADS[name]
            small = k2.startswith("file:") or nb2 <= 64 * MIB  # sub-millisecond passes: more steps
            r2 = measure(k2, bs2, nb2, cfg2, 20 if small else 3, 2 if small else 1)
            steps2 = 20 if small else 3
            entry = {"workload": name, "description": desc2, "block_size": bs2, "bytes_per_gpu": r2["nbytes"],
                     "steps": steps2, "value": r2["nbytes"] * world * steps2 / r2["elapsed"] / 1e9, "unit": "GB/s",
                     "forward_ms": r2["fwd_ms"], "inverse_ms": r2["inv_ms"],
                     "roundtrip_verified": r2["ok"],
                     "later_rounds_per_block": r2["phase_share"]["later_rounds_per_block"]}
            entry.update(roofline_entries(name, r2["nbytes"], r2["nblocks"], r2["fwd_ms"], r2["inv_ms"]))
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # the CPU reference beside every workload, its forward bytes compared with the GPU's: the
                # reference's own CPU-runnable case (the file) in full, the others on a prefix of whole blocks
                sample2 = r2["nbytes"]
                if not k2.startswith("file:"):
                    sample2 = min(sample2, args.other_cpu_sample_mib * MIB)
                    if sample2 > bs2:
                        sample2 -= sample2 % bs2
                entry["cpu_baseline"] = cpu_reference(k2, bs2, sample2, r2["seed"], r2["d_fwd"])
            others.append(entry)
            all_ok = all_ok and r2["ok"]
            del r2
            torch.cuda.empty_cache()
    if rank == 0:
        line["other_workloads"] = others
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not all_ok:
        raise SystemExit("round trip did not reproduce the input")


if __name__ == "__main__":
    main()
