"""Multi-GPU host logic on CPU: block scatter offsets and the delimiter agreement, with
two gloo ranks.  Each rank transforms only its shard (with the CPU oracle standing in for
the device) and rank 0 checks that the shards, placed at their closed-form offsets, are
byte-identical to the whole-file transform."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from bwt_amd import shard, synth


def test_shards_tile_the_file():
    for total, bs, world in [(0, 4, 2), (1, 4, 2), (10, 3, 4), (4096, 1024, 2), (4097, 1024, 8), (1 << 20, 65536, 3)]:
        nb = shard.num_blocks(total, bs)
        pos_in, pos_out, blocks = 0, 1, 0
        for r in range(world):
            s = shard.shard_of(total, bs, r, world)
            assert s.in_offset == pos_in and s.out_offset == pos_out and s.first_block == blocks
            pos_in += s.in_bytes
            pos_out += s.out_bytes
            blocks += s.num_blocks
        assert pos_in == total and blocks == nb and pos_out == total + 1 + nb


def test_mask_helpers():
    words = [0xFFFFFFFF, 0x0000FFFF, 0, 0, 0, 0, 0, 0]
    flags = shard.mask_words_to_flags(words)
    assert sum(flags) == 48 and shard.lowest_absent(flags) == 48
    assert shard.lowest_absent([1] * 256) == -1
    assert shard.agree_on_delimiter(flags) == 48  # single process: no group needed


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, bs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data_all = synth.generate("random255", 11, 0, total)
        data_all[data_all < 3] = 9  # values 0..2 unused -> the delimiter must come out as 0 ...
        if total:
            data_all[total - 1] = 0  # ... unless the LAST rank's shard uses it: then 1
        s = shard.shard_of(total, bs, rank, world)
        mine = data_all[s.in_offset:s.in_offset + s.in_bytes]
        flags = np.zeros(256, dtype=np.int32)
        flags[np.unique(mine)] = 1
        delim = shard.agree_on_delimiter(flags.tolist())
        orc = oracle.oracle()
        out = bytearray()
        for b in range(s.num_blocks):
            blk = mine[b * bs:(b + 1) * bs].tobytes()
            out += orc.forward(blk, delim)
        assert len(out) == s.out_bytes
        gathered = [None] * world
        dist.all_gather_object(gathered, (s.out_offset, bytes(out), delim))
        if rank == 0:
            whole = bytearray(total + 1 + shard.num_blocks(total, bs))
            whole[0] = delim
            for off, part, d in gathered:
                assert d == delim
                whole[off:off + len(part)] = part
            q.put((delim, bytes(whole) == orc.forward_buffer(data_all.tobytes(), bs, 2)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,bs", [(10000, 1024), (8192, 1024), (5, 4)])
def test_two_ranks_reproduce_the_whole_file(total, bs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    delim, same = q.get(timeout=10)
    assert delim == 1 and same
