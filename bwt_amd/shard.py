"""Host-side block scatter for multi-GPU runs (one process per GPU).

Blocks are independent (reference ``src/bwt.cpp:143-156``: a worker needs only its chunk
and the file-wide delimiter), so a file is cut into contiguous block ranges, one per rank,
and every offset is closed-form: block ``i`` reads ``i * block_size`` and writes
``1 + i * (block_size + 1)`` (the leading 1 is the file's delimiter header byte,
reference ``src/bwt.cpp:131-139``).  No data-path collective exists.  The only global
datum is the delimiter: the lowest byte value absent from the WHOLE file, i.e. the lowest
clear bit of the OR of the ranks' 256-bit presence masks -- 256 flags exchanged once,
before the transform starts.
"""
from __future__ import annotations

from typing import NamedTuple, Sequence


class Shard(NamedTuple):
    first_block: int
    num_blocks: int
    in_offset: int    # bytes into the input file
    in_bytes: int
    out_offset: int   # bytes into the transformed file (after the 1-byte header)
    out_bytes: int


def num_blocks(total_len: int, block_size: int) -> int:
    return (total_len + block_size - 1) // block_size


def shard_of(total_len: int, block_size: int, rank: int, world: int) -> Shard:
    """Contiguous block range of `rank`: blocks [nb*rank/world, nb*(rank+1)/world)."""
    if block_size <= 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("bad shard request")
    nb = num_blocks(total_len, block_size)
    lo, hi = nb * rank // world, nb * (rank + 1) // world
    in_off = lo * block_size
    in_end = min(total_len, hi * block_size)
    in_bytes = max(0, in_end - in_off)
    return Shard(lo, hi - lo, in_off, in_bytes, 1 + lo * (block_size + 1), in_bytes + (hi - lo))


def mask_words_to_flags(words: Sequence[int]) -> list:
    """8 x 32-bit presence words -> 256 flags (0/1)."""
    return [(int(words[v >> 5]) >> (v & 31)) & 1 for v in range(256)]


def lowest_absent(flags: Sequence[int]) -> int:
    """Lowest byte value whose flag is clear, or -1 (reference src/bwt.cpp:38-46)."""
    for v in range(256):
        if not flags[v]:
            return v
    return -1


def agree_on_delimiter(local_flags: Sequence[int], device=None) -> int:
    """OR the ranks' presence flags (an element-wise MAX all-reduce of 256 ints when a
    process group is initialised; a no-op for a single process) and pick the delimiter."""
    import torch
    import torch.distributed as dist

    t = torch.tensor(list(local_flags), dtype=torch.int32, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return lowest_absent(t.cpu().tolist())
