"""Row sharding of the LUT GEMV across the GPUs of one node.

Output rows are independent and K is never split, so every integer partial sum on a shard is identical
to the single-GPU one; the only exchange step is an all-gather of a produced activation vector before
the next LUT build.  Shards are aligned to the reference's M-tiles (bm/bits output rows) so a shard of
a reference-layout blob is a contiguous byte range (python/t_mac/weights.py:69-73).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class RowShard:
    rank: int
    tile_begin: int      # first reference M-tile of this rank
    tile_count: int      # tiles actually owned (may be fewer than tiles_per_rank on the last ranks)
    tiles_per_rank: int  # uniform (padded) tile count used for equal-size collectives
    rows_per_tile: int

    @property
    def row_begin(self) -> int:
        return self.tile_begin * self.rows_per_tile

    @property
    def rows(self) -> int:
        return self.tile_count * self.rows_per_tile

    @property
    def padded_rows(self) -> int:
        return self.tiles_per_rank * self.rows_per_tile


def plan_row_shards(Mw: int, bits: int, bm: int, world: int) -> List[RowShard]:
    """Split Mw output rows into `world` tile-aligned shards; ragged tails get fewer (possibly zero) tiles."""
    if (Mw * bits) % bm:
        raise ValueError("Mw*bits must be a multiple of bm")
    rpt = bm // bits
    ntiles = Mw // rpt
    per = (ntiles + world - 1) // world
    shards = []
    for r in range(world):
        b = min(r * per, ntiles)
        e = min(b + per, ntiles)
        shards.append(RowShard(r, b, e - b, per, rpt))
    return shards


def shard_blob_ranges(shard: RowShard, K: int, bits: int, bm: int, group_size: int, zero_point: bool, scale_itemsize: int):
    """(byte_begin, byte_end) of the shard inside the reference-layout weight blob and scale blob."""
    tile_w = (bm // 2) * (K // 4)
    tile_s = (K // group_size) * (bm // bits) * (2 if zero_point else 1) * scale_itemsize
    return ((shard.tile_begin * tile_w, (shard.tile_begin + shard.tile_count) * tile_w),
            (shard.tile_begin * tile_s, (shard.tile_begin + shard.tile_count) * tile_s))


def all_gather_rows(local_out, shard: RowShard, Mw: int, group=None):
    """All-gather the per-rank output slices (equal, padded sizes) and trim to the logical Mw rows.
    local_out: torch tensor [..., shard.padded_rows]; returns [..., Mw]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    pieces = [torch.empty_like(local_out) for _ in range(world)]
    dist.all_gather(pieces, local_out.contiguous(), group=group)
    return torch.cat(pieces, dim=-1)[..., :Mw]
