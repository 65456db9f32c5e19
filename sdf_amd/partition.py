"""Framebuffer / query-batch partitioner (multi-GPU, one process per GPU).

The reference has no multi-device code at all (no threads, no
collectives); every query is a pure function of (tree, point)
(faux_shader.rs:96 takes &SdfTreeBuffer), so the path shards with the tree
replicated on every GPU:

  frames   8x8-pixel tiles numbered row-major, dealt round-robin: rank r of N
           renders tiles r, r+N, r+2N, ...  (interleaving balances sky against
           dense regions).  Each rank writes its tiles compactly, 64 pixels per
           tile, into ONE packed byte buffer
               [depth f32 | normal 3 x f32 | prim u32 | hit u8]   (21 B / pixel)
           so the final assembly is a single RCCL all-gather of equal chunks,
           followed by sdfgpu_untile_dev on the assembling rank.
  queries  contiguous slices of the point array; results stay sharded.
"""
BYTES_PER_PIXEL = 21
TILE = 8
TILE_PIXELS = TILE * TILE


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def n_tiles(width, height):
    tx, ty = tile_grid(width, height)
    return tx * ty


def rank_tiles(rank, world, tiles):
    """(tile_first, tile_stride, tile_count) of one rank."""
    return rank, world, len(range(rank, tiles, world))


def max_tiles_per_rank(world, tiles):
    return (tiles + world - 1) // world


def packed_bytes(world, tiles):
    """Bytes of one rank's packed buffer (equal on every rank: padded to the max tile count)."""
    return max_tiles_per_rank(world, tiles) * TILE_PIXELS * BYTES_PER_PIXEL


def planes(packed, world, tiles):
    """Views (hit u8, depth f32, normal f32 x3, prim i32) into one rank's packed uint8 torch tensor."""
    import torch

    px = max_tiles_per_rank(world, tiles) * TILE_PIXELS
    depth = packed[: px * 4].view(torch.float32)
    normal = packed[px * 4: px * 16].view(torch.float32)
    prim = packed[px * 16: px * 20].view(torch.int32)
    hit = packed[px * 20: px * 21]
    return hit, depth, normal, prim


def query_slice(rank, world, n):
    """Contiguous [lo, hi) slice of n queries for one rank."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
