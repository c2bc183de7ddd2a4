"""The multi-rank path on CPU: tile ownership, buffer layout and the gather/assemble step, with two
gloo ranks.  Rendering itself needs a GPU; here each rank fills its tile buffer with a function of the
pixel position exactly where mirt_render_tiles would put that pixel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import owned_tiles, scatter_tiles, tile_buffer_pixels
from raytracer_amd import _abi as A
from raytracer_amd.api import tile_owner
from raytracer_amd import distributed as D


def synthetic_tiles(width, height, tw, th, rank, world):
    """What a rank's buffer must hold if pixel (x, y) has colour (x, y, x*y); padding stays 0."""
    tiles_x = (width + tw - 1) // tw
    tiles = owned_tiles(width, height, tw, th, rank, world)
    buf = np.zeros((len(tiles), th, tw, 3))
    for slot, t in enumerate(tiles):
        x0, y0 = (t % tiles_x) * tw, (t // tiles_x) * th
        for ly in range(min(th, height - y0)):
            for lx in range(min(tw, width - x0)):
                buf[slot, ly, lx] = (x0 + lx, y0 + ly, (x0 + lx) * (y0 + ly))
    return buf


def expected_frame(width, height):
    ys, xs = np.mgrid[0:height, 0:width]
    return np.stack([xs, ys, xs * ys], axis=-1).astype(np.float64)


@pytest.mark.parametrize("width,height,tile,world", [(50, 37, 16, 2), (64, 32, 16, 4), (33, 17, 8, 3), (20, 20, 32, 2),
                                                     (128, 70, 16, 8), (96, 50, 16, 2), (1920, 1080, 16, 8)])
def test_ownership_and_assemble_single_process(width, height, tile, world):
    tiles_x, tiles_y, n, slots = D.tile_counts(width, height, tile, tile, world)
    owned = [owned_tiles(width, height, tile, tile, r, world) for r in range(world)]
    assert sorted(t for o in owned for t in o) == list(range(n))          # a partition of the tiles
    lib = A.engine_lib()
    assert all(lib.mirt_tile_owner(t % tiles_x, t // tiles_x, tiles_x, world) == r == tile_owner(t % tiles_x, t // tiles_x, tiles_x, world)
               for r, o in enumerate(owned) for t in o)                   # the C ABI's rule
    if tiles_x % world == 0 and tiles_y > 1:
        # no rank owns whole columns: the owner of the first column's tile changes from row to row
        assert all(tile_owner(0, ty, tiles_x, world) != tile_owner(0, ty + 1, tiles_x, world) for ty in range(tiles_y - 1))
    assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    gathered = []
    frame = np.zeros((height, width, 3))
    for r in range(world):
        buf = synthetic_tiles(width, height, tile, tile, r, world)
        assert buf.size == tile_buffer_pixels(width, height, tile, tile, r, world) * 3  # the C ABI's own arithmetic
        scatter_tiles(frame, buf, width, height, tile, tile, r, world)                  # host-side de-interleave
        padded = np.zeros((slots, tile, tile, 3))
        padded[: buf.shape[0]] = buf
        gathered.append(padded.reshape(-1))
    assert np.array_equal(frame, expected_frame(width, height))
    dev = D.assemble(torch.from_numpy(np.concatenate(gathered)), width, height, tile, tile, world)
    assert np.array_equal(dev.numpy(), expected_frame(width, height))


def _worker(rank, world, port, width, height, tile, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = D.local_buffer(width, height, tile, tile, world, "cpu")
        mine = synthetic_tiles(width, height, tile, tile, rank, world).reshape(-1)
        local[: mine.size] = torch.from_numpy(mine)
        frame = D.gather_frame(local, width, height, tile, tile)
        at_root = D.gather_frame(local, width, height, tile, tile, dst=0)  # what bench.py does
        assert (at_root is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(at_root, frame)
        # the bench's max-over-ranks timing reduction
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rays = torch.tensor([100 * (rank + 1)], dtype=torch.int64)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        np.save(os.path.join(result_dir, f"frame{rank}.npy"), frame.numpy())
        np.save(os.path.join(result_dir, f"red{rank}.npy"), np.array([t.item(), rays.item()]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    width, height, tile, world = 70, 45, 16, 2
    mp.spawn(_worker, args=(world, port, width, height, tile, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"frame{r}.npy"), expected_frame(width, height))
        assert np.array_equal(np.load(tmp_path / f"red{r}.npy"), np.array([2.0, 300.0]))
