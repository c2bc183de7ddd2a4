"""One process per GPU: every rank renders the interleaved screen tiles it owns from its own
replica of the scene; the only exchange is the frame gather at the end (RCCL all-gather over xGMI
when the tensors are on GPUs; gloo on CPU tensors in the tests).

Tile (tx, ty) belongs to rank (ty * tiles_x + tx) % world, or to (tx + k * ty) % world, k = round(sqrt(world)),
when tiles_x is a multiple of world (mirt_tile_owner); a rank's buffer holds its tiles in row-major order, each tile as
tile_h rows of tile_w pixels (mirt_render_tiles).
"""
import torch
import torch.distributed as dist

TILE = 16  # pixels; equals the engine's thread-block footprint, the finest interleave that costs nothing


def tile_counts(width, height, tile_w, tile_h, world):
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    n = tiles_x * tiles_y
    return tiles_x, tiles_y, n, (n + world - 1) // world  # ..., most tiles any rank owns


def local_buffer(width, height, tile_w, tile_h, world, device, dtype=torch.float64):
    """Per-rank tile buffer, padded to the largest rank's size so that all ranks gather equal sizes."""
    _, _, _, slots = tile_counts(width, height, tile_w, tile_h, world)
    return torch.zeros(slots * tile_h * tile_w * 3, dtype=dtype, device=device)


def assemble(gathered, width, height, tile_w, tile_h, world):
    """[world * slots * tile_h * tile_w * 3] (rank-major) -> (height, width, 3) frame; tensor ops only."""
    tiles_x, tiles_y, n, slots = tile_counts(width, height, tile_w, tile_h, world)
    if world > 1 and tiles_x % world == 0:
        # staggered rows: rank r's slot (ty, j) is tile tx = (r - k * ty) % world + j * world of row ty
        from .api import tile_lattice_step
        k = tile_lattice_step(world)
        per_row = tiles_x // world
        g = gathered.view(world, tiles_y, per_row, tile_h, tile_w, 3)
        ty = torch.arange(tiles_y, device=gathered.device).view(-1, 1, 1)
        j = torch.arange(per_row, device=gathered.device).view(1, -1, 1)
        q = torch.arange(world, device=gathered.device).view(1, 1, -1)
        g = g[(q + k * ty) % world, ty, j]  # [ty, j, q] -> tile tx = j * world + q
    else:
        g = gathered.view(world, slots, tile_h, tile_w, 3)
        g = g.permute(1, 0, 2, 3, 4).reshape(slots * world, tile_h, tile_w, 3)[:n]  # tile order: t = slot*world + rank
    g = g.reshape(tiles_y, tiles_x, tile_h, tile_w, 3).permute(0, 2, 1, 3, 4)
    return g.reshape(tiles_y * tile_h, tiles_x * tile_w, 3)[:height, :width]


def gather_frame(local, width, height, tile_w, tile_h, group=None, dst=None):
    """Gathers the ranks' tile buffers and returns the assembled frame.
    dst=None: all-gather, the frame on every rank.  dst=r: gather to rank r only (the rank that writes
    the image), None elsewhere -- every other rank sends its buffer over its own xGMI link to r at
    the same time instead of passing everybody's around a ring."""
    world = dist.get_world_size(group)
    if dst is None:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return assemble(out, width, height, tile_w, tile_h, world)
    if dist.get_rank(group) == dst:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.view(world, -1).unbind(0)), dst=dst, group=group)
        return assemble(out, width, height, tile_w, tile_h, world)
    dist.gather(local, None, dst=dst, group=group)
    return None


def render_frame(scene, camera, config, width, height, spp, local, tile=TILE, group=None, dst=None):
    """Renders this rank's tiles into `local` (a CUDA tensor from local_buffer) and gathers the frame
    (see gather_frame for dst).  Returns (frame, stats of this rank)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    stream = torch.cuda.current_stream(local.device).cuda_stream
    stats = scene.render_tiles(camera, config, width, height, spp, tile, tile, rank, world, local.data_ptr(), stream)
    return gather_frame(local, width, height, tile, tile, group, dst), stats
