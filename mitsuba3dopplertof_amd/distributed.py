"""Pixel-row sharding of one render across the GPUs of a node and the single film gather.

The reference has no distributed layer; the partition is sound because every lane's RNG
streams are pure functions of its global lane index (src/render/sampler.cpp:115-134,
src/samplers/correlated.cpp:38-64) and lanes of one pixel are contiguous
(src/render/integrator.cpp:273-285), so any pixel partition reproduces the single-device image up to
the (already unordered) float accumulation order of the splat.

Rank r renders crop rows [r0, r1).  With a reconstruction filter of footprint radius `halo` pixels
(tent r=1 -> 1) its splats touch rows [r0-halo, r1+halo), so each rank keeps a zero-padded slab of
(rows_per_rank + 2*halo) rows; ONE gather (RCCL over xGMI: torch.distributed backend "nccl") brings the
slabs to rank 0, which overlap-adds them and develops RGB/W.
"""
import numpy as np


def rows_per_rank(height, world):
    return (height + world - 1) // world


def row_band(height, world, rank):
    """[r0, r1) of `rank`; trailing ranks may get a shorter (or empty) band."""
    n = rows_per_rank(height, world)
    r0 = min(rank * n, height)
    return r0, min(r0 + n, height)


def padded_rows(height, world, halo):
    """rows of the padded film every rank allocates: film row y lives at padded row y + halo"""
    return rows_per_rank(height, world) * world + 2 * halo


def slab_range(height, world, rank, halo):
    """padded-row range [p0, p1) that holds everything `rank` splats; the same size on every rank"""
    n = rows_per_rank(height, world)
    p0 = rank * n
    return p0, p0 + n + 2 * halo


def overlap_add(slabs, height, world, halo, xp=np):
    """slabs[r]: (rows_per_rank + 2*halo, W, C) -> full film (height, W, C).  `xp` = numpy or torch."""
    n = rows_per_rank(height, world)
    s0 = slabs[0]
    if xp is np:
        out = np.zeros((n * world + 2 * halo,) + tuple(s0.shape[1:]), dtype=s0.dtype)
    else:
        out = xp.zeros((n * world + 2 * halo,) + tuple(s0.shape[1:]), dtype=s0.dtype, device=s0.device)
    for r, s in enumerate(slabs):
        p0, p1 = slab_range(height, world, r, halo)
        out[p0:p1] += s
    return out[halo:halo + height]


def gather_film(slab, rank, world, group=None):
    """The one collective of a frame: gather equal-sized slabs on rank 0 (torch.distributed)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [slab]
    if slab.is_cuda and dist.get_backend(group) == "gloo":   # test set-ups without RCCL: stage through host memory
        host = slab.cpu()
        dst = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, dst, dst=0, group=group)
        return [d.to(slab.device) for d in dst] if rank == 0 else None
    dst = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
    dist.gather(slab, dst, dst=0, group=group)
    return dst


def overlap_add_stacked(stack, height, world, halo):
    """overlap_add for slabs stacked as ONE tensor [world, rows_per_rank + 2*halo, W, C] (torch): the bands are copied out
    with one reshape and the halo rows of every neighbour pair are added with two sliced adds -- three device operations
    whatever the number of ranks, instead of one per rank."""
    n = rows_per_rank(height, world)
    if halo == 0:
        return stack.reshape((world * n,) + tuple(stack.shape[2:]))[:height]
    if halo > n:      # bands thinner than the filter footprint: fall back to the general loop
        return overlap_add([stack[r] for r in range(world)], height, world, halo, xp=__import__("torch"))
    full = stack[:, halo:halo + n].clone()                        # [world, n, W, C]: every band's own rows
    full[1:, :halo] += stack[:-1, halo + n:]                      # rows a rank splatted below its band -> next band's head
    full[:-1, n - halo:] += stack[1:, :halo]                      # rows a rank splatted above its band -> previous band's tail
    return full.reshape((world * n,) + tuple(stack.shape[2:]))[:height]


def gather_film_stacked(slab, rank, world, group=None, out=None, force=False):
    """gather_film into one preallocated [world, ...] tensor on rank 0 (returned; None on the other ranks).  force: issue the collective for one rank as well."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return slab.unsqueeze(0)
    if slab.is_cuda and dist.get_backend(group) == "gloo":
        parts = gather_film(slab, rank, world, group)
        return torch.stack(parts) if rank == 0 else None
    if rank == 0:
        if out is None:
            out = torch.empty((world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
        dist.gather(slab, [out[r] for r in range(world)], dst=0, group=group)
        return out
    dist.gather(slab, None, dst=0, group=group)
    return None


def stripe_layout(world, rank, stripe_rows):
    """(first_row, stripe_rows, stripe_period) of `rank`: stripes of `stripe_rows` rows dealt round-robin to the ranks"""
    return rank * stripe_rows, stripe_rows, world * stripe_rows


def stripe_rows_of(height, world, rank, stripe_rows):
    """the film rows `rank` renders under stripe_layout, ascending (host-side mirror of the library's mapping; tests, oracle)"""
    first, rows, period = stripe_layout(world, rank, stripe_rows)
    return [y for y in range(first, height) if (y - first) % period < rows]


def reduce_film(film, rank, world, group=None):
    """The one collective of a striped frame: reduce(sum) of the ranks' full-size films -- [planes, H, W, 4]: K offset films, + the alpha film of an rgba scene -- to rank 0
    (RCCL over xGMI; gloo test set-ups stage device tensors through host memory).  Returns the summed film on rank 0; the other ranks' buffers are scratch afterwards."""
    import torch.distributed as dist
    if world == 1:
        return film
    if film.is_cuda and dist.get_backend(group) == "gloo":
        host = film.cpu()
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM, group=group)
        return host.to(film.device) if rank == 0 else None
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM, group=group)
    return film if rank == 0 else None


def render_striped(scene, seed=0, spp=0, stripe_rows=16, group=None):
    """One frame across the ranks with INTERLEAVED stripes of pixel rows ("interleaved row bands ... if Domino is spatially
    unbalanced"): rows that see only sky cost a tenth of rows full of dominoes, so contiguous bands leave ranks idle (measured on
    one GPU, 8 bands of Domino: 0.75 efficiency; stripes of 16-32 rows: ranks within 3 % of each other).  Every rank accumulates
    its stripes into a zeroed full-size film and ONE reduce(sum) to rank 0 (RCCL; 16 MB at 1024 x 1024) replaces gather +
    overlap-add.  Returns the (H, W, 3) image -- (H, W, 4) for an rgba film -- on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    W, H = scene.size
    dev = torch.device("cuda", torch.cuda.current_device())
    planes = scene.film_planes()                  # 2 for an rgba film: the alpha film is one more RGBW plane behind the colour film
    film = torch.zeros((planes, H, W, 4), dtype=torch.float32, device=dev)
    scene.set_film_layout(planes)
    torch.cuda.synchronize()
    first, rows, period = stripe_layout(world, rank, stripe_rows)
    scene.render_stripes(film.data_ptr(), seed, spp, first, rows, period)
    film = reduce_film(film, rank, world, group)
    if rank != 0:
        return None
    return _develop(film, planes, H, W, dev)


def _develop(film, planes, H, W, dev):
    """HDRFilm::develop of a [planes, H, W, 4] device film: (H, W, 3), or (H, W, 4) when the second plane is the alpha film of an rgba scene"""
    import torch
    from . import _check, _lib
    rgb = torch.zeros((H, W, 4 if planes == 2 else 3), dtype=torch.float32, device=dev)
    if planes == 2:
        _check(_lib().dtof_develop_rgba(film[0].data_ptr(), film[1].data_ptr(), rgb.data_ptr(), H * W))
    else:
        _check(_lib().dtof_develop(film.data_ptr(), rgb.data_ptr(), H * W))
    return rgb.cpu().numpy()


def render_sharded(scene, seed=0, spp=0, halo=None, group=None):
    """One frame across the ranks of an initialised torch.distributed job (one process per GPU, backend "nccl" = RCCL): every
    rank renders its band of pixel rows into a zero-padded device slab (dtof_render_rows), ONE gather brings the slabs to
    rank 0, which overlap-adds the shared halo rows and develops RGB / W.  Returns the (H, W, 3) image -- (H, W, 4) for an rgba film -- on rank 0, None elsewhere.
    World size 1 (or no process group) renders the whole frame on the current device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    W, H = scene.size
    need = int(scene.info()["filter_halo"])       # ceil(radius - 0.5): tent r=1 -> 1, the default gaussian (radius 2) -> 2
    halo = need if halo is None else halo
    if halo < need:
        raise ValueError("halo of %d rows is smaller than the reconstruction filter's reach of %d rows: splats across the band "
                         "seams would be dropped" % (halo, need))
    dev = torch.device("cuda", torch.cuda.current_device())
    r0, r1 = row_band(H, world, rank)
    planes, rows = scene.film_planes(), padded_rows(H, world, halo)
    film = torch.zeros((planes, rows, W, 4), dtype=torch.float32, device=dev)
    scene.set_film_layout(planes, rows * W * 4)   # the planes of the padded film lie a whole padded film apart
    torch.cuda.synchronize()
    scene.render_rows(film.data_ptr() + halo * W * 4 * 4, seed=seed, spp=spp, row_begin=r0, row_end=r1)
    p0, p1 = slab_range(H, world, rank, halo)
    # a band of ALL planes as one slab of rows, [rows, planes * W, 4]: one gather moves it and the overlap-add works on rows
    slab = film[:, p0:p1].permute(1, 0, 2, 3).reshape(p1 - p0, planes * W, 4).contiguous()
    stack = gather_film_stacked(slab, rank, world, group)
    if rank != 0:
        return None
    full = overlap_add_stacked(stack, H, world, halo) if world > 1 else slab[halo:halo + H]
    full = full.reshape(H, planes, W, 4).permute(1, 0, 2, 3).contiguous()
    return _develop(full, planes, H, W, dev)
