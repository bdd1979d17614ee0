"""The N>1 path on CPU: two gloo ranks shard the pixel rows, each produces the raw film of its band (here with
the CPU oracle standing in for the GPU renderer -- the sharding / gather / overlap-add code under test is the
product's mitsuba3dopplertof_amd.distributed, the same code bench.py runs over RCCL) and rank 0 reassembles."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES


def test_row_band_bookkeeping():
    from mitsuba3dopplertof_amd import distributed as D
    for H, world in [(512, 8), (30, 4), (7, 3), (5, 8), (1024, 1)]:
        bands = [D.row_band(H, world, r) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == H
        assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        sizes = {D.slab_range(H, world, r, 1)[1] - D.slab_range(H, world, r, 1)[0] for r in range(world)}
        assert len(sizes) == 1
        assert D.slab_range(H, world, world - 1, 1)[1] <= D.padded_rows(H, world, 1)
    slabs = [np.ones((4, 3, 2), np.float32), np.ones((4, 3, 2), np.float32)]   # H=4, world=2, halo=1: bands of 2 rows + 2 halo rows
    out = D.overlap_add(slabs, 4, 2, 1)
    assert out.shape == (4, 3, 2) and out[:, 0, 0].tolist() == [1, 2, 2, 1]


def _worker(rank, world, port, q, gaussian=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mitsuba3dopplertof_amd import distributed as D
    from oracle import orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    H = W = 20
    text = open(os.path.join(SCENES, "cornell_wall.xml")).read()
    if gaussian:     # a film WITHOUT an rfilter child: the default gaussian, radius 2 -> splats reach two rows beyond their pixel
        text = text.replace('<rfilter type="tent" />', "")
    sc = orc.Scene(text, dict(resx=W, resy=H), is_string=True)
    pd = sc.params()
    halo = int(np.ceil(sc.flat.sensor["filter_radius"] - 0.5))                 # what dtof_scene_info.filter_halo reports
    assert halo == (2 if gaussian else 1)
    r0, r1 = D.row_band(H, world, rank)
    band, _ = sc.render(pd, seed=5, spp=4, rows=(r0, r1), raw=True)            # (H, W, 4) with only rows r0-halo..r1+halo touched
    padded = np.zeros((D.padded_rows(H, world, halo), W, 4), np.float32)
    padded[halo:halo + H] = band
    p0, p1 = D.slab_range(H, world, rank, halo)
    assert np.count_nonzero(padded[:p0]) == 0 and np.count_nonzero(padded[p1:]) == 0   # a rank only writes inside its slab
    mine = torch.from_numpy(np.ascontiguousarray(padded[p0:p1]))
    slabs = D.gather_film(mine, rank, world)
    stack = D.gather_film_stacked(mine, rank, world)
    if rank == 0:
        full = D.overlap_add(slabs, H, world, halo, xp=torch).numpy()
        assert np.array_equal(D.overlap_add_stacked(stack, H, world, halo).numpy(), full)
        ref, _ = sc.render(pd, seed=5, spp=4, raw=True)
        q.put(float(np.abs(full - ref).max() / np.abs(ref).max()))
    dist.barrier()
    dist.destroy_process_group()


def test_stripe_bookkeeping():
    """interleaved shards: the stripes of all ranks partition the rows, whatever the frame height"""
    from mitsuba3dopplertof_amd import distributed as D
    for height, world, stripe in [(48, 3, 5), (37, 4, 4), (50, 8, 3), (1024, 8, 32), (7, 8, 16), (1, 2, 1)]:
        rows = [D.stripe_rows_of(height, world, r, stripe) for r in range(world)]
        assert sorted(y for rr in rows for y in rr) == list(range(height))
        first, n, period = D.stripe_layout(world, 1 % world, stripe)
        assert period == world * stripe and n == stripe and first == (1 % world) * stripe
        assert max(len(rr) for rr in rows) - min(len(rr) for rr in rows) <= stripe


@pytest.mark.parametrize("gaussian", [False, True], ids=["tent_halo1", "default_gaussian_halo2"])
def test_two_rank_gloo_film_gather_reproduces_the_single_rank_film(gaussian):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, gaussian)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-6
