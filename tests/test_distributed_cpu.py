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


def _worker8(rank, world, port, q, mode):
    """world-8 rehearsal of both exchanges before the first 8-GPU run: a frame whose height divides by neither the world size nor world x stripe rows.
    bands  : default gaussian filter (halo 2), bands of ceil(27 / 8) = 4 rows -- the last rank's band is EMPTY, the one before it is short; one gather, overlap-add.
    stripes: K = 4 hetero_offset films, interleaved 2-row stripes (period 16; 27 rows = one full period + 11), one reduce(sum) of the [4, H, W, 4] films."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mitsuba3dopplertof_amd import distributed as D
    from oracle import orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    H, W, spp = 27, 12, 2
    text = open(os.path.join(SCENES, "cornell_wall.xml")).read()
    if mode == "bands":
        text = text.replace('<rfilter type="tent" />', "")
    sc = orc.Scene(text, dict(resx=W, resy=H), is_string=True)
    assert sc.size == (W, H)
    if mode == "bands":
        pd, halo = sc.params(), 2
        r0, r1 = D.row_band(H, world, rank)
        assert (r1 - r0) == (4 if rank < 6 else 3 if rank == 6 else 0)
        padded = np.zeros((D.padded_rows(H, world, halo), W, 4), np.float32)
        if r1 > r0:
            padded[halo:halo + H] = sc.render(pd, seed=5, spp=spp, rows=(r0, r1), raw=True)[0]
        p0, p1 = D.slab_range(H, world, rank, halo)
        assert np.count_nonzero(padded[:p0]) == 0 and np.count_nonzero(padded[p1:]) == 0
        stack = D.gather_film_stacked(torch.from_numpy(np.ascontiguousarray(padded[p0:p1])), rank, world)
        if rank == 0:
            full = D.overlap_add_stacked(stack, H, world, halo).numpy()
            assert np.array_equal(full, D.overlap_add([stack[r] for r in range(world)], H, world, halo, xp=torch).numpy())
            ref = sc.render(pd, seed=5, spp=spp, raw=True)[0]
            q.put(float(np.abs(full - ref).max() / np.abs(ref).max()))
    else:
        offsets, stripe = [0.0, 0.25, 0.5, 0.75], 2
        rows = D.stripe_rows_of(H, world, rank, stripe)
        assert rows == [y for y in range(H) if (y // stripe) % world == rank]
        film = np.zeros((len(offsets), H, W, 4), np.float32)
        pds = [sc.params(integrator=dict(type="dopplertofpath", max_depth=4, path_correlation_depth=4, hetero_frequency=1.0, hetero_offset=o, time_sampling_method="stratified")) for o in offsets]
        for k, pd in enumerate(pds):
            for y in rows:                                        # the oracle renders a row range; a rank's stripes are the sum of its rows' films
                film[k] += sc.render(pd, seed=5, spp=spp, rows=(y, y + 1), raw=True)[0]
        total = D.reduce_film(torch.from_numpy(film), rank, world)
        if rank == 0:
            ref = np.stack([sc.render(pd, seed=5, spp=spp, raw=True)[0] for pd in pds])
            assert np.abs(ref[0] - ref[2]).max() > 1e-3 * np.abs(ref).max()      # the four films differ (offset 0 vs 0.5 flips the sign)
            q.put(float(np.abs(total.numpy() - ref).max() / np.abs(ref).max()))
        else:
            assert total is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["bands", "stripes"], ids=["bands_gather_halo2_empty_last_band", "stripes_reduce_four_films"])
def test_eight_rank_gloo_exchanges_reproduce_the_single_rank_films(mode):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q, mode)) for r in range(8)]
    for p in procs:
        p.start()
    try:
        err = q.get(timeout=400)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert err < 1e-6


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
