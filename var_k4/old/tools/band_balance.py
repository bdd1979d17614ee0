#!/usr/bin/env python3
"""Load balance of the row-band sharding: time of each of N contiguous bands vs N sets of interleaved stripes, on ONE GPU.
    python tools/band_balance.py [scene.xml [N [stripe_rows]]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
import torch
import mitsuba3dopplertof_amd as mi
scene = sys.argv[1] if len(sys.argv) > 1 else "domino.xml"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stripe = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc = mi.load_file(os.path.join(ROOT, "scenes", scene))
W, H = sc.size
film = torch.zeros((H + 4, W, 4), dtype=torch.float32, device="cuda")
ptr = film.data_ptr() + 2 * W * 16
def t_rows(bands):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for a, b in bands: sc.render_rows(ptr, seed=0, spp=0, row_begin=a, row_end=b)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
sc.render_rows(ptr, seed=0, spp=0, row_begin=0, row_end=H)
full = t_rows([(0, H)])
n = (H + N - 1) // N
cont = [t_rows([(r * n, min(H, (r + 1) * n))]) for r in range(N)]
from mitsuba3dopplertof_amd import distributed as D
def t_stripes(r):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        sc.render_stripes(ptr, 0, 0, *D.stripe_layout(N, r, stripe))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
inter = [t_stripes(r) for r in range(N)]
print("%s %dx%d full %.2f ms | %d contiguous bands: max %.2f mean %.2f -> efficiency %.3f | %d-row stripes: max %.2f mean %.2f -> efficiency %.3f (vs full/N %.2f)" % (
    scene, W, H, full, N, max(cont), sum(cont) / N, full / N / max(cont), stripe, max(inter), sum(inter) / N, full / N / max(inter), full / N))
print("contiguous", " ".join("%.2f" % c for c in cont)); print("stripes   ", " ".join("%.2f" % c for c in inter))
