import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mitsuba3dopplertof_amd as mi
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sc = mi.load_file(R + "/scenes/cornell_wall.xml"); W, H = sc.size
film = torch.zeros((H + 2, W, 4), device="cuda"); rgb = torch.zeros((H, W, 3), device="cuda")
ptr = film.data_ptr() + W * 16; lib = mi._lib()
import ctypes as C
def run(n, stats=True):
    acc = [0.0] * 5
    for _ in range(n):
        t0 = time.perf_counter(); film.zero_(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if stats: sc.render_rows(ptr, 0, 64, 0, H)
        else: mi._check(lib.dtof_render_rows(sc._h, 0, 64, 0, H, None, 0, ptr, None))
        t3 = time.perf_counter(); full = film[1:1 + H].contiguous(); lib.dtof_develop(full.data_ptr(), rgb.data_ptr(), H * W); t4 = time.perf_counter()
        torch.cuda.synchronize(); t5 = time.perf_counter()
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += d
    return [a / n * 1e3 for a in acc]
run(5)
print("with stats   : zero %.3f sync %.3f render %.3f develop %.3f sync %.3f ms" % tuple(run(30)), "gpu total", sc.last_stats["ms_total"])
print("without stats: zero %.3f sync %.3f render %.3f develop %.3f sync %.3f ms" % tuple(run(30, False)))
