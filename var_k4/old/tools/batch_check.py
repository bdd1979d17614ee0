"""K = 4 offset batch against four single renders, both pipelines, for a scene file under scenes/ (development helper)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3dopplertof_amd as mi
os.chdir(os.path.join(ROOT, "scenes"))
xml = open(sys.argv[1]).read()
params = dict(resx=32, resy=32, max_depth=6)
offs = [0.0, 0.25, 0.5, 0.75]
worst = 0.0
for pipe in ("fused", "split"):
    os.environ["DTOF_PIPELINE"] = pipe
    sc = mi.load_string(xml, **params)
    batch = sc.render(seed=5, spp=8, offsets=offs)
    for k, off in enumerate(offs):
        s1 = mi.load_string(xml, hetero_offset=off, **params)
        single = np.asarray(s1.render(seed=5, spp=8))
        d = np.abs(np.asarray(batch[k]) - single)
        rel = d.max() / np.abs(single).max()
        worst = max(worst, rel)
        print(pipe, off, "max diff %.3e" % d.max(), "peak %.3e" % np.abs(single).max(), "bad pixels", int((d > 1e-5 * np.abs(single).max()).any(axis=-1).sum()))
sys.exit(1 if worst > 1e-5 else 0)
