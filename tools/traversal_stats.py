"""Traversal counters of a scene (development): per ray the node steps, leaf visits, triangle tests, and how many wave-level
iterations the 64 lanes of a wave needed for them (lane utilisation of the while-while loop).  Needs `make -C mitsuba3dopplertof_amd/csrc stats`.
usage: python tools/traversal_stats.py scene.xml [spp [key=value ...]]"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DTOF_LIB"] = os.path.join(HERE, "mitsuba3dopplertof_amd", "libdtof_stats.so")
os.environ.setdefault("DTOF_PIPELINE", "split")
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
path = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(HERE, "scenes", sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
params = dict(a.split("=") for a in sys.argv[3:])
sc = mi.load_file(path, **params)
out = (C.c_ulonglong * 8)()
L = mi._lib()
L.dtof_debug_traversal_stats(out)
sc.render(seed=0, spp=spp)
L.dtof_debug_traversal_stats(out)
rays, nl, nw, ll, lw, ml, tt, bl = [int(x) for x in out]
print("%s spp %d: %d rays (closest-hit + occlusion)" % (os.path.basename(path), spp, rays))
print("  TLAS node steps / ray %.1f   wave iterations / wave %.1f  -> lane utilisation %.2f" % (nl / rays, nw / (rays / 64), nl / max(nw * 64, 1)))
print("  leaf visits / ray %.2f       wave leaf rounds / wave %.1f -> lane utilisation %.2f" % (ll / rays, lw / (rays / 64), ll / max(lw * 64, 1)))
print("  mesh loops entered / ray %.2f, triangle tests / ray %.1f, BLAS node steps / ray %.1f" % (ml / rays, tt / rays, bl / rays))
