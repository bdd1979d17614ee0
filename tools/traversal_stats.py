"""Traversal counters of a scene (development): per ray the node steps, leaf visits, triangle tests, and how many wave-level
iterations the 64 lanes of a wave needed for them (lane utilisation of the while-while loop).  Needs `make -C mitsuba3dopplertof_amd/csrc stats`.
usage: python tools/traversal_stats.py scene.xml [spp [key=value ...]] [--json OUT.json] [--pipeline split|fused]"""
import ctypes as C, json, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = {}
it = iter(sys.argv[1:])
for a in it:
    if a.startswith("--"):
        opts[a[2:]] = next(it)
args = [a for a in args if a not in opts.values()]
os.environ["DTOF_LIB"] = os.path.join(HERE, "mitsuba3dopplertof_amd", "libdtof_stats.so")
os.environ["DTOF_PIPELINE"] = opts.get("pipeline", os.environ.get("DTOF_PIPELINE", "split"))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
path = args[0] if os.path.exists(args[0]) else os.path.join(HERE, "scenes", args[0])
spp = int(args[1]) if len(args) > 1 else 4
params = dict(a.split("=") for a in args[2:])
sc = mi.load_file(path, **params)
out = (C.c_ulonglong * 16)()
L = mi._lib()
L.dtof_debug_traversal_stats(out)
sc.render(seed=0, spp=spp)
L.dtof_debug_traversal_stats(out)
st = sc.last_stats
rays, nl, nw, ll, lw, ml, tt, bl, il, iw, mw, tw, rl, rw = [int(x) for x in out][:14]
print("%s spp %d: %d rays (closest-hit + occlusion), %d paths, %d path-bounces, %d shadow rays" % (os.path.basename(path), spp, rays, st["n_paths"], st["n_bounces"], st["n_shadow_rays"]))
print("  TLAS node steps / ray %.1f   wave iterations / wave %.1f  -> lane utilisation %.2f" % (nl / rays, nw / (rays / 64), nl / max(nw * 64, 1)))
print("  leaf visits / ray %.2f       wave leaf rounds / wave %.1f -> lane utilisation %.2f" % (ll / rays, lw / (rays / 64), ll / max(lw * 64, 1)))
print("  mesh loops entered / ray %.2f, triangle tests / ray %.1f, BLAS node steps / ray %.1f" % (ml / rays, tt / rays, bl / rays))
bw = int(out[15])
if bw: print("  BLAS node steps: %.1f wave-level executions per wave of 64 rays, lane utilisation %.3f" % (bw / (rays / 64), bl / (bw * 64)))
W = rays / 64
print("  per wave of 64 rays, wave-level executions (lane utilisation): node steps %.1f (%.2f) | leaf rounds %.2f (%.2f) | rectangle tests %.2f (%.2f) | instance transforms %.2f (%.2f) | mesh loops %.2f (%.2f) | triangle tests %.1f (%.2f)" % (
    nw / W, nl / max(nw * 64, 1), lw / W, ll / max(lw * 64, 1), rw / W, rl / max(rw * 64, 1), iw / W, il / max(iw * 64, 1), mw / W, ml / max(mw * 64, 1), tw / W, tt / max(tw * 64, 1)))
if "json" in opts:
    json.dump({"scene": os.path.basename(path), "spp": spp, "params": params, "pipeline": os.environ["DTOF_PIPELINE"], "paths": st["n_paths"], "path_bounces": st["n_bounces"],
               "shadow_rays": st["n_shadow_rays"], "rays": rays, "rays_per_path": rays / st["n_paths"], "tlas_node_steps_per_ray": nl / rays, "leaf_visits_per_ray": ll / rays,
               "mesh_loops_per_ray": ml / rays, "triangle_tests_per_ray": tt / rays, "blas_node_steps_per_ray": bl / rays,
               "node_phase_lane_utilisation": nl / max(nw * 64, 1), "leaf_phase_lane_utilisation": ll / max(lw * 64, 1),
               "wave_level_per_64_rays": {"node_steps": nw / W, "leaf_rounds": lw / W, "rectangle_tests": rw / W, "instance_transforms": iw / W, "mesh_loops": mw / W, "triangle_tests": tw / W},
               "lane_utilisation": {"node_steps": nl / max(nw * 64, 1), "leaf_rounds": ll / max(lw * 64, 1), "rectangle_tests": rl / max(rw * 64, 1), "instance_transforms": il / max(iw * 64, 1),
                                    "mesh_loops": ml / max(mw * 64, 1), "triangle_tests": tt / max(tw * 64, 1)}}, open(opts["json"], "w"), indent=1)
