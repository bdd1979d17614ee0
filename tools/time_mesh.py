#!/usr/bin/env python3
"""Times the mesh workload: the Cornell room with two procedural blobs (static ply + moving obj-like ply) of
2*n_u*(n_v-1) triangles each.   python tools/time_mesh.py [n_u n_v res spp]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_mesh
import mitsuba3dopplertof_amd as mi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")); import make_scenes; make_scenes.ensure()

n_u, n_v, res, spp = [int(x) for x in (sys.argv[1:5] + ["512", "256", "512", "64"][len(sys.argv) - 1:])]
d = os.environ.get("DTOF_MESH_DIR") or tempfile.mkdtemp(prefix="dtof_mesh_")   # DTOF_MESH_DIR: keep the scene (tools/traversal_stats.py $DTOF_MESH_DIR/s.xml 64)
os.makedirs(d, exist_ok=True)
t = time.time()
pos, nrm, uv, faces = make_mesh.blob(n_u, n_v)
make_mesh.write_ply(os.path.join(d, "blob.ply"), pos, nrm, uv, faces)
make_mesh.write_ply(os.path.join(d, "blob2.ply"), pos, nrm, uv, faces, with_normals=False)
xml = make_mesh.cornell_mesh_xml(moving_file="blob2.ply", res=res, spp=spp).replace('<shape type="obj" id="MovingBlob">', '<shape type="ply" id="MovingBlob">')
open(os.path.join(d, "s.xml"), "w").write(xml)
t_gen = time.time() - t
t = time.time(); sc = mi.load_file(os.path.join(d, "s.xml")); t_load = time.time() - t
info = sc.info()
best = None
for i in range(5):
    t = time.time(); img = sc.render(seed=0, spp=0); dt = time.time() - t
    st = sc.last_stats
    if best is None or st["ms_total"] < best["ms_total"]: best = dict(st, wall=dt * 1e3)
loop = best["ms_trace"] + best["ms_shade"] + best["ms_shadow"]
print("mesh %dx%d: %d tris, %d nodes, blob %.1f MB | gen %.1fs load+build %.2fs | %dx%dx%d: total %.2f ms Mpaths/s %.0f | gen %.2f trace %.2f shade %.2f shadow %.2f splat %.2f | bounces %d shadow rays %d -> %.0f Mrays/s" % (
    n_u, n_v, info["n_triangles"], info["n_bvh_nodes"], info["scene_blob_bytes"] / 1e6, t_gen, t_load, res, res, spp, best["ms_total"],
    best["n_paths"] / best["ms_total"] / 1e3, best["ms_generate"], best["ms_trace"], best["ms_shade"], best["ms_shadow"], best["ms_splat"],
    best["n_bounces"], best["n_shadow_rays"], (best["n_bounces"] + best["n_shadow_rays"]) / ((best["ms_trace"] + best["ms_shadow"]) or best["ms_shade"]) / 1e3))
