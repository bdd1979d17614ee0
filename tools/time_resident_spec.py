#!/usr/bin/env python3
"""Resident first-bounce stage on an every-BSDF Domino field (the scene of test_resident_stage_with_the_every_bsdf_kernels, larger): frame time of K = 1 and K = 4 renders
for DTOF_RESIDENT = 0 / 8 / 12 / 16 (development helper)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
SCENES = os.path.join(ROOT, "scenes")
xml = make_scenes.domino(n_side=14, res=512, spp=64)
ground = ('<bsdf type="twosided" id="GroundBSDF"><bsdf type="diffuse"><texture type="checkerboard" name="reflectance"><rgb name="color0" value="0.7, 0.6, 0.5"/><rgb name="color1" value="0.2, 0.3, 0.4"/>'
          '<transform name="to_uv"><scale x="6" y="6"/></transform></texture></bsdf></bsdf>')
domino = ('<bsdf type="mask" id="DominoBSDF"><float name="opacity" value="0.9"/><bsdf type="twosided"><bsdf type="roughplastic"><string name="distribution" value="ggx"/>'
          '<float name="alpha" value="0.2"/><rgb name="diffuse_reflectance" value="0.75, 0.55, 0.35"/></bsdf></bsdf></bsdf>')
xml = xml.replace(make_scenes.bsdf("GroundBSDF", "0.6, 0.6, 0.6"), ground + "\n").replace(make_scenes.bsdf("DominoBSDF", "0.75, 0.55, 0.35"), domino + "\n")
path = os.path.join(SCENES, "_domino_spec_timing.xml")
open(path, "w").write(xml)
try:
    os.environ["DTOF_PIPELINE"] = "fused"
    for res_waves in ("0", "8", "12", "16"):
        os.environ["DTOF_RESIDENT"] = res_waves
        sc = mi.load_file(path, max_depth=4)
        out = []
        for offsets in (None, [0.0, 0.25, 0.5, 0.75]):
            t = []
            for i in range(4):
                sc.render(seed=0, spp=64, **({"offsets": offsets} if offsets else {})); t.append(sc.last_stats["ms_total"])
            out.append(min(t[1:]))
        print("DTOF_RESIDENT=%-3s K = 1: %8.3f ms   K = 4: %8.3f ms" % (res_waves, out[0], out[1]), flush=True)
finally:
    os.remove(path)
