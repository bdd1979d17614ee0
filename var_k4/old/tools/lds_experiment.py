"""How much does LDS residency of the scene buy on an instanced scene?  A 12 x 12 Domino field (145 objects, blob < 48 KiB) rendered with
the whole blob staged into LDS (default) and with DTOF_STAGE=0 (every node / object / triangle read through L1 / L2), split pipeline both.
usage: python tools/lds_experiment.py [n_side [res [spp]]]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(HERE, "scenes")); sys.path.insert(0, HERE)
import make_scenes
n_side, res, spp = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 12), (2, 1024), (3, 32)))
path = os.path.join(HERE, "gpurun_out", "domino_%d.xml" % n_side)
os.makedirs(os.path.dirname(path), exist_ok=True)
open(path, "w").write(make_scenes.domino(n_side=n_side, res=res, spp=spp))
code = ("import sys, time; sys.path.insert(0, %r); import mitsuba3dopplertof_amd as mi\n"
        "sc = mi.load_file(%r); print(sc.info()['scene_blob_bytes'], sc.info()['n_objects'], sc.info()['n_bvh_nodes'])\n"
        "for k in range(4): sc.render(seed=k); st = sc.last_stats\n"
        "print({k: round(st[k], 3) for k in ('ms_total', 'ms_trace', 'ms_shade', 'ms_shadow')})\n" % (HERE, path))
for env in ({}, {"DTOF_STAGE": "0"}, {"DTOF_STAGE": "0", "DTOF_TRACE_BLOCK": "64"}):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DTOF_PIPELINE="split", **env), capture_output=True, text=True)
    print(env, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-300:])
