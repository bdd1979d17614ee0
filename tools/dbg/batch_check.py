import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mitsuba3dopplertof_amd as mi
xml = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1] if len(sys.argv) > 1 else "sweep_b0_it24.xml")).read()
offs = [0.0, 0.25, 0.5, 0.75]
for pipe in ("fused", "split"):
    os.environ["DTOF_PIPELINE"] = pipe
    sc = mi.load_string(xml)
    batch = sc.render(seed=5, spp=4, offsets=offs)
    for k, off in enumerate(offs):
        s1 = mi.load_string(xml.replace('<integrator type="dopplertofpath">', '<integrator type="dopplertofpath"><float name="hetero_offset" value="%s"/>' % off))
        single = np.asarray(s1.render(seed=5, spp=4))
        d = np.abs(np.asarray(batch[k]) - single)
        print(pipe, off, "max diff %.3e" % d.max(), "peak %.3e" % np.abs(single).max(), "bad pixels", int((d > 1e-5 * np.abs(single).max()).any(axis=-1).sum()))
