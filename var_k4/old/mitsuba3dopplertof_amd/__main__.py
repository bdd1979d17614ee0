"""Command line front end mirroring `mitsuba scene.xml -D key=value -o out` (src/mitsuba/mitsuba.cpp:150-423) for the
plugins this library implements:

    python -m mitsuba3dopplertof_amd scene.xml [-D key=value ...] [-o out.exr|.npy|.pfm] [--spp N] [--seed S]
                                               [--offsets 0,0.25,0.5,0.75] [-v]
    python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 -m mitsuba3dopplertof_amd scene.xml ...
        one process per GPU: the pixel rows are sharded across the G ranks, rank 0 gathers and writes the image
"""
import argparse
import os
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m mitsuba3dopplertof_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("scene")
    ap.add_argument("-D", "--define", action="append", default=[], metavar="key=value",
                    help="define a constant that scene files reference as $key (mitsuba.cpp:241-248)")
    ap.add_argument("-o", "--output", default=None, help="output file (.exr, .npy or .pfm); default: <scene>.exr")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (0 = the sampler's sample_count)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--offsets", default=None, help="comma separated hetero_offset values evaluated in ONE traversal")
    ap.add_argument("--stripes", type=int, default=0, metavar="ROWS",
                    help="multi-GPU runs: interleave stripes of ROWS pixel rows across the ranks (load balance) instead of one contiguous band per rank")
    ap.add_argument("-m", "--mode", default="hip_rgb", help="accepted for command-line compatibility (only hip_rgb exists)")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args(argv)
    import mitsuba3dopplertof_amd as mi
    from mitsuba3dopplertof_amd.io import write_image
    defines = {}
    for d in args.define:
        if "=" not in d:
            ap.error("-D expects key=value")
        k, v = d.split("=", 1)
        defines[k] = v
    try:
        scene = mi.load_file(args.scene, **defines)
        t0 = time.time()
        offsets = [float(x) for x in args.offsets.split(",")] if args.offsets else None
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:   # launched by torch.distributed.run: shard the rows, one film gather (distributed.py)
            import torch
            import torch.distributed as dist
            from mitsuba3dopplertof_amd import distributed as D
            if offsets is not None:
                ap.error("--offsets is a single-GPU feature")
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl")
            img = (D.render_striped(scene, seed=args.seed, spp=args.spp, stripe_rows=args.stripes) if args.stripes > 0
                   else D.render_sharded(scene, seed=args.seed, spp=args.spp))
            dist.barrier()
            dist.destroy_process_group()
            if img is None:
                return 0
        else:
            img = scene.render(seed=args.seed, spp=args.spp, offsets=offsets)
        dt = time.time() - t0
    except mi.DtofError as e:
        print("Error: %s" % e, file=sys.stderr)
        return 1
    out = args.output or os.path.splitext(args.scene)[0] + ".exr"
    if offsets is None:
        write_image(out, img)
    else:
        base, ext = os.path.splitext(out)
        for k, off in enumerate(offsets):
            write_image("%s_offset_%.3f%s" % (base, off, ext), img[k])
    if args.verbose:
        st = scene.last_stats
        print("Rendering finished. (took %.1f ms, %.0f Mpaths/s on the GPU: %s)" % (dt * 1e3, st["n_paths"] / max(st["ms_total"], 1e-9) / 1e3, st))
    return 0


if __name__ == "__main__":
    sys.exit(main())
