#!/usr/bin/env python3
"""A/B timing of one path-trace pass on the larger scenes under environment knobs of the library (read at every launch), variants
interleaved in ONE process, accumulation images compared bit for bit with the first variant's.
python tools/ab_bigscene_render.py [atrium|tetra] [spp] "NAME=ENV1=V1;ENV2=V2,NAME2=..." [tetra_tris]   e.g.  "base=,pool=PTGPU_PT_POOL=1" """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    variants = []
    for item in (sys.argv[3] if len(sys.argv) > 3 else "base=,pool=PTGPU_PT_POOL=1").split(","):
        name, _, envs = item.partition("=")
        variants.append((name, dict(e.split("=", 1) for e in envs.split(";") if e)))
    if which == "atrium":
        mesh, view, W, H, dof = scene.atrium_scene(), scene.atrium_view, 3840, 2160, False
    else:
        mesh, view, W, H, dof = scene.tetra_cloud_scene(int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000), scene.tetra_cloud_view, 1920, 1080, True
    nodes, lst, st = hostapi.bvh_build(mesh["pos"], True)
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
              materials=mesh["materials"], textures=[], nodes=nodes, list=lst)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    sc = capi.Scene(ctx, sd)
    cam, lights = view(W, H)
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    keys = sorted({k for _, e in variants for k in e})
    times = {n: [] for n, _ in variants}
    ref = None
    for r in range(4):
        for name, envs in variants:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(envs)
            prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=True, depth_of_field=dof, aperture=0.01,
                                    focal_distance=1.2, termination_prob=0.2, pass_seed=5)
            accum.zero_()
            ev[0].record(stream)
            sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
            ev[1].record(stream)
            torch.cuda.synchronize()
            ctx.spill_status()
            if r == 0:
                if ref is None:
                    ref = accum.clone()
                else:
                    assert torch.equal(accum.view(torch.int32), ref.view(torch.int32)), "%s: image differs from %s's" % (name, variants[0][0])
            else:
                times[name].append(ev[0].elapsed_time(ev[1]))
    print("%s: %d triangles, %dx%d, %d spp" % (which, mesh["pos"].shape[0], W, H, spp))
    for name, _ in variants:
        ms = float(np.median(times[name]))
        print("%-14s %9.2f ms per pass   %7.1f Msamples/s" % (name, ms, W * H * spp / ms / 1e3), flush=True)


if __name__ == "__main__":
    main()
