#!/usr/bin/env python3
"""Timing of the path-trace pass (ptgpu_render_pass_dev) on head.obj, variants interleaved in ONE process.
Run on the GPU box:  python tools/tune_render.py [spp] [rounds] [WxH]
Prints ms per pass, Msamples/s and Mrays/s per (AA mode, service threshold, oversubscription)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    W, H = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1920x1080").split("x"))
    variants = [tuple(int(x) for x in v.split(",")) for v in (sys.argv[4].split(";") if len(sys.argv) > 4 else ["24,1", "16,1", "32,1", "48,1", "24,2"])]
    mesh = scene.load_ptmesh()
    nodes, lst, _ = hostapi.bvh_build(mesh["pos"], True)
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
              materials=scene.mesh_materials(mesh), textures=[], nodes=nodes, list=lst)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    sc = capi.Scene(ctx, sd)
    cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = {}
    for r in range(rounds + 1):
        for sobol_block in (True, False):
            for (thr, ov) in variants:
                os.environ["PTGPU_PT_SERVICE"], os.environ["PTGPU_PT_OVERSUB"] = str(thr), str(ov)
                prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=sobol_block, filter_width=1.5, pass_seed=r)
                ctx.reset_counters()
                ev[0].record(stream)
                sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
                ev[1].record(stream)
                torch.cuda.synchronize()
                ms = ev[0].elapsed_time(ev[1])
                rays = ctx.counters()["rays"]
                if r > 0:
                    res.setdefault((sobol_block, thr, ov), []).append((ms, rays))
    print("head.obj %dx%d %d spp, RR after 1 bounce (p=0.2); %d samples per pass" % (W, H, spp, W * H * spp))
    for k, v in res.items():
        ms = np.array([x[0] for x in v])
        rays = v[0][1]
        print("sobol_block=%-5s service=%2d oversub=%d  ms med/min %.2f/%.2f  Msamples/s %.1f  Mrays/s %.1f  rays/path %.2f" % (
            k[0], k[1], k[2], np.median(ms), ms.min(), W * H * spp / np.median(ms) / 1e3, rays / np.median(ms) / 1e3, rays / (W * H * spp)), flush=True)


if __name__ == "__main__":
    main()
