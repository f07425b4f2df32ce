#!/usr/bin/env python3
"""How much of the textured path-trace pass is texel-fetch locality: the configs[2] frame at 16 spp with the stand-in maps at
several sizes (a 64x64 map stays in L1/L2, the 1600x1600 ones are 10 MB each and live in the Infinity Cache).
Run on the GPU box:  python tools/tune_textures.py [sizes, comma separated]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,400,1600").split(",")]
    W, H = bench.W, bench.H
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    mesh = scene.load_ptmesh()
    nodes, lst, _ = hostapi.bvh_build(mesh["pos"], True)
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
    prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=16, sobol_block=False, bilinear=True, normal_mapped=True,
                            filter_width=1.5, termination_prob=0.2, pass_seed=0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for size in sizes + [0]:
        textures, index = scene.head_standin_textures(mesh, size=size) if size else ([], None)
        sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
                  materials=scene.mesh_materials(mesh, textured=bool(size), texture_index=index), textures=textures, nodes=nodes, list=lst)
        sc = capi.Scene(ctx, sd)
        ms = []
        for r in range(4):
            ev[0].record(stream)
            sc.render_pass_dev(prm, cam, [light], accum.data_ptr())
            ev[1].record(stream)
            torch.cuda.synchronize()
            ms.append(ev[0].elapsed_time(ev[1]))
        print("texture size %4d: ms per 16 spp pass med/min %.2f/%.2f" % (size, float(np.median(ms[1:])), min(ms[1:])), flush=True)
        sc.close()
    ctx.close()


if __name__ == "__main__":
    main()
