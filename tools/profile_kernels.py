#!/usr/bin/env python3
"""Short, fixed launch sequence for rocprofv3 (kernel trace or one --pmc pass per run):
  5 x configs[1] frame as bench.py's step runs it (ptgpu_trace_frame_dev, one launch) and 5 x as four launches (gen_primary, raycast
  closest, gen_shadow, raycast any-hit) on head.obj 1920x1080, then
  1 x path-trace pass of the configs[2] frame at 16 spp (Gaussian AA, textures + normal maps, RR); with PROFILE_POOLED=1 in the
  environment the same pass once more through the pooled integrator.
Usage on the GPU box:   rocprofv3 --kernel-trace --stats ... -- python3 tools/profile_kernels.py
                        rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/profile_kernels.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402


def main():
    W, H = bench.W, bench.H
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    sd, _ = bench.build_head_scene()
    sc = capi.Scene(ctx, sd)
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    n = W * H
    f = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
    t, u, v = (torch.empty(n, **f) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731
    for _ in range(5):
        sc.trace_frame_dev(cam, light, W, H, P(hid), P(t), P(u), P(v), P(sid))
    for _ in range(5):
        ctx.gen_primary_dev(cam, W, H, P(o), P(d))
        sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        sc.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
        sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
    torch.cuda.synchronize()
    sc.close()
    if spp > 0:
        sdt, _ = bench.build_head_scene(textured=True)
        sct = capi.Scene(ctx, sdt)
        accum = torch.zeros(H * W * 4, **f)
        prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=False, bilinear=True, normal_mapped=True,
                                filter_width=1.5, termination_prob=0.2, pass_seed=0)
        sct.render_pass_dev(prm, cam, [light], accum.data_ptr())
        torch.cuda.synchronize()
        print("path-trace pass: %d samples, %d rays" % (n * spp, ctx.counters()["rays"]))
        if os.environ.get("PROFILE_POOLED", "0") != "0":  # the same pass through the experimental pooled integrator (k_integrate_pool)
            os.environ["PTGPU_PT_POOL"] = "1"
            sct.render_pass_dev(prm, cam, [light], accum.data_ptr())
            torch.cuda.synchronize()
            os.environ["PTGPU_PT_POOL"] = "0"
        sct.close()
    ctx.close()


if __name__ == "__main__":
    main()
