#!/usr/bin/env python3
"""Knob sweep of the pooled integrator in one process: python tools/tune_pool.py "refill,low;refill,low;..." [spp] [tex]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402

variants = [v.split(",") for v in sys.argv[1].split(";")]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tex = len(sys.argv) > 3 and sys.argv[3] == "tex"
W, H = 1920, 1080
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx = capi.Context(0)
ctx.set_stream(stream.cuda_stream)
sd, _ = bench.build_head_scene(textured=tex)
sc = capi.Scene(ctx, sd)
cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=False, filter_width=1.5, normal_mapped=tex, bilinear=tex, termination_prob=0.2)
res = {}
for r in range(4):
    for v in [["owned"]] + variants:
        if v[0] == "owned":
            os.environ["PTGPU_PT_POOL"] = "0"
        else:
            os.environ["PTGPU_PT_POOL"] = "1"
            os.environ["PTGPU_POOL_REFILL"], os.environ["PTGPU_POOL_LOW"] = v[0], v[1]
        ev[0].record(stream)
        sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
        ev[1].record(stream)
        torch.cuda.synchronize()
        if r:
            res.setdefault(",".join(v), []).append(ev[0].elapsed_time(ev[1]))
if hasattr(ctx.lib, "ptgpu_debug_counters"):
    import ctypes as C
    a = (C.c_ulonglong * 32)()
    ctx.lib.ptgpu_debug_counters(ctx.handle, a)
    print("watchdog aborts:", a[7], list(a[8:19]) if a[7] else "")
for k, v in res.items():
    print("%-10s min %.3f ms" % (k, min(v)), flush=True)
