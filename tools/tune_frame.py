#!/usr/bin/env python3
"""A/B of the configs[1] frame on the GPU box: four launches (gen_primary, raycast closest, gen_shadow, raycast any-hit) against the fused
ptgpu_trace_frame_dev (one launch, rays in registers); interleaved in one process, medians over rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402

W, H = bench.W, bench.H
n = W * H
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx = capi.Context(0)
ctx.set_stream(stream.cuda_stream)
sd, _ = bench.build_head_scene()
sc = capi.Scene(ctx, sd)
cam, light = scene.make_camera(width=W, height=H), scene.make_light()
f = dict(dtype=torch.float32, device=dev)
o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
t, u, v = (torch.empty(n, **f) for _ in range(3))
hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
P = lambda x: x.data_ptr()  # noqa: E731


def four():
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    sc.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)


def fused():
    sc.trace_frame_dev(cam, light, W, H, P(hid), P(t), P(u), P(v), P(sid))


variants = [("four launches", four, "8")] + [("one launch %dx%d" % (w, 64 // w), fused, str(w)) for w in (4, 8, 16, 32, 64)]
times = {k: [] for k, _, _ in variants}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for name, fn, tile in variants:
        os.environ["PTGPU_FRAME_TILE_W"] = tile
        fn()
        ev[0].record(stream)
        for _ in range(20):
            fn()
        ev[1].record(stream)
        torch.cuda.synchronize()
        if r:
            times[name].append(ev[0].elapsed_time(ev[1]) / 20)
for name, _, _ in variants:
    a = np.array(times[name])
    print("%-18s median %.4f ms  min %.4f ms" % (name, np.median(a), a.min()), flush=True)
