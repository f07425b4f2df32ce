#!/usr/bin/env python3
"""A few configs[1] frames for rocprofv3 --kernel-trace --stats (per-kernel split of the raycast launches).  PTGPU_* env knobs apply."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402

W, H = bench.W, bench.H
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
sd, _ = bench.build_head_scene()
sc = capi.Scene(ctx, sd)
cam, light = scene.make_camera(width=W, height=H), scene.make_light()
n = W * H
f = dict(dtype=torch.float32, device=dev)
o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
t, u, v = (torch.empty(n, **f) for _ in range(3))
hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
P = lambda x: x.data_ptr()  # noqa: E731
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    sc.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
ctx.sync()
sc.close()
ctx.close()
