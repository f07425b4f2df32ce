#!/usr/bin/env python3
"""Diagnostic: where the pooled integrator's wave time goes (needs a -DPT_POOL_STATS build: PTGPU_LIB=build/libptgpu_poolstats.so)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402

os.environ["PTGPU_PT_POOL"] = "1"
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
tex = len(sys.argv) > 2 and sys.argv[2] == "tex"
ctx = capi.Context(0)
sd, _ = bench.build_head_scene(textured=tex)
sc = capi.Scene(ctx, sd)
cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=False, filter_width=1.5, normal_mapped=tex, bilinear=tex)


def counters():
    a = (C.c_ulonglong * 32)()
    ctx.lib.ptgpu_debug_counters(ctx.handle, a)
    return np.array(a[:], dtype=np.int64)


sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
ctx.sync()
a = counters()
sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
ctx.sync()
d = counters() - a
t = d[8:19]
print("TRACE waves: life %d ticks total" % t[10])
print("  hand-over+refill %5.1f%% (%d events, %.1f ended + %.1f empty lanes each)" % (100.0 * t[0] / t[10], t[1], t[2] / max(t[1], 1), t[3] / max(t[1], 1)))
print("  idle sleep       %5.1f%% (%d sleeps)" % (100.0 * t[4] / t[10], t[5]))
print("  step phases      %5.1f%% (%d phases, %.1f lanes with a ray, %.1f of the chosen kind at the first step; %.1f ticks per phase)" % (
    100.0 * t[6] / t[10], t[7], t[8] / max(t[7], 1), t[9] / max(t[7], 1), t[6] / max(t[7], 1)))
h = d[19:28]
print("SHADE waves: life %d ticks total" % h[8])
print("  idle sleep %5.1f%% (%d sleeps)" % (100.0 * h[0] / h[8], h[1]))
print("  hit jobs   %5.1f%% (%d jobs, %.1f lanes each, %.0f ticks each)" % (100.0 * h[2] / h[8], h[3], h[4] / max(h[3], 1), h[2] / max(h[3], 1)))
print("  new jobs   %5.1f%% (%d jobs, %.1f lanes each, %.0f ticks each)" % (100.0 * h[5] / h[8], h[6], h[7] / max(h[6], 1), h[5] / max(h[6], 1)))
