#!/usr/bin/env python3
"""Diagnostic: where k_integrate's wave cycles go (needs a -DPT_STAMPS build: PTGPU_LIB=build/libptgpu_stamps.so)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402

W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = capi.Context(0)
sd, _ = bench.build_head_scene(textured=len(sys.argv) > 2 and sys.argv[2] == "tex")
sc = capi.Scene(ctx, sd)
cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
accum = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
for sobol_block in (True, False):
    prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=sobol_block, filter_width=1.5, normal_mapped=True)
    ctx.lib.ptgpu_memset(ctx.handle, C.c_void_p(0), 0, 0)
    sc.render_pass_dev(prm, cam, lights, accum.data_ptr())  # warm
    ctx.sync()
    # zero the diagnostic words by reading the base, running, and differencing
    a = (C.c_ulonglong * 32)(); ctx.lib.ptgpu_debug_counters(ctx.handle, a); a = np.array(a[:], dtype=np.int64)
    sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
    b = (C.c_ulonglong * 32)(); ctx.lib.ptgpu_debug_counters(ctx.handle, b); b = np.array(b[:], dtype=np.int64)
    d = b - a
    names = ["control", "shade", "hit", "wshadow", "bounce", "new", "step"]
    tot = sum(d[8 + 3 * k] for k in range(7))
    print("sobol_block=%s  spp=%d" % (sobol_block, spp))
    for k, nme in enumerate(names):
        cyc, inv, lanes = d[8 + 3 * k], d[9 + 3 * k], d[10 + 3 * k]
        print("  %-8s cycles %5.1f%%  invocations %10d  cycles/inv %8.1f  lanes/inv %5.1f" % (nme, 100.0 * cyc / tot, inv, cyc / max(inv, 1), lanes / max(inv, 1)))
    steps = d[9 + 18]
    print("  phase: lanes of the chosen kind at its first step %.1f, node phases %.1f%%, alive lanes %.1f" % (d[29] / steps, 100.0 * d[30] / steps, d[31] / steps))
