#!/usr/bin/env python3
"""Stress of the pooled integrator's start-up and drain phases: many small launches (few samples per workgroup: most trace waves wait),
every image compared with the lane-owned integrator's, the watchdog word read after every launch.  python tools/stress_pool.py [launches]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = capi.Context(0)
sd, _ = bench.build_head_scene(textured=False)
sc = capi.Scene(ctx, sd)
rng = np.random.RandomState(1)
t0 = time.time()
bad = 0
for k in range(n):
    w, h = int(rng.randint(8, 400)), int(rng.randint(8, 300))
    spp = int(rng.choice([1, 2, 3, 8]))
    sb = bool(rng.randint(2))
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    prm = scene.make_params(width=w, height=h, bounces=int(rng.choice([-1, 0, 2])), num_aa_rays=spp, sobol_block=sb, filter_width=1.5, pass_seed=k)
    os.environ["PTGPU_PT_POOL"] = "0"
    a = sc.render_pass(prm, cam, lights)
    os.environ["PTGPU_PT_POOL"] = "1"
    b = sc.render_pass(prm, cam, lights)
    ctx.spill_status()  # raises if the watchdog ended the launch
    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
        bad += 1
        print("launch %d (%dx%d, %d spp, sobol_block=%s): images differ" % (k, w, h, spp, sb), flush=True)
    if k % 50 == 49:
        print("%d launches, %.1f s, %d mismatches" % (k + 1, time.time() - t0, bad), flush=True)
print("done: %d launches, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
