#!/usr/bin/env python3
"""Pooled integrator (PTGPU_PT_POOL=1) against the lane-owned one in one process: images bit-identical, ray counters equal, ms per pass.
python tools/ab_pool.py [spp] [WxH] [tex]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256x144").split("x"))
    tex = len(sys.argv) > 3 and sys.argv[3] == "tex"
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    sd, _ = bench.build_head_scene(textured=tex)
    sc = capi.Scene(ctx, sd)
    cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ok = True
    for sobol_block in (True, False):
        imgs, rays, best = {}, {}, {}
        for r in range(rounds):
            for pool in ("0", "1"):
                os.environ["PTGPU_PT_POOL"] = pool
                prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=sobol_block, filter_width=1.5, normal_mapped=tex,
                                        bilinear=tex, termination_prob=0.2, pass_seed=3)
                accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
                ctx.reset_counters()
                ev[0].record(stream)
                sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
                ev[1].record(stream)
                torch.cuda.synchronize()
                ms = ev[0].elapsed_time(ev[1])
                best[pool] = min(best.get(pool, 1e30), ms)
                imgs[pool] = accum.cpu().numpy().view(np.uint32)
                rays[pool] = ctx.counters()["rays"]
                print("sobol_block=%s pool=%s round %d: %.3f ms, rays %d" % (sobol_block, pool, r, ms, rays[pool]), flush=True)
                if hasattr(ctx.lib, "ptgpu_debug_counters") and pool == "1":
                    import ctypes as C
                    a = (C.c_ulonglong * 32)()
                    ctx.lib.ptgpu_debug_counters(ctx.handle, a)
                    if a[7]:
                        print("  WATCHDOG: aborts %d first group %d wave %d where %d | ray q %d %d hit q %d %d new q %d %d retired %d cursor %d" % (
                            a[7], a[9] & 0xffffffff, a[9] >> 32, a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18]), flush=True)
        a, b = imgs["0"], imgs["1"]
        fa, fb = a.view(np.float32), b.view(np.float32)
        same = bool(((a == b) | (np.isnan(fa) & np.isnan(fb))).all()) and rays["0"] == rays["1"]
        ok = ok and same
        print("sobol_block=%s: identical=%s  lane-owned %.3f ms  pooled %.3f ms  (x%.3f)" % (sobol_block, same, best["0"], best["1"], best["0"] / best["1"]), flush=True)
        if not same:
            print("  differing words: %d of %d; rays %d vs %d" % (int((a != b).sum()), a.size, rays["0"], rays["1"]))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
