#!/usr/bin/env python3
"""A/B timing of the path-trace pass across library builds (tools/build_variant.sh): one child process per build and round, rounds
interleaved.  python tools/ab_render.py "base,unph1,unph2" [spp] [rounds] [tex]   (base = the in-tree library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
import torch, numpy as np
import bench
from pathtracer_amd import capi, scene
spp, tex = int(sys.argv[1]), sys.argv[2] == "tex"
W, H = 1920, 1080
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = capi.Context(0); ctx.set_stream(stream.cuda_stream)
sd, _ = bench.build_head_scene(textured=tex)
sc = capi.Scene(ctx, sd)
cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
out = []
for sobol_block in (True, False):
    ms = []
    for r in range(4):
        prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=sobol_block, filter_width=1.5, normal_mapped=tex, bilinear=tex, termination_prob=0.2, pass_seed=r)
        ev[0].record(stream); sc.render_pass_dev(prm, cam, lights, accum.data_ptr()); ev[1].record(stream); torch.cuda.synchronize()
        if r: ms.append(ev[0].elapsed_time(ev[1]))
    out.append(min(ms))
print("%%.3f %%.3f %%.6f" %% (out[0], out[1], float(accum.sum().item())))
""" % ROOT


def main():
    names = sys.argv[1].split(",")
    spp = sys.argv[2] if len(sys.argv) > 2 else "16"
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    tex = sys.argv[4] if len(sys.argv) > 4 else "plain"
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            env = dict(os.environ)
            if n != "base":
                env["PTGPU_LIB"] = os.path.join(ROOT, "build", "libptgpu_%s.so" % n)
                if n.startswith("pool_"):  # a build meant to be run with the pooled integrator (NOT any name in p: "prev" once was)
                    env["PTGPU_PT_POOL"] = "1"
            if n == "pool":
                env.pop("PTGPU_LIB", None)
                env["PTGPU_PT_POOL"] = "1"
            p = subprocess.run([sys.executable, "-c", CHILD, spp, tex], env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                print(n, "FAILED", p.stderr[-400:], flush=True)
                return 1
            res[n].append(p.stdout.split())
            print(n, p.stdout.strip(), flush=True)
    for n in names:
        sb = min(float(x[0]) for x in res[n]); ga = min(float(x[1]) for x in res[n])
        print("%-12s sobol-block %.2f ms  gaussian %.2f ms  checksum %s" % (n, sb, ga, res[n][0][2]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
