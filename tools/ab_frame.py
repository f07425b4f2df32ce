#!/usr/bin/env python3
"""A/B timing of the configs[1] frame across library builds (tools/build_variant.sh): one child process per build and round, rounds
interleaved.  python tools/ab_frame.py "base,x4" [rounds]   (base = the in-tree library) -> fused frame, closest, any-hit, random rays"""
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
W, H = bench.W, bench.H
n = W * H
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = capi.Context(0); ctx.set_stream(stream.cuda_stream)
sd, _ = bench.build_head_scene()
sc = capi.Scene(ctx, sd)
cam, light = scene.make_camera(width=W, height=H), scene.make_light()
f = dict(dtype=torch.float32, device=dev)
o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
t, u, v = (torch.empty(n, **f) for _ in range(3))
hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
P = lambda x: x.data_ptr()
ctx.gen_primary_dev(cam, W, H, P(o), P(d))
sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
sc.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
g = torch.Generator(device="cpu").manual_seed(1)
ro = (torch.rand(n, 3, generator=g) * torch.tensor([2.8, 1.4, 0.9]) + torch.tensor([-1.4, -0.45, -0.45])).to(dev).contiguous()
rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).mul(3.0).to(dev).contiguous()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def timed(fn, reps=20):
    best = 1e9
    for r in range(4):
        fn(); ev[0].record(stream)
        for _ in range(reps): fn()
        ev[1].record(stream); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / reps)
    return best
res = [timed(lambda: sc.trace_frame_dev(cam, light, W, H, P(hid), P(t), P(u), P(v), P(sid))),
       timed(lambda: sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))),
       timed(lambda: sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)),
       timed(lambda: sc.raycast_dev(P(ro), P(rd), n, False, P(hid), None, None, None))]
print(" ".join("%%.4f" %% x for x in res), int(hid.sum().item()) ^ int(sid.sum().item()))
""" % ROOT


def main():
    names = sys.argv[1].split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            env = dict(os.environ)
            if n != "base":
                env["PTGPU_LIB"] = os.path.join(ROOT, "build", "libptgpu_%s.so" % n)
            p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                print(n, "FAILED", p.stderr[-600:], flush=True)
                return 1
            res[n].append(p.stdout.split())
            print(n, p.stdout.strip(), flush=True)
    for n in names:
        best = [min(float(x[k]) for x in res[n]) for k in range(4)]
        print("%-10s frame %.4f  closest %.4f  any-hit %.4f  random %.4f ms   checksum %s" % (n, *best, res[n][0][4]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
