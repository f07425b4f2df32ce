#!/usr/bin/env python3
"""Does the ORDER of the rays in a batch matter?  The configs[1] frame's primary rays in scanline order (a wave = 64x1 pixels) against
the same rays in TWxTH-tiled order (a wave = 8x8 pixels, ...).  Run on the GPU box:  python tools/tune_tiling.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import capi, scene  # noqa: E402


def tiled_perm(W, H, tw, th):
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    key = ((ys // th) * ((W + tw - 1) // tw) + (xs // tw)) * (tw * th) + (ys % th) * tw + (xs % tw)
    return np.argsort(key.reshape(-1), kind="stable")


def main():
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
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    torch.cuda.synchronize()
    o0, d0 = o.view(n, 3).clone(), d.view(n, 3).clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for name, tile in (("scanline 64x1", None), ("8x8", (8, 8)), ("16x4", (16, 4)), ("4x16", (4, 16)), ("32x2", (32, 2)), ("16x16", (16, 16))):
        if tile is None:
            oo, dd = o0.contiguous(), d0.contiguous()
        else:
            perm = torch.from_numpy(tiled_perm(W, H, *tile)).to(dev)
            oo, dd = o0[perm].contiguous(), d0[perm].contiguous()
        cms, ams = [], []
        for k in range(12):
            ev[0].record(stream)
            sc.raycast_dev(P(oo), P(dd), n, False, P(hid), P(t), P(u), P(v))
            ev[1].record(stream)
            sc.gen_shadow_dev(light, n, P(oo), P(dd), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
            ev[2].record(stream)
            sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
            ev[3].record(stream)
            torch.cuda.synchronize()
            if k >= 2:
                cms.append(ev[0].elapsed_time(ev[1]))
                ams.append(ev[2].elapsed_time(ev[3]))
        print("%-14s closest %.4f ms  any-hit %.4f ms  hits %d" % (name, np.median(cms), np.median(ams), int((hid >= 0).sum().item())), flush=True)


if __name__ == "__main__":
    main()
