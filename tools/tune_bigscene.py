#!/usr/bin/env python3
"""A/B timing of the raycast kernels on the larger scenes (variants interleaved in ONE process, identical hit ids asserted): the
one-ray-per-lane kernel's LDS stack size, or the persistent kernel with refill (impl 2) at a refill threshold and grid oversubscription.
Run on the GPU box:  python tools/tune_bigscene.py [atrium|tetra] [tetra_tris] [variant,variant,...]   variant = nlds | impl:nlds[:refill[:oversub]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    variants = (sys.argv[3] if len(sys.argv) > 3 else "8,12,16,24").split(",")
    if which == "atrium":
        mesh, view, W, H = scene.atrium_scene(), scene.atrium_view, 3840, 2160
    else:
        mesh, view, W, H = scene.tetra_cloud_scene(ntri), scene.tetra_cloud_view, 1920, 1080
    nodes, lst, st = hostapi.bvh_build(mesh["pos"], True, threads=min(16, os.cpu_count() or 1))
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
              materials=mesh["materials"], textures=[], nodes=nodes, list=lst)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    sc = capi.Scene(ctx, sd)
    cam, lights = view(W, H)
    n = W * H
    f = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
    t, u, v = (torch.empty(n, **f) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    sc.gen_shadow_dev(lights[0], n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
    torch.cuda.synchronize()
    ref_hid, ref_sid = hid.clone(), sid.clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = {k: dict(closest=[], anyhit=[]) for k in variants}
    for r in range(6):
        for k in variants:
            f_ = (k.split(":") + ["", "", ""])[:4] if ":" in k else ["1", k, "", ""]
            os.environ["PTGPU_RAYCAST_IMPL"], os.environ["PTGPU_RAYCAST_NLDS"] = f_[0], f_[1]
            os.environ["PTGPU_RAYCAST_REFILL"], os.environ["PTGPU_RAYCAST_OVERSUB"] = f_[2] or "32", f_[3] or "1"
            for kind in ("closest", "anyhit"):
                ev[0].record(stream)
                if kind == "closest":
                    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
                else:
                    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
                ev[1].record(stream)
                torch.cuda.synchronize()
                if r == 0:
                    assert torch.equal(hid, ref_hid) and torch.equal(sid, ref_sid), k
                else:
                    times[k][kind].append(ev[0].elapsed_time(ev[1]))
    print("%s: %d triangles, tree depth %d, %d rays" % (which, mesh["pos"].shape[0], st["true_depth"], n))
    for k in variants:
        print("variant %-12s closest ms %.3f  anyhit ms %.3f" % (k, np.median(times[k]["closest"]), np.median(times[k]["anyhit"])), flush=True)


if __name__ == "__main__":
    main()
