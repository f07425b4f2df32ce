#!/usr/bin/env python3
"""A/B timing of the interior-record orders (PTGPU_NODE_ORDER, read at scene creation: csrc/ptgpu.hip interior_slots) on the larger
scenes: one scene per order over the SAME host-built tree in ONE process, launches interleaved, hit ids and the accumulated image
asserted identical to pre-order's.
Run on the GPU box:  python tools/tune_layout.py [atrium|tetra] [tetra_tris] [orders, e.g. 0,1,3,4,9] [spp]"""
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
    orders = (sys.argv[3] if len(sys.argv) > 3 else "0,1,3,4,9").split(",")
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    if which == "atrium":
        mesh, view, W, H, dof = scene.atrium_scene(), scene.atrium_view, 3840, 2160, False
    else:
        mesh, view, W, H, dof = scene.tetra_cloud_scene(ntri), scene.tetra_cloud_view, 1920, 1080, True
    nodes, lst, st = hostapi.bvh_build(mesh["pos"], True, threads=min(16, os.cpu_count() or 1))
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
              materials=mesh["materials"], textures=[], nodes=nodes, list=lst)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    scenes = {}
    for k in orders:
        os.environ["PTGPU_NODE_ORDER"] = k
        scenes[k] = capi.Scene(ctx, sd)
    os.environ.pop("PTGPU_NODE_ORDER")
    cam, lights = view(W, H)
    n = W * H
    f = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
    t, u, v = (torch.empty(n, **f) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    accum = torch.zeros(H * W * 4, **f)
    P = lambda x: x.data_ptr()  # noqa: E731
    sc0 = scenes[orders[0]]
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    sc0.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    sc0.gen_shadow_dev(lights[0], n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
    prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=True, depth_of_field=dof, aperture=0.01,
                            focal_distance=1.2, termination_prob=0.2, pass_seed=5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = {k: dict(closest=[], anyhit=[], render=[]) for k in orders}
    ref = {}
    for r in range(5):
        for k in orders:
            sc = scenes[k]
            for kind in ("closest", "anyhit", "render"):
                if kind == "render":
                    accum.zero_()
                ev[0].record(stream)
                if kind == "closest":
                    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
                elif kind == "anyhit":
                    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
                else:
                    sc.render_pass_dev(prm, cam, lights, P(accum))
                ev[1].record(stream)
                torch.cuda.synchronize()
                if r == 0:
                    got = dict(closest=hid, anyhit=sid, render=accum.view(torch.int32))[kind]
                    if kind not in ref:
                        ref[kind] = got.clone()
                    assert torch.equal(got, ref[kind]), "order %s: %s differs from order %s" % (k, kind, orders[0])
                else:
                    times[k][kind].append(ev[0].elapsed_time(ev[1]))
    print("%s: %d triangles, tree depth %d, %d rays, render %d spp" % (which, mesh["pos"].shape[0], st["true_depth"], n, spp))
    for k in orders:
        print("order %-3s closest ms %.3f  anyhit ms %.3f  pass ms %.2f" % (k, np.median(times[k]["closest"]), np.median(times[k]["anyhit"]),
                                                                             np.median(times[k]["render"])), flush=True)


if __name__ == "__main__":
    main()
