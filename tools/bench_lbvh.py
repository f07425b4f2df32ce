#!/usr/bin/env python3
"""Device-built linear BVH (ptgpu_bvh_build_lbvh) against the reference's SAH sweep (host builder): build time and the cost of
traversing each tree with the same kernels.  Run on the GPU box:  python tools/bench_lbvh.py [head|atrium|tetra] [tetra_tris]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "head"
    if which == "head":
        mesh = scene.load_ptmesh()
        mesh = dict(mesh, materials=scene.mesh_materials(mesh))
        view, W, H = (lambda w, h: (scene.make_camera(width=w, height=h), [scene.make_light()])), 1920, 1080
    elif which == "atrium":
        mesh, view, W, H = scene.atrium_scene(), scene.atrium_view, 3840, 2160
    else:
        mesh, view, W, H = scene.tetra_cloud_scene(int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000), scene.tetra_cloud_view, 1920, 1080
    pos = mesh["pos"]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    t0 = time.perf_counter()
    sah = hostapi.bvh_build(pos, True, threads=min(16, os.cpu_count() or 1))
    t_sah = time.perf_counter() - t0
    ctx.bvh_build_lbvh(pos[: min(1000, pos.shape[0])], 3)  # warm-up (module load)
    t0 = time.perf_counter()
    lb = ctx.bvh_build_lbvh(pos, 3)
    t_lbvh = time.perf_counter() - t0
    woop = hostapi.woop_from_positions(pos)
    cam, lights = view(W, H)
    n = W * H
    f = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
    t, u, v = (torch.empty(n, **f) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    out = {"scene": which, "triangles": int(pos.shape[0]), "sah_build_s": t_sah, "lbvh_build_s_host_to_host": t_lbvh}
    ref_t = None
    # scene creation: tree through host arrays (build + ptgpu_scene_create) against ptgpu_scene_create_lbvh (nothing leaves the device)
    sd_l = dict(pos=pos, nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=woop, materials=mesh["materials"], textures=[], nodes=lb[0], list=lb[1])
    t0 = time.perf_counter()
    sc_a = capi.Scene(ctx, sd_l)
    ctx.sync()
    out["scene_create_from_host_tree_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sc_b = capi.Scene(ctx, sd_l, lbvh_max_tris_in_leaf=3)
    ctx.sync()
    out["scene_create_lbvh_on_device_s"] = time.perf_counter() - t0
    out["build_plus_create_through_host_s"] = t_lbvh + out["scene_create_from_host_tree_s"]
    assert sc_a.stats() == sc_b.stats() and sc_b.num_nodes_built == lb[0].shape[0]
    sc_a.close()
    sc_b.close()
    for name, (nodes, lst, extra) in (("sah", sah), ("lbvh", lb)):
        sd = dict(pos=pos, nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=woop, materials=mesh["materials"], textures=[], nodes=nodes, list=lst)
        sc = capi.Scene(ctx, sd)
        ctx.enable_counters(True)
        ctx.reset_counters()
        ctx.gen_primary_dev(cam, W, H, P(o), P(d))
        sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        cp = ctx.counters()
        ctx.enable_counters(False)
        cms, ams = [], []
        for k in range(8):
            ctx.gen_primary_dev(cam, W, H, P(o), P(d))
            ev[0].record(stream)
            sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
            ev[1].record(stream)
            sc.gen_shadow_dev(lights[0], n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
            ev[2].record(stream)
            sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
            ev[3].record(stream)
            torch.cuda.synchronize()
            if k >= 2:
                cms.append(ev[0].elapsed_time(ev[1]))
                ams.append(ev[2].elapsed_time(ev[3]))
        if ref_t is None:
            ref_t = t.clone()
        else:
            out["rays_whose_closest_t_differs_from_sah_tree"] = int((t.view(torch.int32) != ref_t.view(torch.int32)).sum().item())
        out[name] = {"nodes": int(nodes.shape[0]), "depth": int(sc.stats()["depth"]), "closest_ms": float(np.median(cms)), "anyhit_ms": float(np.median(ams)),
                     "nodes_per_ray": cp["node_visits"] / n, "tris_per_ray": cp["tri_tests"] / n}
        sc.close()
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
