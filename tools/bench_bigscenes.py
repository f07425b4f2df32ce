#!/usr/bin/env python3
"""Throughput + parity spot-checks on the larger BASELINE configurations (1 GPU):
  atrium   ~262k-triangle procedural stand-in for Crytek Sponza, 3840x2160 (configs[3])
  tetra    10M-triangle tetrahedra cloud with depth of field, 1920x1080 (configs[4])
Usage on the GPU box:  python tools/bench_bigscenes.py [atrium|tetra|both] [tetra_tris]
Prints one JSON line per scene (also the per-ray traversal counts that define the algorithmic bytes of the roofline)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402

HBM_PEAK = 8000.0


def bench_scene(name, mesh, view, W, H, spp, dof, check_rays=20000):
    t0 = time.perf_counter()
    nodes, lst, st = hostapi.bvh_build(mesh["pos"], True, threads=min(16, os.cpu_count() or 1))
    t_build = time.perf_counter() - t0
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
              materials=mesh["materials"], textures=[], nodes=nodes, list=lst)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    t0 = time.perf_counter()
    sc = capi.Scene(ctx, sd)
    t_upload = time.perf_counter() - t0
    stats = sc.stats()
    cam, lights = view(W, H)
    n = W * H
    f = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
    t, u, v = (torch.empty(n, **f) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def frame(timed):
        ctx.gen_primary_dev(cam, W, H, P(o), P(d))
        if timed:
            ev[0].record(stream)
        sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        if timed:
            ev[1].record(stream)
        sc.gen_shadow_dev(lights[0], n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
        if timed:
            ev[2].record(stream)
        sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
        if timed:
            ev[3].record(stream)

    ctx.enable_counters(True)
    ctx.reset_counters()
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    cp = ctx.counters()
    ctx.enable_counters(False)
    for _ in range(3):
        frame(False)
    cms, ams = [], []
    for _ in range(10):
        frame(True)
        torch.cuda.synchronize()
        cms.append(ev[0].elapsed_time(ev[1]))
        ams.append(ev[2].elapsed_time(ev[3]))
    n_shadow = int(valid.sum().item())
    closest_ms, anyhit_ms = float(np.median(cms)), float(np.median(ams))
    bytes_primary = 64 * cp["node_visits"] + 52 * cp["tri_tests"] + 44 * n
    # parity spot-check against the CPU oracle on a sample of the primary rays (size-independent property: identical hit records)
    from oracle import pyoracle as po
    idx = np.random.RandomState(1).choice(n, check_rays, replace=False)
    ho, hd = o.view(-1, 3)[torch.from_numpy(idx).to(dev)].cpu().numpy(), d.view(-1, 3)[torch.from_numpy(idx).to(dev)].cpu().numpy()
    c = po.raycast(sd, ho, hd, False, nthreads=min(16, os.cpu_count() or 1))
    gid, gt = hid.cpu().numpy()[idx], t.cpu().numpy()[idx]
    parity = bool(np.array_equal(gid, c["id"]) and np.array_equal(gt.view(np.uint32), c["t"].view(np.uint32)))
    # path trace
    accum = torch.zeros(H * W * 4, **f)
    prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=False, filter_width=1.5, depth_of_field=dof,
                            aperture=0.005, focal_distance=1.2, termination_prob=0.2, pass_seed=0)
    sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
    torch.cuda.synchronize()
    pms, rays = [], 0
    for k in range(3):
        prm.pass_seed = k + 1
        ctx.reset_counters()
        ev[0].record(stream)
        sc.render_pass_dev(prm, cam, lights, accum.data_ptr())
        ev[1].record(stream)
        torch.cuda.synchronize()
        pms.append(ev[0].elapsed_time(ev[1]))
        rays = ctx.counters()["rays"]
    pass_ms = float(np.median(pms))
    out = {"scene": name, "triangles": int(mesh["pos"].shape[0]), "bvh_nodes": int(nodes.shape[0]), "bvh_depth": st["true_depth"],
           "bvh_build_s": t_build, "scene_upload_s": t_upload, "device_bytes": stats, "image": "%dx%d" % (W, H),
           "primary_closest_ms": closest_ms, "primary_Grays_per_s": n / closest_ms / 1e6, "shadow_anyhit_ms": anyhit_ms,
           "shadow_Grays_per_s": n / anyhit_ms / 1e6, "shadow_rays": n_shadow,
           "per_ray": {"primary_nodes": cp["node_visits"] / n, "primary_tris": cp["tri_tests"] / n, "bytes": bytes_primary / n},
           "roofline_closest": {"bound": "hbm", "achieved": bytes_primary / (closest_ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
                                "frac": bytes_primary / (closest_ms * 1e-3) / 1e9 / HBM_PEAK},
           "parity_sample": {"rays": check_rays, "hit_fraction": float((c["id"] >= 0).mean()), "bit_exact": parity},
           "path_trace": {"spp": spp, "depth_of_field": bool(dof), "ms_per_pass": pass_ms, "Msamples_per_s": n * spp / pass_ms / 1e3,
                          "Mrays_per_s": rays / pass_ms / 1e3, "rays_per_sample": rays / (n * spp)}}
    print(json.dumps(out), flush=True)
    sc.close()
    ctx.close()
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("atrium", "both"):
        bench_scene("atrium_262k", scene.atrium_scene(), scene.atrium_view, 3840, 2160, 16, False)
    if which in ("tetra", "both"):
        ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
        bench_scene("tetra_cloud_%dM" % (ntri // 1_000_000), scene.tetra_cloud_scene(ntri), scene.tetra_cloud_view, 1920, 1080, 16, True)


if __name__ == "__main__":
    main()
