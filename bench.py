#!/usr/bin/env python3
"""bench.py -- Mrays/s and samples/s of the hot path on N MI355X (one process per GPU, torch.distributed over RCCL).

Headline workload (`value`, Mrays/s) = BASELINE.json configs[1]: head.obj, 1920x1080, one pixel-centre primary ray per
pixel traced closest-hit (RayTracer::raycast, occlusion=false) plus one shadow ray per primary hit traced any-hit
(occlusion=true); rays are generated on the device, every buffer is resident in HBM before the timed region.  One "step" =
one such frame: gen_primary -> raycast(closest) -> gen_shadow -> raycast(any-hit).  1 ray = 1 raycast() call, like the
reference's counter (reference src/RayTracer.cpp:192, src/PathTraceRenderer.cpp:884-897).
N>1: the path shards by image tiles with no data-path exchange, so every rank traces its own 1920x1080 frame (weak
scaling); ranks only meet in the timing barriers and the ray-count sum.

Second leg (`path_trace`, samples/s) = BASELINE.json configs[2]: head.obj full path trace 1920x1080, 256 spp, Gaussian AA,
bilinear textures + tangent-space normal maps (procedural stand-ins for the maps, see scene.head_standin_textures),
Russian roulette after 1 bounce (p = 0.2).  One pass = PathTraceRenderer's pass over all 32x32 blocks.  With N>1 the SAME
frame is sharded by blocks (rank r renders blocks r, r+N, ...) and the accumulation tiles are gathered over RCCL
(pathtracer_amd.multigpu.TileGather), inside the timed region: that leg is strong scaling and says so.

The CPU oracle (oracle/) is used ONLY for the cpu_baseline legs (rank 0, N=1) -- never in the measured path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pathtracer_amd import capi, hostapi, multigpu, scene  # noqa: E402

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
GATHER_PEAK_GBS, GATHER_INCOHERENT_GBS = 37950.0, 14230.0  # measured: tools/l1_gather_peak.hip, profiles/r01_n_l1_gather_peak.jsonl
PT_SPP = 256


def build_head_scene(textured=False):
    """Product-side scene construction: triangle array + Woop data + SAH BVH from the C++ host classes (no oracle)."""
    mesh = scene.load_ptmesh()
    woop = hostapi.woop_from_positions(mesh["pos"])
    nodes, lst, stats = hostapi.bvh_build(mesh["pos"], sah=True)
    textures, index = scene.head_standin_textures(mesh) if textured else ([], None)
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=woop,
              materials=scene.mesh_materials(mesh, textured=textured, texture_index=index), textures=textures, nodes=nodes, list=lst)
    return sd, stats


def algorithmic_bytes(ctr, rays):
    """B_ray = 64 B per FlatNode visit + 52 B per Woop test + 28 B ray in + 16 B hit out."""
    return 64 * ctr["node_visits"] + 52 * ctr["tri_tests"] + 44 * rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pt-passes", type=int, default=3, help="timed passes of the path-trace leg (0 = skip the leg)")
    ap.add_argument("--pt-spp", type=int, default=PT_SPP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the measured configuration); gloo = rehearsal of the N>1 control flow on fewer GPUs than ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible: one rank per GPU" % (local_rank, ndev))
    dev_index = local_rank % max(ndev, 1)  # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the scalar reductions live

    sd, bvh_stats = build_head_scene()
    ctx = capi.Context(dev_index)
    # a real (non-null) stream shared by the library's launches and torch's events: torch.cuda.Event only sees the stream
    # it is recorded on, and a NULL handle would make the library fall back to its own stream
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    gscene = capi.Scene(ctx, sd)
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    n = W * H
    f32 = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f32) for _ in range(4))
    t, u, v = (torch.empty(n, **f32) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731

    def step(ev=None):
        ctx.gen_primary_dev(cam, W, H, P(o), P(d))
        if ev:
            ev[0].record(stream)
        gscene.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        if ev:
            ev[1].record(stream)
        gscene.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
        if ev:
            ev[2].record(stream)
        gscene.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
        if ev:
            ev[3].record(stream)

    # per-ray traversal work, measured once with the counting kernels (identical to the CPU traversal's counts)
    ctx.enable_counters(True)
    ctx.reset_counters()
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    gscene.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    cp = ctx.counters()
    gscene.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
    n_shadow = int(valid.sum().item())
    ctx.reset_counters()
    gscene.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
    cs = ctx.counters()
    ctx.enable_counters(False)
    rays_per_step = n + n_shadow
    bytes_primary = algorithmic_bytes(cp, n)

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total_rays = torch.tensor([rays_per_step * args.steps], dtype=torch.int64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_rays, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_rays = int(total_rays.item())
    closest_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    anyhit_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))
    gscene.close()

    pt = path_trace_leg(args, ctx, stream, dev, red_dev, rank, world) if args.pt_passes > 0 else None

    if rank == 0:
        achieved = bytes_primary / (closest_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get("k_raycast_closest_hbm_bytes_per_launch")
        out = {
            "metric": "Mrays/s", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "head.obj 1920x1080 primary closest-hit + shadow any-hit rays (BASELINE configs[1])",
                       "triangles": int(sd["mat_id"].shape[0]), "bvh_nodes": int(sd["nodes"].shape[0]),
                       "rays_per_step_per_gpu": rays_per_step, "parallelism": "tiles x%d" % world},
            "samples_per_s": pt["samples_per_s"] if pt else None,
            "kernels_ms": {"raycast_closest": closest_ms, "raycast_anyhit": anyhit_ms},
            "per_ray": {"primary_nodes": cp["node_visits"] / n, "primary_tris": cp["tri_tests"] / n,
                        "shadow_nodes": cs["node_visits"] / n, "shadow_tris": cs["tri_tests"] / n},
            "roofline": {"kernel": "k_raycast_step<closest-hit>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "algorithmic bytes (64 B/node visit + 52 B/triangle test + 44 B/ray); head.obj's 1.6 MB of "
                                 "nodes+triangles is L2-resident, so frac measures traversal rate, not HBM pressure (`traffic` is what "
                                 "reaches HBM: the rays in and the hit records out). Measured limiter: the vector L1 data path, "
                                 "64 B per lane per step (TD_TD_BUSY 66 % of kernel cycles, L1 hit rate 98 %: profiles/l1a_pmc_summary.json)",
                         "cache_gather": {"achieved": achieved, "peak": GATHER_PEAK_GBS, "peak_incoherent": GATHER_INCOHERENT_GBS,
                                          "unit": "GB/s", "frac": achieved / GATHER_PEAK_GBS,
                                          "note": "the bound that applies: 64-byte records gathered per lane from cache, measured on this "
                                                  "GPU by tools/l1_gather_peak.hip (profiles/r01_n_l1_gather_peak.jsonl): `peak` with >= 4 "
                                                  "neighbouring lanes on the same record (64 B/clk/CU), `peak_incoherent` with every "
                                                  "lane on its own record; same rate from a 16 KiB and a 4 MiB table"}},
        }
        if pt:
            out["path_trace"] = pt
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, cam, light)
            if pt:
                out["path_trace"]["cpu_baseline"] = cpu_baseline_path_trace(sd)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def path_trace_leg(args, ctx, stream, dev, red_dev, rank, world):
    """configs[2]: full path trace of the 1080p head frame; blocks dealt round-robin over ranks, tiles gathered over RCCL."""
    sd, _ = build_head_scene(textured=True)
    gscene = capi.Scene(ctx, sd)
    cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
    tg = multigpu.TileGather(W, H, rank, world, dev, stage_on_cpu=(args.backend == "gloo"))

    def params(seed):
        return scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=args.pt_spp, sobol_block=False, bilinear=True, normal_mapped=True,
                                 filter_width=1.5, termination_prob=0.2, pass_seed=seed)

    def one_pass(seed):
        gscene.render_pass_dev(params(seed), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
        tg.gather(accum)

    # traversal work per ray for the roofline of the integrator kernel: one counted low-spp pass
    ctx.enable_counters(True)
    ctx.reset_counters()
    p = params(1000)
    p.num_aa_rays = 4
    gscene.render_pass_dev(p, cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
    c = ctx.counters()
    ctx.enable_counters(False)
    bytes_per_ray = algorithmic_bytes(c, c["rays"]) / max(c["rays"], 1)
    accum.zero_()
    one_pass(999)  # warm-up (also sizes the workspace)
    accum.zero_()
    ctx.reset_counters()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.pt_passes)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.pt_passes):
        ev[k][0].record(stream)
        gscene.render_pass_dev(params(k), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
        ev[k][1].record(stream)
        tg.gather(accum)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rays = torch.tensor([ctx.counters()["rays"]], dtype=torch.int64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed, rays = float(tmax.item()), int(rays.item())
    render_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    local_rays = ctx.counters()["rays"]
    gscene.close()
    samples = W * H * args.pt_spp * args.pt_passes
    achieved = bytes_per_ray * (local_rays / args.pt_passes) / (render_ms * 1e-3) / 1e9
    return {"workload": "head.obj full path trace %dx%d, %d spp, Gaussian AA, bilinear textures + normal maps (procedural stand-in maps), "
                        "RR after 1 bounce p=0.2 (BASELINE configs[2])" % (W, H, args.pt_spp),
            "passes": args.pt_passes, "scaling": "strong" if world > 1 else "n/a", "tile_gather": ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)") + " all_gather of packed 32x32-block tiles" if world > 1 else "none",
            "samples_per_s": samples / elapsed, "Mrays_per_s": rays / elapsed / 1e6, "rays_per_sample": rays / samples,
            "ms_per_pass": elapsed / args.pt_passes * 1e3, "render_ms_per_pass_rank0": render_ms,
            "per_ray": {"nodes": c["node_visits"] / max(c["rays"], 1), "tris": c["tri_tests"] / max(c["rays"], 1), "bytes": bytes_per_ray},
            "roofline": {"kernel": "k_integrate (+ accumulate)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "note": "algorithmic traversal bytes per ray x rays per pass / device time of the pass; scene is L2-resident"}}


def cpu_baseline(sd, cam, light):
    """The oracle (CPU restatement of the reference's raycast) on the same C2 frame, on this box's host cores."""
    from oracle import pyoracle as po
    threads = min(16, os.cpu_count() or 1)
    o, d = po.gen_primary(cam, W, H)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        c = po.raycast(sd, o, d, False, nthreads=threads)
        t1 = time.perf_counter()
        so, sdir, valid = po.gen_shadow(sd, light, o, d, c)  # single-threaded bookkeeping in the oracle: not timed
        m = valid == 1
        so, sdir = so[m], sdir[m]
        t2 = time.perf_counter()
        po.raycast(sd, so, sdir, True, nthreads=threads)
        t3 = time.perf_counter()
        dt = (t1 - t0) + (t3 - t2)
        rays = o.shape[0] + int(m.sum())
        best = dt if best is None or dt < best else best
    return {"value": rays / best / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "the full configs[1] frame (2,073,600 primary + one shadow ray per hit), best of 3, the two raycast batches timed"}


def cpu_baseline_path_trace(sd_untextured):
    """The oracle's multithreaded pathTraceBlock loop (libm transcendentals, as the reference calls them) on a bounded sample:
    BASELINE configs[0] (head 512x512, 16 spp, Sobol-block AA, RR) = 4.2 M samples."""
    from oracle import pyoracle as po
    threads = min(16, os.cpu_count() or 1)
    w = h = 512
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=16, sobol_block=True, termination_prob=0.2)
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    t0 = time.perf_counter()
    _, ctr = po.render_pass(sd_untextured, prm, cam, lights, nthreads=threads, libm=True)
    dt = time.perf_counter() - t0
    return {"value": w * h * 16 / dt, "unit": "samples/s", "Mrays_per_s": ctr["rays"] / dt / 1e6, "cores": threads, "kind": "port",
            "sample": "configs[0]: head 512x512, 16 spp, Sobol-block AA, RR p=0.2, one pass (4,194,304 samples)"}


if __name__ == "__main__":
    main()
