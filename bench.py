#!/usr/bin/env python3
"""bench.py -- Mrays/s and samples/s of the hot path on N MI355X (one process per GPU).

N = 1.  Headline (`value`, Mrays/s) = BASELINE.json configs[1]: head.obj, 1920x1080, one pixel-centre primary ray per pixel traced
closest-hit (RayTracer::raycast, occlusion=false) plus one shadow ray per primary hit traced any-hit (occlusion=true); scene and
output buffers are resident in HBM before the timed region.  One "step" = one such frame = ONE launch of ptgpu_trace_frame_dev (rays
built and kept in registers; outputs: primary hit id/t/u/v and the shadow ray's occluder per pixel, bit-identical to the four-call
form gen_primary -> raycast(closest) -> gen_shadow -> raycast(any-hit), which is timed beside it as `four_launch`).  1 ray = 1
raycast() call, like the reference's counter (reference src/RayTracer.cpp:192, src/PathTraceRenderer.cpp:884-897).
Side legs at N = 1: `path_trace` = configs[2] (head.obj 1080p 256 spp, Gaussian AA, textures + normal maps), `extra` = the large
scenes (configs[3] stand-in at 3840x2160, configs[4] 10 M triangles + depth of field), each with a `cpu_baseline` or counters.

N > 1.  Headline = BASELINE.json configs[3], the configuration BASELINE names for 2/4/8 GPUs: the ~262k-triangle scene path traced at
3840x2160, ONE frame sharded by image tiles -- rank r renders the 32x32 blocks r, r+N, ... (reference blocks,
src/PathTraceRenderer.cpp:816-841) -- and after every pass the tiles are exchanged by ptgpu_gather_tiles_dev (ONE ncclAllGather over
xGMI behind the C ABI) INSIDE the timed region: strong scaling, and the line says so.  One "step" = one pass of 64 samples per pixel over
the whole frame + the exchange (the reference accumulates passes of m_NumAaRays samples; 16 such passes are configs[3]'s 1024 spp;
tests/test_config3_gpu.py runs a 1024-spp pass in one piece).  `strong_scaling_base` = the same pass on rank 0 alone, measured in the
same job before the timed region.  Side legs: `independent_frames` = configs[1] with every rank tracing its own frame (no exchange: weak
scaling by construction), `path_trace` = configs[2] sharded the same way.

Launch: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run, one per GPU) before touching a GPU; under
torch.distributed.run (WORLD_SIZE set by the launcher) it is one of the ranks.  The CPU oracle (oracle/) is used ONLY for the
cpu_baseline legs (rank 0, N=1) -- never in the measured path.  Hardware-counter fields (`traffic`, `l2_hit`, `td_busy`, ...) are NOT
collected by this run: they are replayed from committed rocprofv3 --pmc passes and every roofline object names its `counters_from`."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
PT_SPP = 256
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md "HBM": 8.0 TB/s spec
L2_GATHER_PEAK_GBS = 36900.0  # MI355X_MICROARCH.md "L2 (per XCD)": 34.5 TB/s aggregate, 36.9 TB/s with the L1-reuse contribution
# measured on this GPU by tools/l1_gather_peak.hip (profiles/r01_n_l1_gather_peak.jsonl): 64-byte records gathered per lane from cache
GATHER_COHERENT_GBS, GATHER_INCOHERENT_GBS = 37950.0, 14230.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200 at N=1, 24 at N>1 where a step is a 4K 64-spp pass)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before (default 20 at N=1, 2 at N>1)")
    ap.add_argument("--pt-passes", type=int, default=3, help="timed passes of the path-trace leg (0 = skip the leg)")
    ap.add_argument("--pt-spp", type=int, default=PT_SPP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the large-scene legs (atrium 262k triangles, 10 M-triangle cloud; N=1 only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the measured configuration); gloo = rehearsal of the N>1 control flow on fewer GPUs than ranks")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 200 if a.gpus == 1 else 24
    if a.warmup is None:
        a.warmup = 20 if a.gpus == 1 else 2
    return a


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (nothing in this process has touched a GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_head_scene(textured=False):
    """Product-side scene construction: triangle array + Woop data + SAH BVH from the C++ host classes (no oracle)."""
    from pathtracer_amd import hostapi, scene
    mesh = scene.load_ptmesh()
    woop = hostapi.woop_from_positions(mesh["pos"])
    nodes, lst, stats = hostapi.bvh_build(mesh["pos"], sah=True)
    textures, index, labels = scene.head_textures(mesh) if textured else ([], None, {})
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=woop,
              materials=scene.mesh_materials(mesh, textured=textured, texture_index=index), textures=textures, nodes=nodes, list=lst)
    sd["texture_labels"] = labels
    return sd, stats


def algorithmic_bytes(ctr, rays):
    """B_ray = 64 B per FlatNode visit + 52 B per Woop test + 28 B ray in + 16 B hit out."""
    return 64 * ctr["node_visits"] + 52 * ctr["tri_tests"] + 44 * rays


COUNTERS_SMALL = "profiles/traffic.json"      # head.obj kernels: rocprofv3 --pmc passes over tools/profile_kernels.py
COUNTERS_BIG = "profiles/traffic_big.json"    # atrium / 10 M cloud: rocprofv3 --pmc passes over tools/bench_bigscenes.py


def pmc_profile(which=COUNTERS_SMALL):
    """Per-kernel counters of COMMITTED rocprofv3 passes (written by tools/summarize_profiles.py from separate --pmc runs of the same
    launches as this bench): HBM bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes, + WRITE_SIZE), L2 hit rate,
    VALU utilisation.  Replayed, not measured by this run: every roofline object says so in `counters_from`."""
    f = os.path.join(ROOT, which)
    return json.load(open(f)) if os.path.exists(f) else {}


def roofline(kernel, alg_bytes, ms, prof, note, counters_from=COUNTERS_SMALL):
    """The bound that binds a traversal kernel is the cache-side gather of 64-byte records (the scene is L2-resident), so `peak` is the
    guide's aggregate L2 rate; the HBM view the contract names is kept beside it: `hbm` = counter bytes per launch / time / 8 TB/s.
    Live in this run: `achieved`, `frac`, `ms_per_launch`, `algorithmic_bytes_per_launch`.  Replayed from `counters_from`: the rest."""
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    traffic = prof.get("hbm_bytes_per_launch")
    out = {"kernel": kernel, "bound": "l2-gather", "achieved": achieved, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / L2_GATHER_PEAK_GBS, "traffic": traffic, "algorithmic_GBps": achieved, "ms_per_launch": ms,
           "algorithmic_bytes_per_launch": alg_bytes,
           "live": ["achieved", "frac", "algorithmic_GBps", "ms_per_launch", "algorithmic_bytes_per_launch"],
           "counters_from": (counters_from + " (" + str(prof.get("source_tag", "committed rocprofv3 --pmc passes of the same launches, not this run")) + ")")
           if prof else None,
           "hbm": {"achieved": (traffic / (ms * 1e-3) / 1e9) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                   "algorithmic_over_hbm_peak": achieved / HBM_PEAK_GBS},
           "l2_hit": prof.get("l2_hit_rate"), "valu_util": prof.get("valu_active_frac_of_wave_cycles"),
           "valu_lane_util": prof.get("valu_lane_utilisation"),
           # vector memory path (texture-data / texture-address units' busy cycles over the kernel's cycles, rocprofv3 --pmc): the unit
           # that is closest to saturation in every traversal kernel here
           "td_busy": prof.get("td_busy_frac"), "ta_busy": prof.get("ta_busy_frac"),
           "measured_gather_ceiling": {"coherent": GATHER_COHERENT_GBS, "incoherent": GATHER_INCOHERENT_GBS, "unit": "GB/s",
                                       "frac_of_coherent": achieved / GATHER_COHERENT_GBS,
                                       "from": "profiles/r01_n_l1_gather_peak.jsonl (tools/l1_gather_peak.hip on this GPU model)"},
           "note": note}
    return out


def host_info():
    """What the CPU baselines ran on: the machine's logical CPUs, the CPUs this job may use, the CPU model, and the thread count used.
    The GPU box is a share of a larger machine: os.cpu_count() reports the whole machine, the job's share is what the harness exports as
    OMP_NUM_THREADS (16 per GPU); the baselines use that share, not the machine."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = share if share > 0 else (affinity or os.cpu_count() or 1)
    return {"host_cores": os.cpu_count(), "affinity_cores": affinity, "cpu_model": model, "threads_used": threads,
            "threads_reason": ("OMP_NUM_THREADS=%d: the job's CPU share on a shared machine" % share) if share > 0 else "every CPU the process may run on"}


DATA_NOTE = ("head.obj and three of its six 1600x1600 maps are the reference's own assets (assets/SHA256SUMS); the three maps the reference "
             "checkout lacks and the large scenes (262k-triangle atrium for Crytek Sponza, 10 M-triangle cloud) are seeded procedural "
             "stand-ins; cameras, lights and sample sequences are the build's own (SURVEY.md 8d)")


def config1_leg(ctx, stream, dev, red_dev, rank, world, steps, warmup):
    """BASELINE configs[1] on every rank (its own frame: with N > 1 ranks only meet in the timing barriers and the ray-count sum)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from pathtracer_amd import capi, scene
    sd, bvh_stats = build_head_scene()
    gscene = capi.Scene(ctx, sd)
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    n = W * H
    f32 = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f32) for _ in range(4))
    t, u, v = (torch.empty(n, **f32) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731

    def four_launch(ev=None):
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

    def step(ev=None):
        if ev:
            ev[0].record(stream)
        gscene.trace_frame_dev(cam, light, W, H, P(hid), P(t), P(u), P(v), P(sid))
        if ev:
            ev[1].record(stream)

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
    bytes_primary, bytes_shadow = algorithmic_bytes(cp, n), algorithmic_bytes(cs, n)

    for _ in range(warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(events[k])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total_rays = torch.tensor([rays_per_step * steps], dtype=torch.int64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_rays, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_rays = int(total_rays.item())
    frame_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    # the same frame as four launches (the public raycast entry points with ray arrays in HBM), outside the timed region: per-kernel times
    ev4 = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(min(steps, 50))]
    for _ in range(5):
        four_launch()
    torch.cuda.synchronize(dev)
    t4 = time.perf_counter()
    for e in ev4:
        four_launch(e)
    torch.cuda.synchronize(dev)
    four_ms = (time.perf_counter() - t4) / len(ev4) * 1e3
    closest_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev4]))
    anyhit_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev4]))
    gscene.close()
    prof = pmc_profile()
    for k in ("k_trace_frame", "k_raycast_closest", "k_raycast_anyhit"):
        if k in prof:
            prof[k] = dict(prof[k], source_tag=prof.get("source", ""))
    note = ("algorithmic bytes (64 B/node visit + 52 B/triangle test + 44 B/ray, SURVEY.md 8d) of one launch / HIP-event time of the launch.  "
            "head.obj's 1.6 MB of records never leave L2 (`l2_hit`; `traffic` = what reaches the fabric: the hit records out), so the HBM "
            "roofline does not bind; the record gather from cache does: `peak` = the guide's aggregate L2 rate.")
    dominant = roofline("k_trace_frame (primary closest-hit + shadow any-hit, one launch)", bytes_primary + bytes_shadow, frame_ms,
                        prof.get("k_trace_frame", {}), note + "  One launch processes 2,073,600 primary rays and one shadow ray per hit.")
    other = [roofline("k_raycast_step<closest-hit> (four-launch form)", bytes_primary, closest_ms, prof.get("k_raycast_closest", {}),
                      "2,073,600 primary rays from HBM ray arrays"),
             roofline("k_raycast_step<any-hit> (four-launch form)", bytes_shadow, anyhit_ms, prof.get("k_raycast_anyhit", {}),
                      "2,073,600 shadow rays from HBM ray arrays (rays of missed pixels are empty); rays end at different depths (`valu_lane_util`)")]
    return {"sd": sd, "cam": cam, "light": light,
            "Mrays_per_s": total_rays / elapsed / 1e6, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "workload": "head.obj 1920x1080 primary closest-hit + shadow any-hit rays (BASELINE configs[1]), one fused launch per frame",
            "triangles": int(sd["mat_id"].shape[0]), "bvh_nodes": int(sd["nodes"].shape[0]), "rays_per_step_per_gpu": rays_per_step,
            "kernels_ms": {"trace_frame": frame_ms},
            "four_launch": {"ms_per_frame": four_ms, "Mrays_per_s": rays_per_step / (four_ms * 1e-3) / 1e6,
                            "kernels_ms": {"raycast_closest": closest_ms, "raycast_anyhit": anyhit_ms},
                            "note": "gen_primary -> raycast(closest) -> gen_shadow -> raycast(any-hit) through the public raycast entry points, "
                                    "ray arrays in HBM; same outputs"},
            "per_ray": {"primary_nodes": cp["node_visits"] / n, "primary_tris": cp["tri_tests"] / n,
                        "shadow_nodes": cs["node_visits"] / n, "shadow_tris": cs["tri_tests"] / n},
            "roofline": dominant, "roofline_other_kernels": other}


C3_W, C3_H, C3_PASS_SPP = 3840, 2160, 64


def build_atrium_scene():
    from pathtracer_amd import hostapi, scene
    m = scene.atrium_scene()
    t0 = time.perf_counter()
    nodes, lst, st = hostapi.bvh_build(m["pos"], True)
    build_s = time.perf_counter() - t0
    sd = dict(pos=m["pos"], nrm=m["nrm"], uv=m["uv"], mat_id=m["mat_id"], woop=hostapi.woop_from_positions(m["pos"]),
              materials=m["materials"], textures=[], nodes=nodes, list=lst)
    return sd, build_s


def sharded_frame_leg(args, ctx, stream, dev, red_dev, rank, world, steps, warmup):
    """BASELINE configs[3]: ONE 3840x2160 frame of the 262k-triangle scene, blocks dealt round-robin over the ranks, the tiles exchanged
    after every pass inside the timed region.  A step = one pass of C3_PASS_SPP samples per pixel + the exchange."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from pathtracer_amd import capi, multigpu, scene
    w, h = C3_W, C3_H
    sd, build_s = build_atrium_scene()
    gscene = capi.Scene(ctx, sd)
    cam, lights = scene.atrium_view(w, h)
    accum = torch.zeros(h * w * 4, dtype=torch.float32, device=dev)
    if args.backend == "nccl":
        tg = multigpu.DeviceTileGather(ctx, w, h, rank, world)  # ptgpu_gather_tiles_dev: pack + ncclAllGather + unpack on the stream
    else:
        tg = multigpu.TileGather(w, h, rank, world, dev, ctx=ctx)  # rehearsal: device pack / unpack around a gloo all_gather

    def params(seed, spp=C3_PASS_SPP):
        return scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=spp, sobol_block=True, termination_prob=0.2, pass_seed=seed)

    # traversal work per ray for the roofline of the integrator kernel: one counted 1-spp pass over this rank's blocks
    ctx.enable_counters(True)
    ctx.reset_counters()
    gscene.render_pass_dev(params(1000, 1), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
    c = ctx.counters()
    ctx.enable_counters(False)
    bytes_per_ray = algorithmic_bytes(c, c["rays"]) / max(c["rays"], 1)
    # the same pass on ONE GPU (rank 0 renders every block, the others wait): the base of the strong-scaling figure, same job, same box
    base = None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    if world > 1:
        dist.barrier()
    if rank == 0:
        accum.zero_()
        gscene.render_pass_dev(params(998), cam, lights, accum.data_ptr())  # sizes the workspace
        ctx.reset_counters()
        ev[0].record(stream)
        gscene.render_pass_dev(params(997), cam, lights, accum.data_ptr())
        ev[1].record(stream)
        torch.cuda.synchronize(dev)
        ms1 = ev[0].elapsed_time(ev[1])
        r1 = ctx.counters()["rays"]
        base = {"ms_per_pass": ms1, "Mrays_per_s": r1 / (ms1 * 1e-3) / 1e6, "samples_per_s": w * h * C3_PASS_SPP / (ms1 * 1e-3),
                "measured": "rank 0 alone, every block of the same frame, one pass, before the timed region of this job (no exchange)"}
    if world > 1:
        dist.barrier()
    accum.zero_()
    for k in range(warmup):
        gscene.render_pass_dev(params(900 + k), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
        tg.gather(accum)
    accum.zero_()
    ctx.reset_counters()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        evs[k][0].record(stream)
        gscene.render_pass_dev(params(k), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
        evs[k][1].record(stream)
        tg.gather(accum)
        evs[k][2].record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    local_rays = ctx.counters()["rays"]
    rays = torch.tensor([local_rays], dtype=torch.int64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed, rays = float(tmax.item()), int(rays.item())
    render_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    gather_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    st = ctx.spill_status()
    if isinstance(tg, multigpu.DeviceTileGather):
        tg.close()
    gscene.close()
    del accum
    samples = w * h * C3_PASS_SPP * steps
    prof = pmc_profile(COUNTERS_BIG).get("atrium_262k", {})
    kprof = next((v for k, v in prof.items() if k.startswith("k_integrate")), {})
    if kprof:
        kprof = dict(kprof, source_tag=prof.get("source", ""))
        if kprof.get("hbm_bytes_per_launch"):  # the profiled launch was a 16-spp full-frame pass on one GPU: scale to this rank's share of this pass
            kprof["hbm_bytes_per_launch"] = kprof["hbm_bytes_per_launch"] * (C3_PASS_SPP / 16.0) / world
    rl = roofline("k_integrate (+ k_accum_sobol) on rank 0", bytes_per_ray * (local_rays / steps), render_ms, kprof,
                  "algorithmic traversal bytes per ray (counted 1-spp pass) x this rank's rays of one pass / HIP-event time of its render launches; "
                  "bounce rays are incoherent, so the per-lane gather ceiling that applies is `measured_gather_ceiling.incoherent`; the counters "
                  "say the texture-data path is ~94 % busy on this scene", COUNTERS_BIG)
    rl["frac_of_incoherent_ceiling"] = rl["achieved"] / GATHER_INCOHERENT_GBS
    return {"workload": "262k-triangle atrium (procedural stand-in for Crytek Sponza, BASELINE configs[3]) path trace %dx%d, one frame sharded by "
                        "32x32 blocks over %d GPU(s), Sobol-block AA, Russian roulette after 1 bounce (p = 0.2), 2 area lights; step = one pass of "
                        "%d spp + tile exchange (16 passes = 1024 spp)" % (w, h, world, C3_PASS_SPP),
            "triangles": int(sd["mat_id"].shape[0]), "bvh_nodes": int(sd["nodes"].shape[0]), "host_sah_build_s": build_s,
            "tile_gather": (("ptgpu_gather_tiles_dev: ncclAllGather (RCCL over xGMI) of packed 32x32-block tiles + out-of-block samples"
                             if args.backend == "nccl" else "device pack/unpack + gloo all_gather of host copies (rehearsal, not RCCL)") if world > 1 else "none"),
            "slice_MB_per_rank": multigpu.slice_floats(w, h, world) * 4 / 1e6,
            "Mrays_per_s": rays / elapsed / 1e6, "samples_per_s": samples / elapsed, "rays_per_sample": rays / samples,
            "ms_per_step": elapsed / steps * 1e3, "render_ms_per_pass_rank0": render_ms, "gather_ms_per_pass_rank0": gather_ms,
            "steps": steps, "warmup": warmup, "dropped_samples": st["dropped"],
            "per_ray": {"nodes": c["node_visits"] / max(c["rays"], 1), "tris": c["tri_tests"] / max(c["rays"], 1), "bytes": bytes_per_ray},
            "strong_scaling_base": base, "roofline": rl}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    from pathtracer_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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

    ctx = capi.Context(dev_index)
    # a real (non-null) stream shared by the library's launches and torch's events: torch.cuda.Event only sees the stream
    # it is recorded on, and a NULL handle would make the library fall back to its own stream
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    if world == 1:
        c1 = config1_leg(ctx, stream, dev, red_dev, rank, world, args.steps, args.warmup)
        sharded = None
    else:
        sharded = sharded_frame_leg(args, ctx, stream, dev, red_dev, rank, world, args.steps, args.warmup)
        c1 = config1_leg(ctx, stream, dev, red_dev, rank, world, 100, 10)
    pt = path_trace_leg(args, ctx, stream, dev, red_dev, rank, world) if args.pt_passes > 0 else None
    extra = extra_legs(ctx, stream, dev) if (world == 1 and not args.no_extra) else None

    if rank == 0:
        sd, cam, light = c1.pop("sd"), c1.pop("cam"), c1.pop("light")
        common = {"metric": "Mrays/s", "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                  "vs_baseline": None, "dtype": "f32", "data": "reference assets (head.obj + 3 of its texture maps) and synthetic stand-ins for what the reference checkout lacks", "data_note": DATA_NOTE,
                  "counters_note": "fields fed by hardware counters (traffic, hbm, l2_hit, valu_*, td_busy, ta_busy) are replayed from the committed "
                                   "rocprofv3 --pmc summaries each roofline names in `counters_from`; times, rates and algorithmic bytes are live"}
        if world == 1:
            out = dict(common, value=c1["Mrays_per_s"], ms_per_step=c1["ms_per_step"], scaling="weak",
                       config={"workload": c1["workload"], "triangles": c1["triangles"], "bvh_nodes": c1["bvh_nodes"],
                               "rays_per_step_per_gpu": c1["rays_per_step_per_gpu"], "parallelism": "tiles x1"},
                       samples_per_s=pt["samples_per_s"] if pt else None, kernels_ms=c1["kernels_ms"], four_launch=c1["four_launch"],
                       per_ray=c1["per_ray"], roofline=c1["roofline"], roofline_other_kernels=c1["roofline_other_kernels"])
        else:
            out = dict(common, value=sharded["Mrays_per_s"], ms_per_step=sharded["ms_per_step"], scaling="strong",
                       config={"workload": sharded["workload"], "triangles": sharded["triangles"], "bvh_nodes": sharded["bvh_nodes"],
                               "parallelism": "tiles x%d, one frame, RCCL tile gather per pass" % world if args.backend == "nccl"
                               else "tiles x%d, one frame, gloo rehearsal of the gather" % world},
                       samples_per_s=sharded["samples_per_s"], roofline=sharded.pop("roofline"), sharded_frame=sharded,
                       value_note="N>1: one frame of BASELINE configs[3] strong-scaled over the ranks with the tile exchange in the timed region; the "
                                  "N=1 line's `value` is configs[1] (a different workload): compare with `sharded_frame.strong_scaling_base` "
                                  "(this job) or with the N=1 line's `extra[0].pass_64spp`",
                       independent_frames=dict(c1, scaling="weak", note="configs[1], every rank its own frame, no exchange: N x by construction"))
        if pt:
            out["path_trace"] = pt
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, cam, light)
            if pt:
                out["path_trace"]["cpu_baseline"] = cpu_baseline_path_trace(args.pt_spp)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def path_trace_leg(args, ctx, stream, dev, red_dev, rank, world):
    """configs[2]: full path trace of the 1080p head frame; blocks dealt round-robin over ranks, tiles gathered over RCCL."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from pathtracer_amd import capi, multigpu, scene
    sd, _ = build_head_scene(textured=True)
    gscene = capi.Scene(ctx, sd)
    cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
    if args.backend == "nccl":
        tg = multigpu.DeviceTileGather(ctx, W, H, rank, world)  # ptgpu_gather_tiles_dev: pack + ncclAllGather + unpack on the stream
    else:
        tg = multigpu.TileGather(W, H, rank, world, dev, ctx=ctx)  # rehearsal: device pack / unpack around a gloo all_gather

    def params(seed, spp=args.pt_spp):
        return scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=False, bilinear=True, normal_mapped=True,
                                 filter_width=1.5, termination_prob=0.2, pass_seed=seed)

    # traversal work per ray for the roofline of the integrator kernel: one counted low-spp pass
    ctx.enable_counters(True)
    ctx.reset_counters()
    gscene.render_pass_dev(params(1000, 4), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)
    c = ctx.counters()
    ctx.enable_counters(False)
    bytes_per_ray = algorithmic_bytes(c, c["rays"]) / max(c["rays"], 1)
    accum.zero_()
    gscene.render_pass_dev(params(999), cam, lights, accum.data_ptr(), block_begin=rank, block_stride=world)  # warm-up (sizes the workspace)
    tg.gather(accum)
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
    dropped = ctx.spill_status()["dropped"]
    if isinstance(tg, multigpu.DeviceTileGather):
        tg.close()
    gscene.close()
    samples = W * H * args.pt_spp * args.pt_passes
    allprof = pmc_profile()
    prof = allprof.get("k_integrate", {})
    if prof:
        prof = dict(prof, source_tag=allprof.get("source", ""))
    spp_scale = args.pt_spp / float(prof.get("spp", args.pt_spp)) if prof else 1.0
    if prof.get("hbm_bytes_per_launch"):
        prof = dict(prof, hbm_bytes_per_launch=prof["hbm_bytes_per_launch"] * spp_scale)
    rl = roofline("k_integrate (+ k_accum_gauss)", bytes_per_ray * (local_rays / args.pt_passes), render_ms, prof,
                  "algorithmic traversal bytes per ray x rays of one pass / HIP-event time of the pass (integrator + accumulation launches); bounce "
                  "rays are incoherent, so the per-lane gather ceiling that applies is `measured_gather_ceiling.incoherent`.  `traffic` = counter "
                  "bytes of the profiled %s-spp launch scaled to %d spp: per-sample radiance written once (16 B/sample) and read once by the "
                  "accumulation, plus texel gathers from the 61 MB of maps that miss the 4 MB L2 (served by the Infinity Cache)" %
                  (prof.get("spp", "?"), args.pt_spp))
    rl["frac_of_incoherent_ceiling"] = rl["achieved"] / GATHER_INCOHERENT_GBS
    labels = sd["texture_labels"]
    return {"workload": "head.obj full path trace %dx%d, %d spp, Gaussian AA (filter 1.5), bilinear textures + normal maps, RR after 1 bounce p=0.2 "
                        "(BASELINE configs[2]); maps 1600x1600: %s" % (W, H, args.pt_spp, ", ".join("%s = %s" % kv for kv in sorted(labels.items()))),
            "passes": args.pt_passes, "scaling": "strong" if world > 1 else "n/a",
            "tile_gather": (("ptgpu_gather_tiles_dev: ncclAllGather (RCCL) of packed 32x32-block tiles" if args.backend == "nccl" else
                             "device pack/unpack + gloo all_gather (rehearsal)") if world > 1 else "none"),
            "samples_per_s": samples / elapsed, "Mrays_per_s": rays / elapsed / 1e6, "rays_per_sample": rays / samples,
            "ms_per_pass": elapsed / args.pt_passes * 1e3, "render_ms_per_pass_rank0": render_ms, "dropped_samples": dropped,
            "per_ray": {"nodes": c["node_visits"] / max(c["rays"], 1), "tris": c["tri_tests"] / max(c["rays"], 1), "bytes": bytes_per_ray},
            "roofline": rl}


def extra_legs(ctx, stream, dev):
    """Large-scene stand-ins on one GPU (N=1 only): BASELINE configs[3] (Crytek Sponza is not in the reference checkout:
    scene.atrium_scene, ~262k triangles, 3840x2160; N>1 runs make its sharded frame the headline) and configs[4]
    (scene.tetra_cloud_scene, 10 M triangles, 1080p, depth of field).  Primary-ray frame + one path-trace pass each, with the roofline of
    the primary-ray kernel: algorithmic bytes live, hardware counters replayed from the round-3 rocprofv3 passes of the same scenes."""
    import torch

    from pathtracer_amd import capi, hostapi, scene
    out = []
    big = pmc_profile(COUNTERS_BIG)
    for name, key, make, view, (w, h), spp, dof in (
            ("atrium 262k triangles 3840x2160 (procedural stand-in for BASELINE configs[3] Sponza)", "atrium_262k", scene.atrium_scene, scene.atrium_view,
             (C3_W, C3_H), C3_PASS_SPP, False),
            ("tetra cloud 10 M triangles 1920x1080 + depth of field (BASELINE configs[4])", "tetra_cloud_10M", lambda: scene.tetra_cloud_scene(10_000_000),
             scene.tetra_cloud_view, (1920, 1080), 16, True)):
        m = make()
        t0 = time.perf_counter()
        nodes, lst, st = hostapi.bvh_build(m["pos"], True)
        build_s = time.perf_counter() - t0
        sd = dict(pos=m["pos"], nrm=m["nrm"], uv=m["uv"], mat_id=m["mat_id"], woop=hostapi.woop_from_positions(m["pos"]),
                  materials=m["materials"], textures=[], nodes=nodes, list=lst)
        s = capi.Scene(ctx, sd)
        cam, lights = view(w, h)
        n = w * h
        o, d = (torch.empty(n * 3, dtype=torch.float32, device=dev) for _ in range(2))
        t, u, v = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3))
        hid = torch.empty(n, dtype=torch.int32, device=dev)
        P = lambda x: x.data_ptr()  # noqa: E731
        ctx.gen_primary_dev(cam, w, h, P(o), P(d))
        ctx.enable_counters(True)
        ctx.reset_counters()
        s.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        c = ctx.counters()
        ctx.enable_counters(False)
        for _ in range(3):
            s.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 10
        ev[0].record(stream)
        for _ in range(reps):
            s.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
        ev[1].record(stream)
        torch.cuda.synchronize(dev)
        ms = ev[0].elapsed_time(ev[1]) / reps
        alg = algorithmic_bytes(c, n)
        accum = torch.zeros(n * 4, dtype=torch.float32, device=dev)
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=spp, sobol_block=True, depth_of_field=dof, aperture=0.01,
                                focal_distance=1.2, pass_seed=1)
        s.render_pass_dev(prm, cam, lights, P(accum))  # warm-up
        ctx.reset_counters()
        ev[0].record(stream)
        s.render_pass_dev(prm, cam, lights, P(accum))
        ev[1].record(stream)
        torch.cuda.synchronize(dev)
        pt_ms = ev[0].elapsed_time(ev[1])
        rays = ctx.counters()["rays"]
        prof = big.get(key, {})
        # the closest-hit kernel this scene's size selects (csrc/ptgpu.hip launch_raycast): persistent waves with a 12-entry LDS stack
        # beyond 64 MB of records, one ray per lane below
        kname = next((k for pre in ("k_raycast_persist<false, false, false, 12", "k_raycast_step<false") for k in prof if k.startswith(pre)), None)
        kprof = dict(prof[kname], source_tag=prof.get("source", "")) if kname else {}
        rl = roofline("%s (closest-hit), %d primary rays" % (kname or "raycast kernel", n), alg, ms, kprof,
                      "primary rays of the frame, one launch.  Algorithmic bytes count every lane's 64-byte record although the lanes of a coherent "
                      "wave share records (one fetch serves many lanes), so `achieved` can exceed what any data path moved; what binds is in the "
                      "counters: `td_busy` (texture-data path) and `hbm.frac` (`traffic` / time / 8 TB/s)", COUNTERS_BIG)
        out.append({"workload": name, "triangles": int(m["pos"].shape[0]), "bvh_nodes": int(nodes.shape[0]), "host_sah_build_s": build_s,
                    "ray_length_probe": s.ray_length(),
                    "primary_Mrays_per_s": n / (ms * 1e-3) / 1e6, "primary_ms": ms,
                    "per_ray": {"nodes": c["node_visits"] / n, "tris": c["tri_tests"] / n, "bytes": alg / n},
                    "primary_algorithmic_GBps": alg / (ms * 1e-3) / 1e9, "roofline": rl,
                    ("pass_%dspp" % spp): {"spp": spp, "samples_per_s": n * spp / (pt_ms * 1e-3), "Mrays_per_s": rays / (pt_ms * 1e-3) / 1e6, "ms": pt_ms,
                                           "note": "one Sobol-block pass over the whole frame on this GPU" +
                                                   (": the N=1 base of the N>1 headline (bench.py --gpus N)" if key == "atrium_262k" else "")}})
        s.close()
        del o, d, t, u, v, hid, accum, sd, m
    return out


def cpu_baseline(sd, cam, light):
    """The oracle (CPU restatement of the reference's raycast) on the same C2 frame, on this box's host cores."""
    from oracle import pyoracle as po
    hi = host_info()
    threads = hi["threads_used"]
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
    return dict(hi, value=rays / best / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample="the full configs[1] frame (2,073,600 primary + one shadow ray per hit), best of 3, the two raycast batches timed")


def cpu_baseline_path_trace(spp):
    """The oracle's multithreaded pathTraceBlock loop (libm transcendentals, as the reference calls them) on a bounded sample of the SAME
    workload: configs[2] itself (1920x1080, same maps, Gaussian AA, normal maps, RR) on every 24th 32x32 block = 85 of the frame's 2040
    blocks spread over the whole image."""
    from oracle import pyoracle as po

    from pathtracer_amd import scene
    hi = host_info()
    threads = hi["threads_used"]
    sd, _ = build_head_scene(textured=True)
    labels = sd.pop("texture_labels")
    prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=spp, sobol_block=False, bilinear=True, normal_mapped=True,
                            filter_width=1.5, termination_prob=0.2, pass_seed=0)
    cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
    stride = 24
    nblocks = ((W + 31) // 32) * ((H + 31) // 32)
    pixels = 0
    for b in range(0, nblocks, stride):
        bx, by = (b % ((W + 31) // 32)) * 32, (b // ((W + 31) // 32)) * 32
        pixels += min(32, W - bx) * min(32, H - by)
    t0 = time.perf_counter()
    _, ctr = po.render_pass(sd, prm, cam, lights, nthreads=threads, libm=True, block_begin=0, block_end=-1, block_stride=stride)
    dt = time.perf_counter() - t0
    return dict(hi, value=pixels * spp / dt, unit="samples/s", Mrays_per_s=ctr["rays"] / dt / 1e6, cores=threads, kind="port",
                sample="configs[2] itself (1920x1080, %d spp, Gaussian AA, same maps: %d reference files + stand-ins), every %dth 32x32 block: "
                       "%d blocks, %d samples, one pass" % (spp, sum(1 for l in labels.values() if l == "reference file"), stride,
                                                             len(range(0, nblocks, stride)), pixels * spp))


if __name__ == "__main__":
    main()
