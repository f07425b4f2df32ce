#!/usr/bin/env python3
"""A/B timing of the raycast kernel variants on the configs[1] frame, interleaved in ONE process (guide rule 24).
Run on the GPU box:  python tools/tune_raycast.py [rounds]   -> prints median/min ms per variant for closest and any-hit,
after checking that every variant returns identical hit ids."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    W, H = 1920, 1080
    n = W * H
    mesh = scene.load_ptmesh()
    nodes, lst, _ = hostapi.bvh_build(mesh["pos"], True)
    sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
              materials=scene.mesh_materials(mesh), textures=[], nodes=nodes, list=lst)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0)
    ctx.set_stream(stream.cuda_stream)
    sc = capi.Scene(ctx, sd)
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    f = dict(dtype=torch.float32, device=dev)
    o, d, so, sdir = (torch.empty(n * 3, **f) for _ in range(4))
    t, u, v = (torch.empty(n, **f) for _ in range(3))
    hid, valid, sid = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    P = lambda x: x.data_ptr()  # noqa: E731
    os.environ["PTGPU_RAYCAST_IMPL"], os.environ["PTGPU_RAYCAST_NLDS"] = "0", "16"
    ctx.gen_primary_dev(cam, W, H, P(o), P(d))
    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
    sc.gen_shadow_dev(light, n, P(o), P(d), P(hid), P(t), P(u), P(v), P(so), P(sdir), P(valid))
    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
    torch.cuda.synchronize()
    ref_hid, ref_t, ref_sid = hid.clone(), t.clone(), sid.clone()
    # incoherent rays: random origins in the room, random directions
    g = torch.Generator(device="cpu").manual_seed(1)
    ro = (torch.rand(n, 3, generator=g) * torch.tensor([2.8, 1.4, 0.9]) + torch.tensor([-1.4, -0.45, -0.45])).to(dev).contiguous()
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).mul(3.0).to(dev).contiguous()
    rid = torch.empty(n, dtype=torch.int32, device=dev)
    sc.raycast_dev(P(ro), P(rd), n, False, P(rid), None, None, None)
    torch.cuda.synchronize()
    ref_rid = rid.clone()
    variants = [(1, 24, 0, 1), (1, 16, 0, 1), (1, 12, 0, 1), (1, 8, 0, 1), (2, 24, 32, 1)]
    if len(sys.argv) > 2:  # "impl:nlds:refill:oversub,..."
        variants = [tuple(int(x) for x in v.split(":")) for v in sys.argv[2].split(",")]
    times = {vv: dict(closest=[], anyhit=[], random=[]) for vv in variants}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(rounds + 1):
        for vv in variants:
            os.environ["PTGPU_RAYCAST_IMPL"], os.environ["PTGPU_RAYCAST_NLDS"], os.environ["PTGPU_RAYCAST_REFILL"] = str(vv[0]), str(vv[1]), str(vv[2])
            os.environ["PTGPU_RAYCAST_OVERSUB"] = str(vv[3])
            for kind in ("closest", "anyhit", "random"):
                ev[0].record(stream)
                if kind == "closest":
                    sc.raycast_dev(P(o), P(d), n, False, P(hid), P(t), P(u), P(v))
                elif kind == "anyhit":
                    sc.raycast_dev(P(so), P(sdir), n, True, P(sid), None, None, None)
                else:
                    sc.raycast_dev(P(ro), P(rd), n, False, P(rid), None, None, None)
                ev[1].record(stream)
                torch.cuda.synchronize()
                if r == 0:
                    assert torch.equal(hid, ref_hid) and torch.equal(t, ref_t) and torch.equal(sid, ref_sid) and torch.equal(rid, ref_rid), vv
                else:
                    times[vv][kind].append(ev[0].elapsed_time(ev[1]))
    print("variant(impl,nlds,refill)  closest ms med/min   anyhit ms med/min   random ms med/min   [%d rays each]" % n)
    for vv in variants:
        row = ["%s" % (vv,)]
        for kind in ("closest", "anyhit", "random"):
            a = np.array(times[vv][kind])
            row.append("%.4f/%.4f" % (np.median(a), a.min()))
        print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
