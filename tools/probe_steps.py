#!/usr/bin/env python3
"""Traversal steps per ray (node visits + triangle tests, the library's counting variant) of random rays inside a scene's box: the
statistic the integrator's service threshold is chosen from (csrc/ptgpu.hip scene_probe).  python tools/probe_steps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pathtracer_amd import capi, hostapi, scene  # noqa: E402


def main():
    ctx = capi.Context(0)
    cases = [("head", scene.load_ptmesh()["pos"]), ("atrium", scene.atrium_scene()["pos"])]
    for n in (20_000, 100_000, 1_000_000):
        cases.append(("tetra %d" % n, scene.tetra_cloud_scene(n)["pos"]))
    for name, pos in cases:
        nodes, lst, _ = hostapi.bvh_build(pos, True)
        nt = pos.shape[0]
        sd = dict(pos=pos, nrm=np.zeros_like(pos), uv=np.zeros((nt, 3, 2), np.float32), mat_id=np.zeros(nt, np.int32),
                  woop=hostapi.woop_from_positions(pos), materials=[scene.make_material()], textures=[], nodes=nodes, list=lst)
        sc = capi.Scene(ctx, sd)
        lo, hi = pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0)
        rng = np.random.RandomState(1)
        n = 16384
        o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True) * np.linalg.norm(hi - lo)).astype(np.float32)
        ctx.enable_counters(True)
        out = {}
        for occ in (False, True):
            ctx.reset_counters()
            sc.raycast(o, d, occ)
            c = ctx.counters()
            out[occ] = (c["node_visits"] / n, c["tri_tests"] / n)
        ctx.enable_counters(False)
        print("%-16s closest: %.1f nodes + %.1f tris   any-hit: %.1f + %.1f" % (name, *out[False], *out[True]), flush=True)
        sc.close()


if __name__ == "__main__":
    main()
