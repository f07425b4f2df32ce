#!/usr/bin/env python3
"""Debug aid: one Sobol-block 32x32 block of the configs[3] frame (atrium, 3840x2160, 1024 spp) on the device and in the oracle;
where the accumulated pixels differ, the block's samples are traced one by one through ptgpu_trace_samples / oracle.trace_samples and the
pixel sums are re-formed sequentially in float32 from either set of per-sample radiances.  python tools/debug_block.py [spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from pathtracer_amd import capi, hostapi, scene  # noqa: E402

W, H = 3840, 2160
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BX = (W + 31) // 32


def mix(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def rng_key(k0, k1, k2, k3):
    u = lambda v: np.asarray(v, np.uint32)  # noqa: E731
    return mix(u(k3) + mix(u(k2) + mix(u(k1) + mix(u(k0) + np.uint32(0x9E3779B9)))))


def rng_at(key, n):
    return mix(key ^ mix(np.asarray(n, np.uint32) * np.uint32(0x9E3779B9) + np.uint32(0x7F4A7C15)))


def main():
    np.seterr(over="ignore")
    m = scene.atrium_scene()
    nodes, lst, _ = hostapi.bvh_build(m["pos"], True)
    sd = dict(pos=m["pos"], nrm=m["nrm"], uv=m["uv"], mat_id=m["mat_id"], woop=hostapi.woop_from_positions(m["pos"]), materials=m["materials"],
              textures=[], nodes=nodes, list=lst)
    cam, lights = scene.atrium_view(W, H)
    ctx = capi.Context(0)
    s = capi.Scene(ctx, sd)
    seed = 8
    prm = scene.make_params(width=W, height=H, bounces=-1, num_aa_rays=SPP, sobol_block=True, filter_width=1.5, termination_prob=0.2, pass_seed=seed)
    for (x, y) in [(61, 34), (30, 5), (85, 55), (119, 0), (64, 67)]:
        b = y * BX + x
        g = s.render_pass(prm, cam, lights, block_begin=b, block_end=b + 1)
        c, _ = po.render_pass(sd, prm, cam, lights, nthreads=1, block_begin=b, block_end=b + 1)
        diff = g.view(np.uint32) != c.view(np.uint32)
        print("block", (x, y), "differing words", int(diff.sum()), flush=True)
        if not diff.any():
            continue
        ys, xs, ch = np.nonzero(diff)
        for yy, xx, cc in list(zip(ys, xs, ch))[:12]:
            print("  pixel", (int(xx), int(yy)), "ch", int(cc), "gpu %.9g oracle %.9g" % (g[yy, xx, cc], c[yy, xx, cc]))
        # the block's samples, as pathTraceBlock forms them (oracle/pt_core.h path_trace_block, Sobol-block mode)
        bx, by = x * 32, y * 32
        bw, bh = min(32, W - bx), min(32, H - by)
        kb = rng_key(b, seed, 0xB10C, 0)
        mul = 1 + int(rng_at(kb, 0) % np.uint32(8)); add = 1 + int(rng_at(kb, 1) % np.uint32(12344))
        base = b * mul + add
        n = bw * bh * SPP
        i = np.arange(n, dtype=np.int64)
        sa = ctx.sobol_sample((base + i).astype(np.uint64), np.zeros(n, np.uint32))
        sb = ctx.sobol_sample((base + i).astype(np.uint64), np.ones(n, np.uint32))
        px = (np.float32(bx) + np.float32(bw) * sa).astype(np.float32)
        py = (np.float32(by) + np.float32(bh) * sb).astype(np.float32)
        div_x, div_y = np.float32(1.0) / np.float32(W), np.float32(1.0) / np.float32(H)
        cx = (px * div_x * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
        cy = (py * div_y * np.float32(-2.0) + np.float32(1.0)).astype(np.float32)
        xy = np.stack([cx, cy], 1)
        state = np.stack([rng_key(np.full(n, b), np.full(n, seed), i.astype(np.uint32), np.zeros(n)), np.zeros(n, np.uint32)], 1).astype(np.uint32)
        sbase = (base + i).astype(np.int32)
        g_rgb, g_ctr = s.trace_samples(prm, cam, lights, xy, sbase, state)
        c_rgb, c_ctr = po.trace_samples(sd, prm, cam, lights, xy, sbase, state, nthreads=8)
        bad = np.flatnonzero((g_rgb.view(np.uint32) != c_rgb.view(np.uint32)).any(1) | (g_ctr != c_ctr))
        print("  per-sample: %d of %d samples differ" % (bad.size, n))
        for k in bad[:8]:
            print("    sample", int(k), "xy", xy[k], "gpu", g_rgb[k], g_ctr[k], "oracle", c_rgb[k], c_ctr[k])
        pix = (np.minimum(np.floor(py).astype(np.int64), H - 1) * W + np.minimum(np.floor(px).astype(np.int64), W - 1))
        scale = np.float32(1.0) / np.float32(SPP)
        for name, rgb in (("gpu samples", g_rgb), ("oracle samples", c_rgb)):
            img = np.zeros((H * W, 4), np.float32)
            order = np.argsort(pix, kind="stable")
            ps, rs = pix[order], rgb[order]
            starts = np.flatnonzero(np.r_[True, ps[1:] != ps[:-1]])
            ends = np.r_[starts[1:], ps.size]
            for a, e in zip(starts, ends):
                v = np.concatenate([rs[a:e] * scale, np.full((e - a, 1), scale, np.float32)], 1).astype(np.float32)
                img[ps[a]] = np.cumsum(v, axis=0, dtype=np.float32)[-1]
            img = img.reshape(H, W, 4)
            print("  sequential float32 sums of the %s: %d words differ from the device image, %d from the oracle image" % (
                name, int((img.view(np.uint32) != g.view(np.uint32)).sum()), int((img.view(np.uint32) != c.view(np.uint32)).sum())))
    s.close()
    ctx.close()


if __name__ == "__main__":
    main()
