"""ptgpu_raycast_batch_dev_noalloc (include/ptgpu.h): the entry point that allocates nothing.  Its hit records must equal
ptgpu_raycast_batch_dev's bit for bit -- on head.obj (no deferral: only the counter words of the fixed scratch are used), on a
49-level tree (deferred-ray list + global-memory stack overflow inside the scratch) and with the alpha-cutout variant (bounded
persistent grid) -- on two calls in a row, the second with a smaller input, so that what the first call left in the scratch
(deferred count, list entries, stack slots) is seen to do no harm.  Collected last (test_zz_*): it shares a process with every
other GPU test and must pass after them as well as alone."""
import numpy as np
import pytest

from pathtracer_amd import capi, scene

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo, hi, length):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= (length / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d.astype(np.float32)


def _both(ctx, s, o, d, occlusion, use_alpha=True):
    """(existing entry point, allocation-free entry point) on the same device-resident rays."""
    n = o.shape[0]
    d_o, d_d = ctx.to_device(o), ctx.to_device(d)
    out = []
    try:
        for fn in (s.raycast_dev, s.raycast_dev_noalloc):
            bufs = [ctx.malloc(n * 4) for _ in range(4)]
            for b in bufs:
                capi._check(ctx.lib.ptgpu_memset(ctx.handle, b, 0xEE, n * 4))
            fn(d_o, d_d, n, occlusion, *bufs, use_alpha=use_alpha)
            ctx.sync()
            out.append(dict(id=ctx.d2h(bufs[0], n, np.int32), t=ctx.d2h(bufs[1], n, np.float32), u=ctx.d2h(bufs[2], n, np.float32),
                            v=ctx.d2h(bufs[3], n, np.float32)))
            for b in bufs:
                ctx.free(b)
    finally:
        ctx.free(d_o)
        ctx.free(d_d)
    return out


def _assert_equal(a, b, what):
    assert np.array_equal(a["id"], b["id"]), "%s: %d hit ids differ" % (what, int((a["id"] != b["id"]).sum()))
    for k in ("t", "u", "v"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), "%s: %s not bit-exact" % (what, k)


def _chain_scene(n=60, ratio=2.0, reps=3, seed=0):
    rng = np.random.RandomState(seed)
    tris = []
    for i in range(n):
        sc = ratio ** i
        for _ in range(reps):
            c = np.array([sc, rng.uniform(-0.2, 0.2) * sc, rng.uniform(-0.2, 0.2) * sc])
            tris.append(c + rng.uniform(-0.15, 0.15, (3, 3)) * sc)
    return np.asarray(tris, np.float32)


def test_noalloc_matches_on_head_two_calls(gpu_ctx, head, oracle):
    s = capi.Scene(gpu_ctx, head["sd"])
    try:
        for n, seed in ((300_000, 31), (70_001, 32)):  # the second call is smaller
            o, d = _rays(n, seed, [-1.4, -0.45, -0.45], [1.4, 0.95, 0.45], 3.0)
            for occ in (False, True):
                ref, got = _both(gpu_ctx, s, o, d, occ)
                assert (ref["id"] >= 0).mean() > 0.3
                _assert_equal(got, ref, "head n=%d occlusion=%s" % (n, occ))
        # and against the oracle once, so that "equal" cannot mean "equally wrong"
        c = oracle.raycast(head["sd"], o, d, False, nthreads=8)
        _assert_equal(_both(gpu_ctx, s, o, d, False)[1], c, "head vs oracle")
    finally:
        s.close()


def test_noalloc_deep_tree_defers_into_the_scratch(gpu_ctx, oracle):
    """49-level tree: the one-ray-per-lane kernel defers rays into the scratch's list, the persistent kernel's stacks overflow into
    the scratch's stack region (grid clamped to 64 workgroups)."""
    pos = _chain_scene()
    nodes, lst, st = oracle.bvh_build(pos, False)
    assert st["true_depth"] > 40
    sd = dict(pos=pos, nrm=scene.flat_normals(pos), uv=np.zeros((pos.shape[0], 3, 2), np.float32), mat_id=np.zeros(pos.shape[0], np.int32),
              woop=oracle.woop_from_positions(pos), materials=[scene.make_material()], textures=[], nodes=nodes, list=lst)
    s = capi.Scene(gpu_ctx, sd)
    try:
        for n, seed in ((120_000, 4), (9_999, 5)):
            rng = np.random.RandomState(seed)
            o = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], 1).astype(np.float32)
            ts = (2.0 ** rng.uniform(0, 59, n)).astype(np.float32)
            tgt = np.stack([ts, rng.uniform(-0.3, 0.3, n) * ts, rng.uniform(-0.3, 0.3, n) * ts], 1).astype(np.float32)
            d = ((tgt - o) * np.float32(1.5)).astype(np.float32)
            for occ in (False, True):
                ref, got = _both(gpu_ctx, s, o, d, occ)
                assert (ref["id"] >= 0).mean() > 0.2
                _assert_equal(got, ref, "chain n=%d occlusion=%s" % (n, occ))
        c = oracle.raycast(sd, o, d, False, nthreads=8)
        _assert_equal(_both(gpu_ctx, s, o, d, False)[1], c, "chain vs oracle")
    finally:
        s.close()


def test_noalloc_alpha_variant_and_limits(gpu_ctx, soup):
    rng = np.random.RandomState(21)
    tex = np.zeros((64, 64, 4), np.uint8)
    tex[..., 1] = (rng.rand(64, 64) > 0.5) * 255
    sd = dict(soup["sd"])
    sd["materials"] = [scene.make_material(tex_alpha=0)]
    sd["textures"] = [tex]
    s = capi.Scene(gpu_ctx, sd)
    try:
        for n, seed in ((50_000, 6), (1_234, 7)):
            o, d = _rays(n, seed, [-1.5, -1.5, -1.0], [1.5, 1.5, 1.0], 4.0)
            for occ in (False, True):
                ref, got = _both(gpu_ctx, s, o, d, occ, use_alpha=True)
                _assert_equal(got, ref, "alpha n=%d occlusion=%s" % (n, occ))
        # sizes beyond the fixed scratch are refused with a message, before anything is launched
        with pytest.raises(capi.PtgpuError, match="at most"):
            s.raycast_dev_noalloc(1, 1, capi.NOALLOC_MAX_RAYS + 1, False, None, None, None, None)
        s.raycast_dev_noalloc(None, None, 0, False, None, None, None, None)  # empty batch: nothing to do
    finally:
        s.close()


def test_noalloc_allocates_no_device_memory(gpu_ctx, oracle, soup):
    """The property the entry point exists for: free device memory (hipMemGetInfo, through ptgpu_mem_info) is the same before
    and after its calls -- on the 49-level tree, where ptgpu_raycast_batch_dev would size a deferred-ray list and overflow stacks, and on
    the alpha variant -- while the allocating entry point on a FRESH context, same rays, is seen to take memory (so the probe can see it)."""
    pos = _chain_scene()
    nodes, lst, _ = oracle.bvh_build(pos, False)
    sd = dict(pos=pos, nrm=scene.flat_normals(pos), uv=np.zeros((pos.shape[0], 3, 2), np.float32), mat_id=np.zeros(pos.shape[0], np.int32),
              woop=oracle.woop_from_positions(pos), materials=[scene.make_material()], textures=[], nodes=nodes, list=lst)
    n = 3_000_000  # a 12 MB deferred-ray list for the allocating path: above the allocator's granularity
    rng = np.random.RandomState(8)
    o = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], 1).astype(np.float32)
    ts = (2.0 ** rng.uniform(0, 59, n)).astype(np.float32)
    d = ((np.stack([ts, rng.uniform(-0.3, 0.3, n) * ts, rng.uniform(-0.3, 0.3, n) * ts], 1).astype(np.float32) - o) * np.float32(1.5)).astype(np.float32)
    ctx = capi.Context(0)  # fresh: its grow-only buffers are empty
    s = capi.Scene(ctx, sd)
    d_o, d_d = ctx.to_device(o), ctx.to_device(d)
    bufs = [ctx.malloc(n * 4) for _ in range(4)]
    try:
        ctx.sync()
        free0 = ctx.mem_info()[0]
        for occ in (False, True):
            s.raycast_dev_noalloc(d_o, d_d, n, occ, *bufs)
            s.raycast_dev_noalloc(d_o, d_d, n // 7, occ, *bufs)
        ctx.sync()
        free1 = ctx.mem_info()[0]
        assert free1 == free0, "ptgpu_raycast_batch_dev_noalloc changed free device memory by %d bytes" % (free0 - free1)
        got = ctx.d2h(bufs[0], n // 7, np.int32)
        s.raycast_dev(d_o, d_d, n, True, *bufs)  # the allocating entry point: deferred-ray list + overflow stacks appear
        ctx.sync()
        free2 = ctx.mem_info()[0]
        assert free2 < free1, "the probe cannot see the allocating entry point allocate"
        assert np.array_equal(ctx.d2h(bufs[0], n // 7, np.int32), got)
    finally:
        for b in bufs + [d_o, d_d]:
            ctx.free(b)
        s.close()
        ctx.close()
