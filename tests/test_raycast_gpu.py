"""GPU parity of RayTracer::raycast (reference src/RayTracer.cpp:174-290) through the C ABI: hit id, t, u, v must be
BIT-EXACT against the CPU oracle for closest-hit and any-hit rays (BASELINE config 2)."""
import numpy as np
import pytest

from pathtracer_amd import capi, scene

pytestmark = pytest.mark.gpu


def _assert_same(g, c, what):
    assert np.array_equal(g["id"], c["id"]), "%s: %d hit ids differ" % (what, int((g["id"] != c["id"]).sum()))
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32)), "%s: %s not bit-exact" % (what, k)


def _random_rays(n, seed, lo, hi, length=3.0):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= (length / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d.astype(np.float32)


@pytest.fixture(scope="module")
def head_gpu(gpu_ctx, head):
    s = capi.Scene(gpu_ctx, head["sd"])
    yield s
    s.close()


@pytest.fixture(scope="module")
def soup_gpu(gpu_ctx, soup):
    s = capi.Scene(gpu_ctx, soup["sd"])
    yield s
    s.close()


def test_scene_stats(head_gpu, head):
    st = head_gpu.stats()
    assert st["depth"] == head["stats"]["true_depth"] == 22
    n_inner = int((head["nodes"]["leaf"] == 0).sum())
    assert st["inner_bytes"] == n_inner * 64 and st["tri_bytes"] == len(head["list"]) * 64


@pytest.mark.parametrize("occlusion", [False, True])
def test_head_primary_small(head_gpu, head, oracle, occlusion):
    cam = scene.make_camera(width=480, height=270)
    o, d = oracle.gen_primary(cam, 480, 270)
    _assert_same(head_gpu.raycast(o, d, occlusion), oracle.raycast(head["sd"], o, d, occlusion), "head primary")


@pytest.mark.parametrize("occlusion", [False, True])
def test_head_incoherent(head_gpu, head, oracle, occlusion):
    o, d = _random_rays(200000, 11, [-1.4, -0.45, -0.45], [1.4, 0.95, 0.45])
    g, c = head_gpu.raycast(o, d, occlusion), oracle.raycast(head["sd"], o, d, occlusion, nthreads=8)
    assert (c["id"] >= 0).mean() > 0.5
    _assert_same(g, c, "head incoherent")


@pytest.mark.parametrize("order", ["1", "3", "9"])
def test_interior_record_order_does_not_change_results(gpu_ctx, head, oracle, order, monkeypatch):
    """PTGPU_NODE_ORDER (csrc/ptgpu.hip interior_slots): breadth-first, treelet and sibling-pair placement of the interior records are
    layouts only -- hits stay bit-exact against the oracle and the traversal counters stay those of the pre-order scene."""
    monkeypatch.setenv("PTGPU_NODE_ORDER", order)
    s = capi.Scene(gpu_ctx, head["sd"])
    monkeypatch.delenv("PTGPU_NODE_ORDER")
    try:
        o, d = _random_rays(60000, 23, [-1.4, -0.45, -0.45], [1.4, 0.95, 0.45])
        for occlusion in (False, True):
            _assert_same(s.raycast(o, d, occlusion), oracle.raycast(head["sd"], o, d, occlusion, nthreads=8), "order " + order)
    finally:
        s.close()


@pytest.mark.parametrize("occlusion", [False, True])
def test_soup_rays(soup_gpu, soup, oracle, occlusion):
    o, d = _random_rays(50000, 5, [-1.5, -1.5, -1.0], [1.5, 1.5, 1.0], length=4.0)
    g, c = soup_gpu.raycast(o, d, occlusion), oracle.raycast(soup["sd"], o, d, occlusion)
    assert 0.05 < (c["id"] >= 0).mean() < 0.95
    _assert_same(g, c, "soup")


def test_degenerate_rays(soup_gpu, soup, oracle):
    """Axis-parallel directions (1/dir = +-inf, 0*inf = NaN in the slab test), zero and NaN directions, -0 components."""
    o, d = _random_rays(4096, 9, [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0])
    d[0::8, 0] = 0.0
    d[1::8, 1] = -0.0
    d[2::8, 2] = 0.0
    d[3::8, :2] = 0.0
    d[4::64] = 0.0
    d[5::64, 0] = np.nan
    d[6::64, 1] = np.inf
    o[7::64] = soup["mesh"]["pos"][:64, 0][: len(o[7::64])]  # origins exactly on vertices
    for occ in (False, True):
        _assert_same(soup_gpu.raycast(o, d, occ), oracle.raycast(soup["sd"], o, d, occ), "degenerate")


def test_counters_match_cpu_traversal(gpu_ctx, head_gpu, head, oracle):
    """Node visits / triangle tests counted on the GPU equal the reference traversal's (same nodes, same order)."""
    cam = scene.make_camera(width=256, height=256)
    o, d = oracle.gen_primary(cam, 256, 256)
    for occ in (False, True):
        c = oracle.raycast(head["sd"], o, d, occ)
        gpu_ctx.enable_counters(True)
        gpu_ctx.reset_counters()
        g = head_gpu.raycast(o, d, occ)
        got = gpu_ctx.counters()
        gpu_ctx.enable_counters(False)
        _assert_same(g, c, "counting variant")
        assert got == c["counters"], (got, c["counters"])


def test_empty_batch_and_outputs_optional(head_gpu):
    out = head_gpu.raycast(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert out["id"].shape == (0,)
    o, d = _random_rays(1000, 2, [-1, -0.4, -0.4], [1, 0.9, 0.4])
    only_id = head_gpu.raycast(o, d, want=("id",))
    assert set(only_id) == {"id"} and np.array_equal(only_id["id"], head_gpu.raycast(o, d)["id"])


def test_tiny_and_empty_scenes(gpu_ctx, oracle):
    """Root-is-leaf scenes (1..3 triangles, reference src/Bvh.cpp:110) and the empty scene (all rays miss)."""
    o, d = _random_rays(2048, 4, [-1, -1, 1.0], [1, 1, 2.0])
    d[:, 2] = -np.abs(d[:, 2]) - 1.0
    for ntri in (1, 2, 3):
        pos = scene.grid_scene(1, 1, seed=ntri)[:ntri]
        nodes, lst, st = oracle.bvh_build(pos, True)
        assert len(nodes) == 1 and nodes[0]["leaf"] == 1
        sd = dict(pos=pos, nrm=scene.flat_normals(pos), uv=np.zeros((ntri, 3, 2), np.float32), mat_id=np.zeros(ntri, np.int32),
                  woop=oracle.woop_from_positions(pos), materials=[scene.make_material()], textures=[], nodes=nodes, list=lst)
        s = capi.Scene(gpu_ctx, sd)
        for occ in (False, True):
            _assert_same(s.raycast(o, d, occ), oracle.raycast(sd, o, d, occ), "tiny scene %d" % ntri)
        s.close()
    empty = dict(pos=np.zeros((0, 3, 3), np.float32), nrm=np.zeros((0, 3, 3), np.float32), uv=np.zeros((0, 3, 2), np.float32),
                 mat_id=np.zeros(0, np.int32), woop=np.zeros((0, 12), np.float32), materials=[], textures=[],
                 nodes=np.zeros(0, capi.FLATNODE_DTYPE), list=np.zeros(0, np.int32))
    s = capi.Scene(gpu_ctx, empty)
    out = s.raycast(o, d)
    assert (out["id"] == -1).all() and (out["t"] == np.float32(3.402823466e38)).all()
    s.close()


def test_malformed_scene_is_rejected(gpu_ctx, soup):
    """A corrupt tree must fail at scene creation with a message, never reach a kernel."""
    bad = dict(soup["sd"])
    nodes = soup["nodes"].copy()
    inner = np.flatnonzero(nodes["leaf"] == 0)
    nodes["rightChild"][inner[1]] = len(nodes) + 5
    bad["nodes"] = nodes
    with pytest.raises(capi.PtgpuError, match="rightChild"):
        capi.Scene(gpu_ctx, bad)
    bad2 = dict(soup["sd"])
    lst = soup["list"].copy()
    lst[3] = 10 ** 6
    bad2["list"] = lst
    with pytest.raises(capi.PtgpuError, match="out of range"):
        capi.Scene(gpu_ctx, bad2)


def test_alpha_cutout(gpu_ctx, soup, oracle):
    """Alpha-texture cutout in the leaf loop (reference src/RayTracer.cpp:204-225): green < 0.5 skips the triangle."""
    rng = np.random.RandomState(21)
    tex = np.zeros((64, 64, 4), np.uint8)
    tex[..., 1] = (rng.rand(64, 64) > 0.5) * 255
    sd = dict(soup["sd"])
    sd["materials"] = [scene.make_material(tex_alpha=0)]
    sd["textures"] = [tex]
    s = capi.Scene(gpu_ctx, sd)
    o, d = _random_rays(50000, 6, [-1.5, -1.5, -1.0], [1.5, 1.5, 1.0], length=4.0)
    for occ in (False, True):
        with_alpha = oracle.raycast(sd, o, d, occ, use_alpha=True)
        _assert_same(s.raycast(o, d, occ, use_alpha=True), with_alpha, "alpha on")
        without = oracle.raycast(sd, o, d, occ, use_alpha=False)
        _assert_same(s.raycast(o, d, occ, use_alpha=False), without, "alpha off")
    assert not np.array_equal(with_alpha["id"], without["id"])
    s.close()


def test_config2_full_size(gpu_ctx, head_gpu, head, oracle):
    """BASELINE config 2 at full size: 1920x1080 primary rays + one shadow ray per hit, generated on the device,
    bit-exact ids/t/u/v against the oracle (rays, hits and shadow rays all compared)."""
    W, H = 1920, 1080
    n = W * H
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    f3, f1 = n * 12, n * 4
    d_o, d_d, d_so, d_sd = (gpu_ctx.malloc(f3) for _ in range(4))
    d_id, d_t, d_u, d_v, d_valid, d_sid = (gpu_ctx.malloc(f1) for _ in range(6))
    gpu_ctx.gen_primary_dev(cam, W, H, d_o, d_d)
    head_gpu.raycast_dev(d_o, d_d, n, False, d_id, d_t, d_u, d_v)
    head_gpu.gen_shadow_dev(light, n, d_o, d_d, d_id, d_t, d_u, d_v, d_so, d_sd, d_valid)
    head_gpu.raycast_dev(d_so, d_sd, n, True, d_sid, None, None, None)
    gpu_ctx.sync()
    o, d = oracle.gen_primary(cam, W, H)
    assert np.array_equal(gpu_ctx.d2h(d_o, (n, 3), np.float32).view(np.uint32), o.view(np.uint32))
    assert np.array_equal(gpu_ctx.d2h(d_d, (n, 3), np.float32).view(np.uint32), d.view(np.uint32))
    c = oracle.raycast(head["sd"], o, d, False, nthreads=8)
    g = dict(id=gpu_ctx.d2h(d_id, n, np.int32), t=gpu_ctx.d2h(d_t, n, np.float32), u=gpu_ctx.d2h(d_u, n, np.float32),
             v=gpu_ctx.d2h(d_v, n, np.float32))
    _assert_same(g, c, "C2 primary")
    so, sdir, valid = oracle.gen_shadow(head["sd"], light, o, d, c)
    assert np.array_equal(gpu_ctx.d2h(d_valid, n, np.int32), valid)
    assert np.array_equal(gpu_ctx.d2h(d_so, (n, 3), np.float32).view(np.uint32), so.view(np.uint32))
    assert np.array_equal(gpu_ctx.d2h(d_sd, (n, 3), np.float32).view(np.uint32), sdir.view(np.uint32))
    cs = oracle.raycast(head["sd"], so, sdir, True, nthreads=8)
    gs = gpu_ctx.d2h(d_sid, n, np.int32)
    m = valid == 1
    assert np.array_equal(gs[m], cs["id"][m])
    for p in (d_o, d_d, d_so, d_sd, d_id, d_t, d_u, d_v, d_valid, d_sid):
        gpu_ctx.free(p)


def test_intersect_triangle1_matches_reference_goldens(gpu_ctx):
    """rtIntersect.inl intersect_triangle1 on the device against vectors produced by the reference's own source
    (tests/golden/rtintersect_golden.json, oracle/_ref): hit flags equal, t/u/v bit-exact."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rtintersect_golden.json")))["cases"]
    arr = lambda k: np.array([c[k] for c in g], np.float32)  # noqa: E731
    hit, t, u, v = gpu_ctx.intersect_triangle1(arr("o"), arr("d"), arr("v0"), arr("v1"), arr("v2"))
    want_hit = np.array([c["hit"] for c in g], np.int32)
    assert np.array_equal(hit, want_hit) and want_hit.sum() > 20
    got = np.stack([t, u, v], 1).view(np.uint32)
    for i, c in enumerate(g):
        if c["hit"]:
            assert got[i].tolist() == c["tuv_bits"], i


@pytest.mark.parametrize("size", [(1920, 1080), (203, 77)])
def test_trace_frame_equals_the_four_launch_sequence(gpu_ctx, head_gpu, size, monkeypatch):
    """ptgpu_trace_frame_dev (primary closest-hit + shadow any-hit with the rays in registers) against
    gen_primary -> raycast -> gen_shadow -> raycast (itself compared with the oracle in test_config2_full_size): every output word
    equal, also on an image that is not a whole number of 8x8 wave tiles; one ray counted per pixel plus one per hit."""
    W, H = size
    n = W * H
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    d_o, d_d, d_so, d_sd = (gpu_ctx.malloc(n * 12) for _ in range(4))
    d_id, d_t, d_u, d_v, d_valid, d_sid = (gpu_ctx.malloc(n * 4) for _ in range(6))
    gpu_ctx.reset_counters()
    gpu_ctx.gen_primary_dev(cam, W, H, d_o, d_d)
    head_gpu.raycast_dev(d_o, d_d, n, False, d_id, d_t, d_u, d_v)
    head_gpu.gen_shadow_dev(light, n, d_o, d_d, d_id, d_t, d_u, d_v, d_so, d_sd, d_valid)
    head_gpu.raycast_dev(d_so, d_sd, n, True, d_sid, None, None, None)
    want = dict(id=gpu_ctx.d2h(d_id, n, np.int32), t=gpu_ctx.d2h(d_t, n, np.uint32), u=gpu_ctx.d2h(d_u, n, np.uint32),
                v=gpu_ctx.d2h(d_v, n, np.uint32), sid=gpu_ctx.d2h(d_sid, n, np.int32))
    hits = int((want["id"] >= 0).sum())
    assert gpu_ctx.counters()["rays"] == 2 * n and hits > n // 10  # the separate form also "traces" the empty rays of missed pixels
    f_id, f_t, f_u, f_v, f_sid = (gpu_ctx.malloc(n * 4) for _ in range(5))
    for tile_w, nlds in (("8", "16"), ("8", "24"), ("16", "24")):  # the default (16 LDS entries + overflow region) and the all-LDS forms
        monkeypatch.setenv("PTGPU_FRAME_TILE_W", tile_w)
        monkeypatch.setenv("PTGPU_FRAME_NLDS", nlds)
        for p in (f_id, f_t, f_u, f_v, f_sid):
            capi._check(gpu_ctx.lib.ptgpu_memset(gpu_ctx.handle, p, 0xEE, n * 4))
        gpu_ctx.reset_counters()
        head_gpu.trace_frame_dev(cam, light, W, H, f_id, f_t, f_u, f_v, f_sid)
        got = dict(id=gpu_ctx.d2h(f_id, n, np.int32), t=gpu_ctx.d2h(f_t, n, np.uint32), u=gpu_ctx.d2h(f_u, n, np.uint32),
                   v=gpu_ctx.d2h(f_v, n, np.uint32), sid=gpu_ctx.d2h(f_sid, n, np.int32))
        for k in want:
            assert np.array_equal(got[k], want[k]), "tile_w=%s: %s differs in %d places" % (tile_w, k, int((got[k] != want[k]).sum()))
        assert gpu_ctx.counters()["rays"] == n + hits  # one raycast() per pixel plus one per hit
    for p in (d_o, d_d, d_so, d_sd, d_id, d_t, d_u, d_v, d_valid, d_sid, f_id, f_t, f_u, f_v, f_sid):
        gpu_ctx.free(p)


def test_trace_frame_equals_the_oracle_at_1080p(gpu_ctx, head_gpu, head, oracle):
    """The kernel the headline number times, compared DIRECTLY with the oracle on the configs[1] frame: primary hit records
    (RayTracer::raycast closest-hit, reference src/RayTracer.cpp:174-290) and the shadow ray's occluder (any-hit, :235-238) for the
    shadow rays oracle.gen_shadow builds (src/PathTraceRenderer.cpp:338-344) -- 2,073,600 pixels, every word."""
    W, H = 1920, 1080
    n = W * H
    cam, light = scene.make_camera(width=W, height=H), scene.make_light()
    f_id, f_t, f_u, f_v, f_sid = (gpu_ctx.malloc(n * 4) for _ in range(5))
    for p in (f_id, f_t, f_u, f_v, f_sid):
        capi._check(gpu_ctx.lib.ptgpu_memset(gpu_ctx.handle, p, 0xEE, n * 4))
    head_gpu.trace_frame_dev(cam, light, W, H, f_id, f_t, f_u, f_v, f_sid)
    g = dict(id=gpu_ctx.d2h(f_id, n, np.int32), t=gpu_ctx.d2h(f_t, n, np.float32), u=gpu_ctx.d2h(f_u, n, np.float32),
             v=gpu_ctx.d2h(f_v, n, np.float32))
    gs = gpu_ctx.d2h(f_sid, n, np.int32)
    for p in (f_id, f_t, f_u, f_v, f_sid):
        gpu_ctx.free(p)
    o, d = oracle.gen_primary(cam, W, H)
    c = oracle.raycast(head["sd"], o, d, False, nthreads=8)
    _assert_same(g, c, "trace_frame primary")
    so, sdir, valid = oracle.gen_shadow(head["sd"], light, o, d, c)
    cs = oracle.raycast(head["sd"], so, sdir, True, nthreads=8)
    m = valid == 1
    assert m.sum() > n // 10 and np.array_equal(gs[m], cs["id"][m]) and (gs[~m] == -1).all()
    assert 0.05 < (cs["id"][m] >= 0).mean() < 0.95  # both outcomes of the shadow ray occur


def test_trace_frame_rejects_a_tile_width_it_was_not_built_for(gpu_ctx, head_gpu, monkeypatch):
    cam, light = scene.make_camera(width=64, height=64), scene.make_light()
    bufs = [gpu_ctx.malloc(64 * 64 * 4) for _ in range(5)]
    try:
        for bad in ("0", "5", "128"):
            monkeypatch.setenv("PTGPU_FRAME_TILE_W", bad)
            with pytest.raises(capi.PtgpuError, match="PTGPU_FRAME_TILE_W"):
                head_gpu.trace_frame_dev(cam, light, 64, 64, *bufs)
        monkeypatch.setenv("PTGPU_FRAME_TILE_W", "16")
        head_gpu.trace_frame_dev(cam, light, 64, 64, *bufs)
        gpu_ctx.sync()
    finally:
        for b in bufs:
            gpu_ctx.free(b)
