"""GPU parity on the larger BASELINE configurations and on trees deeper than the LDS traversal stack.

configs[3] stand-in: scene.atrium_scene (~262k triangles, procedural: Crytek Sponza is not in the reference checkout) at small image
sizes here -- tests/test_config3_gpu.py runs it as written (3840x2160, 1024 spp); configs[4] stand-in: scene.tetra_cloud_scene at
1 M triangles here -- tests/test_configs_gpu.py::test_config4_ten_million_triangles_dof_exact runs the full 10 M.  Deep trees: a geometric chain of
triangle clusters whose median-split BVH is 49 levels deep, so every kernel takes its global-memory stack-overflow variant."""
import numpy as np
import pytest

from pathtracer_amd import capi, hostapi, scene

pytestmark = pytest.mark.gpu


def _exact(g, c, what):
    assert np.array_equal(g["id"], c["id"]), "%s: %d hit ids differ" % (what, int((g["id"] != c["id"]).sum()))
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32)), "%s: %s not bit-exact" % (what, k)


def _scene_dict(m, nodes, lst):
    return dict(pos=m["pos"], nrm=m["nrm"], uv=m["uv"], mat_id=m["mat_id"], woop=hostapi.woop_from_positions(m["pos"]),
                materials=m["materials"], textures=[], nodes=nodes, list=lst)


def _random_rays(n, seed, lo, hi, length):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= (length / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d.astype(np.float32)


@pytest.fixture(scope="module")
def atrium(oracle):
    m = scene.atrium_scene()
    nodes, lst, st = hostapi.bvh_build(m["pos"], True)
    onodes, olst, ost = oracle.bvh_build(m["pos"], True)
    assert nodes.tobytes() == onodes.tobytes() and np.array_equal(lst, olst), "parallel host builder != serial restatement at 262k triangles"
    return _scene_dict(m, nodes, lst), st


def test_atrium_rays_and_render_exact(gpu_ctx, atrium, oracle):
    sd, st = atrium
    assert 250_000 < sd["pos"].shape[0] < 275_000 and 16 < st["true_depth"] - 1 <= 24
    s = capi.Scene(gpu_ctx, sd)
    try:
        w, h = 320, 180
        cam, lights = scene.atrium_view(w, h)
        o, d = oracle.gen_primary(cam, w, h)
        c = oracle.raycast(sd, o, d, False, nthreads=8)
        assert (c["id"] >= 0).mean() > 0.9
        _exact(s.raycast(o, d, False), c, "atrium primary")
        so, sdir, valid = oracle.gen_shadow(sd, lights[0], o, d, c)
        gs, cs = s.raycast(so, sdir, True), oracle.raycast(sd, so, sdir, True, nthreads=8)
        assert np.array_equal(gs["id"], cs["id"])
        ro, rd = _random_rays(100000, 3, [-5.5, 0.2, -2.2], [5.5, 5.0, 2.2], 8.0)
        _exact(s.raycast(ro, rd, False), oracle.raycast(sd, ro, rd, False, nthreads=8), "atrium incoherent")
        # full path trace, two lights, Gaussian AA, roulette: accumulation image bit-exact
        w, h = 96, 64
        cam, lights = scene.atrium_view(w, h)
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=3, sobol_block=False, filter_width=1.5, pass_seed=2)
        g = s.render_pass(prm, cam, lights)
        cimg, _ = oracle.render_pass(sd, prm, cam, lights, nthreads=8)
        assert cimg[..., :3].max() > 0.01
        assert np.array_equal(g.view(np.uint32), cimg.view(np.uint32)), "atrium render differs"
        rl = s.ray_length()  # long rays (58 steps per probe ray): the pass above ran with the low service threshold
        assert 40.0 < rl["steps_per_ray"] < 80.0 and rl["service_threshold"] == 12, rl
    finally:
        s.close()


def test_tetra_cloud_1m_dof_exact(gpu_ctx, oracle):
    m = scene.tetra_cloud_scene(1_000_000)
    nodes, lst, st = hostapi.bvh_build(m["pos"], True)
    sd = _scene_dict(m, nodes, lst)
    s = capi.Scene(gpu_ctx, sd)
    try:
        w, h = 256, 144
        cam, lights = scene.tetra_cloud_view(w, h)
        o, d = oracle.gen_primary(cam, w, h)
        c = oracle.raycast(sd, o, d, False, nthreads=8)
        assert (c["id"] >= 0).mean() > 0.5
        _exact(s.raycast(o, d, False), c, "tetra primary")
        ro, rd = _random_rays(50000, 5, [-0.6, -0.6, -0.6], [0.6, 0.6, 0.6], 1.5)
        _exact(s.raycast(ro, rd, True), oracle.raycast(sd, ro, rd, True, nthreads=8), "tetra incoherent any-hit")
        w, h = 64, 64  # depth of field with a Shirley-Chiu aperture (configs[4])
        cam, lights = scene.tetra_cloud_view(w, h)
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True, depth_of_field=True, aperture=0.01,
                                focal_distance=1.2, pass_seed=6)
        g = s.render_pass(prm, cam, lights)
        cimg, _ = oracle.render_pass(sd, prm, cam, lights, nthreads=8)
        assert np.array_equal(g.view(np.uint32), cimg.view(np.uint32)), "tetra DoF render differs"
    finally:
        s.close()


def _chain_scene(n=60, ratio=2.0, reps=3, seed=0):
    rng = np.random.RandomState(seed)
    tris = []
    for i in range(n):
        sc = ratio ** i
        for _ in range(reps):
            c = np.array([sc, rng.uniform(-0.2, 0.2) * sc, rng.uniform(-0.2, 0.2) * sc])
            tris.append(c + rng.uniform(-0.15, 0.15, (3, 3)) * sc)
    return np.asarray(tris, np.float32)


@pytest.mark.parametrize("sah", [False, True])
def test_deep_tree_uses_stack_overflow_exact(gpu_ctx, oracle, sah):
    """Tree depth 49 (median split) / 27 (SAH) > 25: raycast and integrator run their stack-overflow variants."""
    pos = _chain_scene()
    nodes, lst, st = oracle.bvh_build(pos, sah)
    assert st["true_depth"] > 26
    m = dict(pos=pos, nrm=scene.flat_normals(pos), uv=np.zeros((pos.shape[0], 3, 2), np.float32), mat_id=np.zeros(pos.shape[0], np.int32),
             materials=[scene.make_material(specular=(0.1, 0.1, 0.1), glossiness=20.0)])
    sd = _scene_dict(m, nodes, lst)
    s = capi.Scene(gpu_ctx, sd)
    try:
        assert s.stats()["depth"] == st["true_depth"]
        rng = np.random.RandomState(4)
        n = 60000
        # rays that run up the chain from the small end (deep far-child stacks) and random ones
        o = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], 1).astype(np.float32)
        tgt_scale = (2.0 ** rng.uniform(0, 59, n)).astype(np.float32)
        tgt = np.stack([tgt_scale, rng.uniform(-0.3, 0.3, n) * tgt_scale, rng.uniform(-0.3, 0.3, n) * tgt_scale], 1).astype(np.float32)
        d = ((tgt - o) * np.float32(1.5)).astype(np.float32)
        for occ in (False, True):
            c = oracle.raycast(sd, o, d, occ, nthreads=8)
            assert (c["id"] >= 0).mean() > 0.2
            g = s.raycast(o, d, occ)
            if occ:
                assert np.array_equal(g["id"], c["id"])
            else:
                _exact(g, c, "chain closest-hit")
        w, h = 64, 32
        cam = scene.make_camera(eye=(-2.0, 0.0, 0.0), forward=(1.0, 0.0, 0.0), up=(0, 1, 0), fov_deg=40.0, near=0.5, far=4e6, width=w, height=h)
        lights = [scene.make_light(position=(30.0, 25.0, 0.0), col0=(1, 0, 0), col1=(0, 0, -1), col2=(0, 1, 0), size=(10.0, 10.0), E=(5000, 5000, 5000))]
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=4, sobol_block=True, pass_seed=1)
        g = s.render_pass(prm, cam, lights)
        cimg, ctr = oracle.render_pass(sd, prm, cam, lights, nthreads=8)
        assert ctr["rays"] > w * h * 4 * 1.3 and cimg[..., :3].max() > 0
        assert np.array_equal(g.view(np.uint32), cimg.view(np.uint32)), "deep-tree render differs"
        # the one-launch frame on the same tree: its 16 LDS stack entries overflow into the per-workgroup global region
        # (depth - 17 entries per lane); primary and shadow hits against the oracle's
        fw, fh = 200, 120
        n = fw * fh
        fo, fd = oracle.gen_primary(cam, fw, fh)
        c = oracle.raycast(sd, fo, fd, False, nthreads=8)
        so, sdir, valid = oracle.gen_shadow(sd, lights[0], fo, fd, c)
        cs = oracle.raycast(sd, so, sdir, True, nthreads=8)
        bufs = [gpu_ctx.malloc(n * 4) for _ in range(5)]
        try:
            s.trace_frame_dev(cam, lights[0], fw, fh, *bufs)
            got_id, got_t = gpu_ctx.d2h(bufs[0], n, np.int32), gpu_ctx.d2h(bufs[1], n, np.uint32)
            got_sid = gpu_ctx.d2h(bufs[4], n, np.int32)
        finally:
            for b in bufs:
                gpu_ctx.free(b)
        assert (c["id"] >= 0).mean() > 0.05
        assert np.array_equal(got_id, c["id"]) and np.array_equal(got_t, c["t"].view(np.uint32))
        assert np.array_equal(got_sid, np.where(valid == 1, cs["id"], -1))
    finally:
        s.close()
