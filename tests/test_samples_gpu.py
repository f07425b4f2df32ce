"""The reference's static per-sample functions through the C ABI, checked value by value against the oracle (they are otherwise only
visible through accumulated pixels): tracePath / traceWhitted (reference src/PathTraceRenderer.cpp:236-551, :553-660; declared static at
src/PathTraceRenderer.hpp:97-98), getTextureParameters (:75-181) and getLightToSample (:183-210)."""
import numpy as np
import pytest

from pathtracer_amd import capi, scene

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(g, c):
    return (_bits(g) == _bits(c)) | (np.isnan(g) & np.isnan(c))


def _samples(n, seed, oracle):
    rng = np.random.RandomState(seed)
    xy = rng.uniform(-1.0, 1.0, (n, 2)).astype(np.float32)
    base = rng.randint(0, 1 << 22, n).astype(np.int32)
    keys = rng.randint(0, 1 << 31, (n, 4))
    state = np.stack([np.array([oracle.rng_key(*k) for k in keys], np.uint32), rng.randint(0, 7, n).astype(np.uint32)], 1)
    return xy, base, state


@pytest.fixture(scope="module")
def head_textured(head):
    tex, index, _ = scene.head_textures(head["mesh"], size=256)  # reference files at 1600^2, stand-ins small: same code paths
    sd = dict(head["sd"])
    sd["materials"], sd["textures"] = scene.mesh_materials(head["mesh"], textured=True, texture_index=index), tex
    return sd


def test_trace_path_100k_samples_exact(gpu_ctx, head_textured, oracle):
    """100,000 random (x, y, samplerBase, stream state) tuples on the textured head: per-sample radiance and the number of random draws
    consumed equal the oracle's trace_path, Russian roulette and fixed bounces, normal maps on."""
    s = capi.Scene(gpu_ctx, head_textured)
    try:
        cam, lights = scene.make_camera(width=640, height=360), [scene.make_light(), scene.make_light(position=(0.4, 0.8, 0.5), size=(0.1, 0.2), E=(30, 40, 50))]
        for n, bounces, seed in ((100_000, -1, 1), (20_000, 2, 2), (20_000, 0, 3)):
            prm = scene.make_params(width=640, height=360, bounces=bounces, bilinear=True, normal_mapped=True, termination_prob=0.2)
            xy, base, state = _samples(n, seed, oracle)
            g_rgb, g_ctr = s.trace_samples(prm, cam, lights, xy, base, state)
            c_rgb, c_ctr = oracle.trace_samples(head_textured, prm, cam, lights, xy, base, state, nthreads=8)
            assert np.nanmax(c_rgb) > 0.01 and (c_rgb.sum(1) > 0).mean() > 0.3
            bad = ~_same(g_rgb, c_rgb)
            assert not bad.any(), "bounces=%d: %d of %d radiance words differ" % (bounces, int(bad.sum()), bad.size)
            assert np.array_equal(g_ctr, c_ctr), "bounces=%d: random draws consumed differ" % bounces
            assert (c_ctr >= state[:, 1]).all() and (c_ctr > state[:, 1]).any()
    finally:
        s.close()


def test_trace_whitted_and_dof_samples_exact(gpu_ctx, head_textured, oracle):
    s = capi.Scene(gpu_ctx, head_textured)
    try:
        cam, lights = scene.make_camera(width=320, height=200), [scene.make_light()]
        xy, base, state = _samples(20_000, 5, oracle)
        for whitted in (True, False):
            prm = scene.make_params(width=320, height=200, bounces=-1, whitted=whitted, num_light_rays=4, depth_of_field=True, aperture=0.01,
                                    focal_distance=0.45, normal_mapped=True, intensity=1.2, attenuation=0.8)
            g_rgb, g_ctr = s.trace_samples(prm, cam, lights, xy, base, state)
            c_rgb, c_ctr = oracle.trace_samples(head_textured, prm, cam, lights, xy, base, state, nthreads=8)
            assert _same(g_rgb, c_rgb).all() and np.array_equal(g_ctr, c_ctr), "whitted=%s" % whitted
        # empty batch and argument checks
        rgb, ctr = s.trace_samples(prm, cam, lights, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), np.zeros((0, 2), np.uint32))
        assert rgb.shape == (0, 3) and ctr.shape == (0,)
    finally:
        s.close()


def test_get_texture_parameters_exact(gpu_ctx, head_textured, oracle):
    """getTextureParameters on 50,000 random (triangle, u, v): diffuse (bilinear and nearest, pow 2.2), tangent-space normal, specular."""
    s = capi.Scene(gpu_ctx, head_textured)
    try:
        rng = np.random.RandomState(8)
        n = 50_000
        ntri = head_textured["mat_id"].shape[0]
        tri = np.concatenate([rng.randint(0, ntri, n - 2000), rng.randint(ntri - 8, ntri, 2000)]).astype(np.int32)  # wall + floor triangles too
        u = rng.uniform(0, 1, n).astype(np.float32)
        v = (rng.uniform(0, 1, n) * (1 - u)).astype(np.float32)
        for bilinear, nm in ((True, True), (False, True), (True, False)):
            prm = scene.make_params(bilinear=bilinear, normal_mapped=nm)
            g, c = s.texture_parameters(prm, tri, u, v), oracle.texture_parameters(head_textured, prm, tri, u, v)
            for name, a, b in zip(("diffuse", "normal", "specular"), g, c):
                assert _same(a, b).all(), "%s bilinear=%s normal_mapped=%s" % (name, bilinear, nm)
        assert np.nanstd(c[0]) > 0.01
    finally:
        s.close()


def test_get_light_to_sample_exact(gpu_ctx, oracle):
    rng = np.random.RandomState(12)
    lights = [scene.make_light(position=tuple(rng.uniform(-1, 1, 3)), size=(0.1 + 0.1 * i, 0.2), E=(10.0 * (i + 1),) * 3, intensity=1.0 + i) for i in range(5)]
    n = 30_000
    pos = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    state = np.stack([rng.randint(0, 1 << 31, n).astype(np.uint32), rng.randint(0, 5, n).astype(np.uint32)], 1)
    for k in (1, 2, 5):
        gi, gc = gpu_ctx.light_to_sample(lights[:k], pos, state)
        ci, cc = oracle.light_to_sample(lights[:k], pos, state)
        assert np.array_equal(gi, ci) and np.array_equal(gc, cc)
        assert (ci == 0).all() if k == 1 else len(np.unique(ci)) == k
