"""GPU parity of the path-trace pass (reference src/PathTraceRenderer.cpp:236-551 tracePath, :663-790 pathTraceBlock,
:855-871 updatePicture) through the C ABI.

The oracle is run with its portable transcendentals (libm=False): the HIP kernels evaluate the same polynomial kernels
with the same un-fused fp32 arithmetic, and pixels are accumulated in the reference's sample order, so the accumulation
image must be BIT-EXACT, not merely close.  The oracle's libm variant (what the reference calls) is compared by RMSE:
tolerance 2e-3 per-pixel RMSE and 1e-3 relative mean luminance on the resolved image (north_star's image criterion)."""
import os

import numpy as np
import pytest

from pathtracer_amd import capi, scene

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_image_exact(g, c, what):
    diff = _bits(g) != _bits(c)
    assert not diff.any(), "%s: %d of %d accumulation words differ (max abs %.3g)" % (
        what, int(diff.sum()), diff.size, float(np.abs(g - c).max()))


@pytest.fixture(scope="module")
def head_gpu(gpu_ctx, head):
    s = capi.Scene(gpu_ctx, head["sd"])
    yield s
    s.close()


@pytest.fixture(scope="module")
def textured_soup(oracle):
    """Triangle soup with every material feature on: diffuse + normal + specular + alpha maps, two materials."""
    pos = scene.grid_scene(10, 10, seed=5)
    rng = np.random.RandomState(9)
    n = pos.shape[0]
    # a floor quad pair under the soup so that bounce rays find something
    floor = np.array([[[-3, -3, -0.6], [3, -3, -0.6], [3, 3, -0.6]], [[-3, -3, -0.6], [3, 3, -0.6], [-3, 3, -0.6]]], np.float32)
    pos = np.concatenate([pos, floor]).astype(np.float32)
    uv = (rng.rand(pos.shape[0], 3, 2) * 3.0 - 1.0).astype(np.float32)  # outside [0,1): exercises frac()
    mat_id = np.concatenate([rng.randint(0, 2, n), [1, 1]]).astype(np.int32)
    tex = [scene.value_noise_texture(1, 64), scene.value_noise_texture(2, 64, normal_map=True),
           scene.value_noise_texture(3, 32), scene.value_noise_texture(4, 48)]
    # sizes that are not whole device tiles (8 x 4 texels), not square
    tex[0], tex[2], tex[3] = tex[0][:61, :59].copy(), tex[2][:30, :27].copy(), tex[3][:45, :43].copy()
    tex[3][..., 1] = (tex[3][..., 0] > 150) * 255  # alpha map: green channel decides
    mats = [scene.make_material(diffuse=(0.7, 0.6, 0.5, 1), specular=(0.15, 0.15, 0.15), glossiness=50.0, tex_diffuse=0, tex_normal=1,
                                tex_specular=2, tex_alpha=3),
            scene.make_material(diffuse=(0.6, 0.7, 0.8, 1), specular=(0.3, 0.2, 0.1), glossiness=8.0)]
    nodes, lst, _ = oracle.bvh_build(pos, True)
    sd = dict(pos=pos, nrm=scene.flat_normals(pos), uv=uv, mat_id=mat_id, woop=oracle.woop_from_positions(pos), materials=mats,
              textures=tex, nodes=nodes, list=lst)
    return sd


@pytest.fixture(scope="module")
def soup_gpu(gpu_ctx, textured_soup):
    s = capi.Scene(gpu_ctx, textured_soup)
    yield s
    s.close()


def _soup_view(w, h):
    cam = scene.make_camera(eye=(0.2, -0.3, 2.2), forward=(-0.05, 0.1, -1.0), up=(0, 1, 0), fov_deg=60.0, near=0.01, far=8.0, width=w, height=h)
    lights = [scene.make_light(position=(0.3, 0.2, 1.5), col0=(1, 0, 0), col1=(0, 1, 0), col2=(0, 0, 1), size=(0.3, 0.2), E=(60, 50, 40)),
              scene.make_light(position=(-1.0, 0.5, 1.0), col0=(1, 0, 0), col1=(0, 1, 0), col2=(0, 0, 1), size=(0.1, 0.4), E=(20, 30, 40),
                               intensity=2.0)]
    return cam, lights


def test_head_sobol_block_rr_exact(gpu_ctx, head_gpu, head, oracle):
    """C1-style settings (Sobol-block AA, Russian roulette after 1 bounce) on an image whose edge blocks are clipped."""
    w, h = 112, 80
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=4, sobol_block=True, pass_seed=3)
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    gpu_ctx.reset_counters()
    g = head_gpu.render_pass(prm, cam, lights)
    rays = gpu_ctx.counters()["rays"]
    c, ctr = oracle.render_pass(head["sd"], prm, cam, lights, nthreads=8)
    assert c[..., 3].min() >= 0 and c[..., :3].max() > 0.01
    _assert_image_exact(g, c, "head sobol-block")
    assert rays == ctr["rays"], "ray counter %d != oracle %d" % (rays, ctr["rays"])


def test_scheduling_knobs_never_change_the_image(gpu_ctx, head_gpu, head, oracle, monkeypatch):
    """The integrator's service threshold (how many lanes wait for shading before the wave serves them) is chosen per scene from a
    ray-length probe, and the control section's entry count is a second knob: whatever their values, every sample is the same sample.
    The probe's rays go to counter words of their own."""
    w, h = 96, 64
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=3, sobol_block=True, pass_seed=6)
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    c, ctr = oracle.render_pass(head["sd"], prm, cam, lights, nthreads=8)
    fresh = capi.Scene(gpu_ctx, head["sd"])  # a scene that has not been probed yet
    try:
        gpu_ctx.reset_counters()
        g = fresh.render_pass(prm, cam, lights)  # runs the probe first
        assert gpu_ctx.counters()["rays"] == ctr["rays"], "the probe's rays leaked into the public counter"
        _assert_image_exact(g, c, "first pass on a fresh scene")
        rl = fresh.ray_length()
        assert 10.0 < rl["steps_per_ray"] < 25.0 and rl["service_threshold"] == 32  # head.obj: short rays (15 steps)
    finally:
        fresh.close()
    for service, ctlmin in ((1, 1), (12, 16), (64, 16), (24, 4), (200, 64)):
        monkeypatch.setenv("PTGPU_PT_SERVICE", str(service))
        monkeypatch.setenv("PTGPU_PT_CTLMIN", str(ctlmin))
        _assert_image_exact(head_gpu.render_pass(prm, cam, lights), c, "service %d, control minimum %d" % (service, ctlmin))


def test_head_two_passes_accumulate(head_gpu, head, oracle):
    w, h = 64, 64
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    g = c = None
    for seed in (0, 1):
        prm = scene.make_params(width=w, height=h, bounces=2, num_aa_rays=2, sobol_block=True, pass_seed=seed)
        g = head_gpu.render_pass(prm, cam, lights, accum=g)
        c, _ = oracle.render_pass(head["sd"], prm, cam, lights, accum=c, nthreads=8)
    _assert_image_exact(g, c, "two passes")
    assert abs(float(c[..., 3].mean()) - 2.0) < 1e-6  # every sample adds 1/NumAaRays to some pixel of its block


@pytest.mark.parametrize("sobol_block", [True, False])
def test_textured_soup_all_features_exact(soup_gpu, textured_soup, oracle, sobol_block):
    """bilinear + normal maps + specular map + alpha cutout + two lights (intensity-biased selection) + depth of field +
    fixed bounces followed by roulette; both AA modes."""
    w, h = 96, 72
    cam, lights = _soup_view(w, h)
    prm = scene.make_params(width=w, height=h, bounces=-2, num_aa_rays=3, sobol_block=sobol_block, bilinear=True, normal_mapped=True,
                            depth_of_field=True, aperture=0.02, focal_distance=2.0, filter_width=1.5, attenuation=0.8,
                            termination_prob=0.3, pass_seed=11)
    g = soup_gpu.render_pass(prm, cam, lights)
    c, _ = oracle.render_pass(textured_soup, prm, cam, lights, nthreads=8)
    assert c[..., :3].max() > 0.01
    _assert_image_exact(g, c, "textured soup sobol_block=%s" % sobol_block)


def test_nearest_filtering_no_alpha_exact(soup_gpu, textured_soup, oracle):
    w, h = 64, 48
    cam, lights = _soup_view(w, h)
    prm = scene.make_params(width=w, height=h, bounces=1, num_aa_rays=2, sobol_block=False, bilinear=False, normal_mapped=False,
                            use_alpha=False, filter_width=1.0, pass_seed=2)
    g = soup_gpu.render_pass(prm, cam, lights[:1])
    c, _ = oracle.render_pass(textured_soup, prm, cam, lights[:1], nthreads=8)
    _assert_image_exact(g, c, "nearest / no alpha")


@pytest.mark.parametrize("bounces", [0, 3, -4])
def test_bounce_settings_on_open_geometry_exact(soup_gpu, textured_soup, oracle, bounces):
    """Direct light only / fixed bounces / fixed bounces then roulette on a triangle soup: open geometry gives back-facing hits (the
    reference flips the normal only when the shadow ray is unoccluded, which decides where the bounce ray starts) and bounce rays
    that leave the scene inside the fixed-bounce loop (then the roulette section still runs from the last hit point)."""
    w, h = 80, 64
    cam, lights = _soup_view(w, h)
    prm = scene.make_params(width=w, height=h, bounces=bounces, num_aa_rays=2, sobol_block=True, bilinear=True, normal_mapped=True,
                            termination_prob=0.25, pass_seed=21 + bounces)
    g = soup_gpu.render_pass(prm, cam, lights)
    c, _ = oracle.render_pass(textured_soup, prm, cam, lights, nthreads=8)
    assert c[..., :3].max() > 0.01
    _assert_image_exact(g, c, "bounces=%d" % bounces)


def test_block_subsets_tile_the_image(head_gpu, head, oracle):
    """Image tiles sharded as rank r of N: block_begin = r, block_stride = N; the union equals the full pass bit for bit."""
    w, h = 96, 96
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True, pass_seed=5)
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    full = head_gpu.render_pass(prm, cam, lights)
    parts = np.zeros_like(full)
    for r in range(3):
        part = head_gpu.render_pass(prm, cam, lights, block_begin=r, block_stride=3)
        c, _ = oracle.render_pass(head["sd"], prm, cam, lights, nthreads=1, block_begin=r, block_end=r + 1)
        bx, by = (r % 3) * 32, (r // 3) * 32
        _assert_image_exact(part[by:by + 32, bx:bx + 32], c[by:by + 32, bx:bx + 32], "block %d" % r)
        parts += part
    _assert_image_exact(parts, full, "union of strided block subsets")


def test_resolve_exact_and_libm_rmse(gpu_ctx, head_gpu, head, oracle):
    w, h = 96, 64
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=8, sobol_block=True, pass_seed=1)
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    g = head_gpu.render_pass(prm, cam, lights)
    img = gpu_ctx.resolve(g)
    _assert_image_exact(img, oracle.resolve(g, libm=False), "resolve")
    # against what the reference calls (libm transcendentals): image-level tolerance
    c, _ = oracle.render_pass(head["sd"], prm, cam, lights, nthreads=8, libm=True)
    ref = oracle.resolve(c, libm=True)
    rmse = float(np.sqrt(np.mean((img[..., :3] - ref[..., :3]) ** 2)))
    lum = lambda a: float((a[..., :3] * np.array([0.299, 0.587, 0.114], np.float32)).sum(-1).mean())  # noqa: E731
    assert rmse < 2e-3, "per-pixel RMSE vs libm oracle %.3g" % rmse
    assert abs(lum(img) - lum(ref)) / lum(ref) < 1e-3


def test_rejects_bad_params(head_gpu):
    cam, lights = scene.make_camera(width=32, height=32), [scene.make_light()]
    with pytest.raises(capi.PtgpuError):
        head_gpu.render_pass(scene.make_params(width=32, height=32, num_aa_rays=0), cam, lights)
    with pytest.raises(capi.PtgpuError):
        head_gpu.render_pass(scene.make_params(width=32, height=32, bounces=-1, termination_prob=0.0), cam, lights)


@pytest.mark.parametrize("sobol_block", [True, False])
def test_whitted_mode_exact(soup_gpu, textured_soup, oracle, sobol_block):
    """traceWhitted (reference :553-660) with AreaLight::sampleDisk shadow rays, two lights, depth of field (no max(0.01,.) here)."""
    w, h = 64, 64
    cam, lights = _soup_view(w, h)
    prm = scene.make_params(width=w, height=h, num_aa_rays=2, sobol_block=sobol_block, whitted=True, num_light_rays=5, bilinear=True,
                            normal_mapped=True, depth_of_field=True, aperture=0.01, focal_distance=2.0, attenuation=0.7, intensity=1.3,
                            filter_width=1.2, pass_seed=8)
    g = soup_gpu.render_pass(prm, cam, lights)
    c, ctr = oracle.render_pass(textured_soup, prm, cam, lights, nthreads=8)
    assert c[..., :3].max() > 0.01 and ctr["rays"] > w * h * 2
    _assert_image_exact(g, c, "whitted sobol_block=%s" % sobol_block)


@pytest.mark.parametrize("spp", [8, 19])
def test_gaussian_many_samples_per_pixel_exact(soup_gpu, textured_soup, oracle, spp):
    """Gaussian AA with 8 and more samples per pixel: k_accum_gauss sums them eight loads at a time plus a remainder, still in sample
    order, so the normalised, saturated pixel is the sequential loop's (reference :743-790)."""
    w, h = 45, 37
    cam, lights = _soup_view(w, h)
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=spp, sobol_block=False, bilinear=True, normal_mapped=True,
                            filter_width=1.5, termination_prob=0.3, pass_seed=spp)
    g = soup_gpu.render_pass(prm, cam, lights)
    c, _ = oracle.render_pass(textured_soup, prm, cam, lights, nthreads=8)
    _assert_image_exact(g, c, "gaussian spp=%d" % spp)


# PT_TEST_SEEDS="200-260" widens the sweep for a soak run (python -m pytest tests/test_render_gpu.py -k randomised)
def _seeds():
    spec = os.environ.get("PT_TEST_SEEDS", "")
    if not spec:
        return [101, 102, 103, 104, 105, 106, 107, 108]
    lo, _, hi = spec.partition("-")
    return list(range(int(lo), int(hi or lo) + 1))


@pytest.mark.parametrize("seed", _seeds())
def test_randomised_settings_exact(soup_gpu, textured_soup, oracle, seed):
    """Random draws over every switch the pass has (AA mode, fixed / roulette bounces, filtering, normal maps, alpha, depth of field,
    Whitted, 1-3 lights, clipped image sizes): whatever the combination, the accumulation image is the oracle's, bit for bit."""
    rng = np.random.RandomState(seed)
    w, h = int(rng.randint(33, 90)), int(rng.randint(33, 70))
    cam, lights = _soup_view(w, h)
    lights = lights + [scene.make_light(position=(0.0, -0.8, 1.2), col0=(1, 0, 0), col1=(0, 1, 0), col2=(0, 0, 1), size=(0.2, 0.2), E=(30, 30, 50))]
    lights = lights[: int(rng.randint(1, 4))]
    whitted = bool(rng.rand() < 0.25)
    prm = scene.make_params(width=w, height=h, bounces=int(rng.choice([-3, -2, -1, 0, 1, 2, 4])), num_aa_rays=int(rng.randint(1, 5)),
                            sobol_block=bool(rng.rand() < 0.5), bilinear=bool(rng.rand() < 0.5), normal_mapped=bool(rng.rand() < 0.5),
                            depth_of_field=bool(rng.rand() < 0.4), aperture=float(rng.uniform(0.001, 0.05)), focal_distance=float(rng.uniform(1.0, 3.0)),
                            whitted=whitted, num_light_rays=int(rng.randint(1, 6)), use_alpha=bool(rng.rand() < 0.7),
                            filter_width=float(rng.uniform(1.0, 2.0)), attenuation=float(rng.uniform(0.5, 1.5)), intensity=float(rng.uniform(0.5, 2.0)),
                            termination_prob=float(rng.uniform(0.1, 0.6)), pass_seed=seed)
    g = soup_gpu.render_pass(prm, cam, lights)
    c, _ = oracle.render_pass(textured_soup, prm, cam, lights, nthreads=8)
    _assert_image_exact(g, c, "seed %d" % seed)
