"""The pooled form of the path-trace integrator (csrc/pt_integrator_pool.hip.h, environment PTGPU_PT_POOL=1: path state in LDS slots,
trace waves and shade waves of one workgroup exchanging slots through LDS rings) against the lane-owned form that ships as the default:
per-sample arithmetic is the same code, so accumulation images and ray counters must be bit-identical, and against the oracle directly
(reference src/PathTraceRenderer.cpp:236-551 tracePath, :706-787 pathTraceBlock).  Collected late: the path is experimental and measured
slower than the default."""
import os

import numpy as np
import pytest

from pathtracer_amd import capi, scene
from test_render_gpu import _soup_view, textured_soup  # noqa: F401  (the all-features soup scene and its view)

pytestmark = pytest.mark.gpu


def _render(s, ctx, prm, cam, lights, pooled):
    old = os.environ.get("PTGPU_PT_POOL")
    os.environ["PTGPU_PT_POOL"] = "1" if pooled else "0"
    try:
        ctx.reset_counters()
        img = s.render_pass(prm, cam, lights)
        st = ctx.spill_status()  # raises if the pooled launch's watchdog fired
        return img, ctx.counters()["rays"], st
    finally:
        if old is None:
            os.environ.pop("PTGPU_PT_POOL", None)
        else:
            os.environ["PTGPU_PT_POOL"] = old


@pytest.mark.parametrize("sobol_block,bounces,dof", [(True, -1, False), (False, -2, False), (True, 2, True), (False, 0, False)])
def test_pooled_integrator_equals_lane_owned(gpu_ctx, head, sobol_block, bounces, dof):
    w, h = 200, 120  # clipped blocks on both edges: padded sample ids are drawn and skipped
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light(), scene.make_light(position=(0.5, 1.2, 0.8))]
    prm = scene.make_params(width=w, height=h, bounces=bounces, num_aa_rays=3, sobol_block=sobol_block, filter_width=1.5, termination_prob=0.3,
                            depth_of_field=dof, aperture=0.02, focal_distance=1.5, pass_seed=5)
    s = capi.Scene(gpu_ctx, head["sd"])
    try:
        a, ra, _ = _render(s, gpu_ctx, prm, cam, lights, False)
        b, rb, _ = _render(s, gpu_ctx, prm, cam, lights, True)
    finally:
        s.close()
    assert a[..., :3].max() > 0.01
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ra == rb


def test_pooled_integrator_textured_equals_oracle(gpu_ctx, textured_soup, oracle):  # noqa: F811
    """Textures, normal maps, bilinear filtering, alpha cutout, two lights, depth of field: the pooled form against the oracle itself."""
    sd = textured_soup
    w, h = 96, 72
    cam, lights = _soup_view(w, h)
    prm = scene.make_params(width=w, height=h, bounces=-2, num_aa_rays=3, sobol_block=False, bilinear=True, normal_mapped=True,
                            depth_of_field=True, aperture=0.02, focal_distance=2.0, filter_width=1.5, attenuation=0.8,
                            termination_prob=0.3, pass_seed=11)
    s = capi.Scene(gpu_ctx, sd)
    try:
        g, rays, _ = _render(s, gpu_ctx, prm, cam, lights, True)
    finally:
        s.close()
    c, ctr = oracle.render_pass(sd, prm, cam, lights, nthreads=8)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32)) and rays == ctr["rays"]


def test_pooled_integrator_empty_scene(gpu_ctx):
    """Every ray misses the scene's box at once: slots go round the rings as fast as they can (the ring's lap tags and the trace waves'
    in-place hand-over are what this exercises)."""
    sd = dict(pos=np.zeros((0, 3, 3), np.float32), nrm=np.zeros((0, 3, 3), np.float32), uv=np.zeros((0, 3, 2), np.float32),
              mat_id=np.zeros(0, np.int32), woop=np.zeros((0, 12), np.float32), materials=[], textures=[],
              nodes=np.zeros(0, capi.FLATNODE_DTYPE), list=np.zeros(0, np.int32))
    w, h = 96, 64
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=64, sobol_block=True, pass_seed=2)
    s = capi.Scene(gpu_ctx, sd)
    try:
        a, ra, _ = _render(s, gpu_ctx, prm, cam, lights, False)
        b, rb, _ = _render(s, gpu_ctx, prm, cam, lights, True)
    finally:
        s.close()
    assert ra == rb == w * h * 64 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_pooled_integrator_watchdog_reports_and_recovers(gpu_ctx, head, monkeypatch):
    """The watchdog's exit path: with a bound of 2^4 ticks every wave that has to wait at all gives the launch up -- ptgpu_render_pass
    and ptgpu_spill_status return PTGPU_E_INTERNAL (the image is incomplete), the device is not hung, and the next passes on the same
    context (pooled with the normal bound, and the default integrator) are bit-identical to each other again."""
    w, h = 200, 120
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=8, sobol_block=True, pass_seed=3)
    s = capi.Scene(gpu_ctx, head["sd"])
    try:
        monkeypatch.setenv("PTGPU_PT_POOL", "1")
        monkeypatch.setenv("PTGPU_POOL_WATCHDOG_LOG2", "4")
        with pytest.raises(capi.PtgpuError, match="pooled integrator"):
            s.render_pass(prm, cam, lights)
        with pytest.raises(capi.PtgpuError, match="pooled integrator"):
            gpu_ctx.spill_status()
        monkeypatch.delenv("PTGPU_POOL_WATCHDOG_LOG2")
        b = s.render_pass(prm, cam, lights)
        assert gpu_ctx.spill_status()["dropped"] == 0  # the abort word is per pass: cleared
        monkeypatch.setenv("PTGPU_PT_POOL", "0")
        a = s.render_pass(prm, cam, lights)
        gpu_ctx.spill_status()
    finally:
        s.close()
    assert a[..., 3].min() > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
