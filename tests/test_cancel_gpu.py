"""PathTraceRenderer::stop() while a pass is in flight (reference src/PathTraceRenderer.cpp:918-930: the workers poll m_bForceExit
before every sample, :709 and :736, and stop() returns as soon as they have): ptgpu_cancel through the C ABI and FW::PathTraceRenderer::stop()
of the host mirror.  A cancelled pass ends within milliseconds instead of seconds, leaves a finite image made of whole batches of
blocks, and the next pass on the same context is bit-exact against the oracle."""
import os
import time

import numpy as np
import pytest

from pathtracer_amd import capi, hostapi, scene

pytestmark = pytest.mark.gpu

W, H, SPP = 1920, 1080, 2048  # 4.2 G samples: ~2.5 s on one MI355X if nobody stops it


def _params(w, h, spp, seed):
    return scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=spp, sobol_block=False, filter_width=1.5, termination_prob=0.2,
                             pass_seed=seed)


@pytest.fixture()
def small_batches():
    """32 Mi samples per integrator launch: a 2048-spp batch is 16 blocks (~20 ms), so a pass stopped after 150 ms has whole batches in it."""
    old = os.environ.get("PTGPU_PT_BATCH_MSAMPLES")
    os.environ["PTGPU_PT_BATCH_MSAMPLES"] = "32"
    yield
    if old is None:
        del os.environ["PTGPU_PT_BATCH_MSAMPLES"]
    else:
        os.environ["PTGPU_PT_BATCH_MSAMPLES"] = old


def test_cancel_ends_a_long_pass_and_leaves_the_context_usable(gpu_ctx, head, oracle, small_batches):
    cam, lights = scene.make_camera(width=W, height=H), [scene.make_light()]
    s = capi.Scene(gpu_ctx, head["sd"])
    d = gpu_ctx.malloc(W * H * 16)
    try:
        # how long the pass takes when it runs to its end (1/8 of the samples, scaled): the bound below is relative to this box
        capi._check(gpu_ctx.lib.ptgpu_memset(gpu_ctx.handle, d, 0, W * H * 16))
        gpu_ctx.sync()
        t0 = time.perf_counter()
        s.render_pass_dev(_params(W, H, SPP // 8, 1), cam, lights, d)
        gpu_ctx.sync()
        full_s = (time.perf_counter() - t0) * 8
        assert full_s > 1.0, "the pass is too short (%.2f s) for this test to say anything" % full_s

        capi._check(gpu_ctx.lib.ptgpu_memset(gpu_ctx.handle, d, 0, W * H * 16))
        gpu_ctx.reset_counters()
        gpu_ctx.sync()
        t0 = time.perf_counter()
        s.render_pass_dev(_params(W, H, SPP, 2), cam, lights, d)  # only enqueues
        time.sleep(0.15)
        t1 = time.perf_counter()
        gpu_ctx.cancel()
        gpu_ctx.sync()
        t2 = time.perf_counter()
        assert t1 - t0 < 0.5 * full_s, "the pass was (nearly) over before it was cancelled: %.2f of %.2f s" % (t1 - t0, full_s)
        assert t2 - t1 < 0.1, "cancel + drain took %.3f s (a full pass: %.2f s)" % (t2 - t1, full_s)
        img = gpu_ctx.d2h(d, (H, W, 4), np.float32)
        rays = gpu_ctx.counters()["rays"]
        # Gaussian mode adds one normalised colour of weight 1 per pixel (reference :780-787): whole batches only => every weight is 0 or 1
        wgt = img[..., 3]
        assert np.isfinite(img).all() and set(np.unique(wgt).tolist()) <= {0.0, 1.0}
        done = float(wgt.mean())
        assert 0.0 < done < 0.5, "fraction of the frame accumulated before the cancel: %.3f" % done
        assert img[..., :3][wgt == 0].max() == 0.0 and img[..., :3].min() >= 0.0 and img[..., :3].max() <= 1.0
        assert 0 < rays < 0.5 * (W * H * SPP * 5.4)
        # blocks are accumulated batch after batch in block order: the finished part is a prefix of the block list
        bx = (W + 31) // 32
        blk_done = np.array([wgt[(b // bx) * 32, (b % bx) * 32] for b in range(bx * ((H + 31) // 32))])
        # (the batch whose accumulation kernel the cancel word reached half-way may be partly there: its blocks test the word one by one)
        last = int(np.nonzero(blk_done)[0].max()) + 1
        assert blk_done[:(last - 1) // 16 * 16].all() and not blk_done[(last + 15) // 16 * 16:].any()

        # the next pass on the same context is not affected: bit-exact against the oracle
        w, h = 128, 72
        cam2 = scene.make_camera(width=w, height=h)
        for sobol_block in (True, False):
            prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=3, sobol_block=sobol_block, filter_width=1.5, pass_seed=5)
            g = s.render_pass(prm, cam2, lights)
            c, _ = oracle.render_pass(head["sd"], prm, cam2, lights, nthreads=8)
            assert c[..., 3].min() > 0 and np.array_equal(g.view(np.uint32), c.view(np.uint32))
        # a cancel with nothing in flight cancels nothing that comes later
        gpu_ctx.cancel()
        prm = scene.make_params(width=w, height=h, bounces=2, num_aa_rays=2, sobol_block=True, pass_seed=6)
        g = s.render_pass(prm, cam2, lights)
        c, _ = oracle.render_pass(head["sd"], prm, cam2, lights, nthreads=8)
        assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    finally:
        gpu_ctx.free(d)
        s.close()


def test_host_renderer_stop_returns_promptly(small_batches):
    """FW::PathTraceRenderer::startPathTracingProcess on a pass that would run for seconds, stop() 150 ms later."""
    rt = hostapi.HostRayTracer(scene.HEAD_MESH, sah=True)
    try:
        prm = _params(W, H, SPP, 0)
        accum, stop_ms, was_running = rt.path_trace_then_stop(prm, [scene.make_light()], stop_after_ms=150)
        assert was_running, "the first pass had finished before stop() was called"
        assert stop_ms < 100.0, "stop() took %.1f ms" % stop_ms
        assert np.isfinite(accum).all() and set(np.unique(accum[..., 3]).tolist()) <= {0.0, 1.0} and 0.0 < accum[..., 3].mean() < 0.5
        # and the renderer starts again afterwards
        w, h = 64, 48
        small = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True)
        a2, _, _, passes = rt.path_trace(small, [scene.make_light()], passes=1)
        assert passes == 1 and a2[..., 3].min() > 0
    finally:
        rt.close()
