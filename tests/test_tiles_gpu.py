"""Image tiles across GPUs through the C ABI (include/ptgpu.h "image tiles", csrc/pt_multigpu.hip): ranks render round-robin shares
of the reference's 32x32 blocks (src/PathTraceRenderer.cpp:816-841) on the HIP path, the tiles are packed, exchanged and unpacked,
and every rank must hold the 1-GPU image bit for bit -- including Sobol-block samples that land in another rank's block.

On the 1-GPU box the ranks are emulated by several contexts on one device and the exchange is a device copy (pack / unpack kernels,
foreign-sample lists and ownership test are the real ones); the RCCL calls are exercised with a 1-rank communicator; the genuine
2-rank RCCL run needs two devices and is skipped otherwise."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd import capi, scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _empty_scene():
    return dict(pos=np.zeros((0, 3, 3), np.float32), nrm=np.zeros((0, 3, 3), np.float32), uv=np.zeros((0, 3, 2), np.float32),
                mat_id=np.zeros(0, np.int32), woop=np.zeros((0, 12), np.float32), materials=[], textures=[],
                nodes=np.zeros(0, capi.FLATNODE_DTYPE), list=np.zeros(0, np.int32))


def _emulate_ranks(sd, prms, cam, lights, world):
    """`world` contexts on device 0, each rendering its share of every pass in `prms` and gathering after each pass.
    Returns (per-rank final images, foreign entries seen, 1-context reference image)."""
    w, h = prms[0].width, prms[0].height
    nbytes = w * h * 16
    ctxs = [capi.Context(0) for _ in range(world)]
    scenes = [capi.Scene(c, sd) for c in ctxs]
    sf = capi.tiles_slice_floats(w, h, world)
    try:
        acc = [c.malloc(nbytes) for c in ctxs]
        slices = ctxs[0].malloc(sf * 4 * world)
        for c, a in zip(ctxs, acc):
            capi._check(c.lib.ptgpu_memset(c.handle, a, 0, nbytes))
        foreign_seen = 0
        for prm in prms:
            for r in range(world):
                scenes[r].render_pass_dev(prm, cam, lights, acc[r], block_begin=r, block_stride=world)
                st = ctxs[r].spill_status()
                assert st["dropped"] == 0
                foreign_seen += st["foreign_pending"]
                ctxs[r].tiles_pack_dev(w, h, world, r, acc[r], slices + r * sf * 4)
                ctxs[r].sync()
                assert ctxs[r].spill_status()["foreign_pending"] == 0  # packing hands the list over
            for r in range(world):
                ctxs[r].tiles_unpack_dev(w, h, world, r, slices, acc[r])
                ctxs[r].sync()
        imgs = [ctxs[r].d2h(acc[r], (h, w, 4), np.float32) for r in range(world)]
        # 1-GPU reference on a fresh buffer of context 0
        ref = ctxs[0].malloc(nbytes)
        capi._check(ctxs[0].lib.ptgpu_memset(ctxs[0].handle, ref, 0, nbytes))
        for prm in prms:
            scenes[0].render_pass_dev(prm, cam, lights, ref)
        ctxs[0].sync()
        assert ctxs[0].spill_status()["foreign_pending"] == 0  # one rank owns every block
        ref_img = ctxs[0].d2h(ref, (h, w, 4), np.float32)
        for c, a in zip(ctxs, acc):
            c.free(a)
        ctxs[0].free(slices)
        ctxs[0].free(ref)
    finally:
        for s in scenes:
            s.close()
        for c in ctxs:
            c.close()
    return imgs, foreign_seen, ref_img


@pytest.mark.parametrize("world", [2, 3])
def test_emulated_ranks_equal_one_gpu_image(head, world):
    w, h = 200, 120  # 7 x 4 blocks, clipped on both edges, not divisible by the world size
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    prms = [scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=3, sobol_block=sb, filter_width=1.5, pass_seed=seed)
            for seed, sb in ((1, True), (2, False), (3, True))]
    imgs, _, ref = _emulate_ranks(head["sd"], prms, cam, lights, world)
    assert ref[..., :3].max() > 0.01 and abs(float(ref[..., 3].mean()) - 3.0) < 1e-5
    for r, img in enumerate(imgs):
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "rank %d of %d differs from the 1-GPU image" % (r, world)


def test_cross_rank_out_of_block_samples_are_delivered(oracle):
    """Forced spill (see tests/test_multigpu_cpu.py): 32768 samples per pixel put the Sobol indices whose dimension-0 / dimension-1
    value is exactly 1.0f into every block, so block 0 (rank 0) has one sample on pixel x = 32 (block 1, rank 1) and one on y = 32
    (block 2, rank 0).  Empty scene: the weight channel (2^-15 per sample, exact sums) shows a lost or doubled sample."""
    w, h, spp = 64, 64, 32768
    cam = scene.make_camera(eye=(0, 0, 2.0), forward=(0, 0, -1), fov_deg=60.0, near=0.01, far=6.0, width=w, height=h)
    lights = [scene.make_light(position=(0, 0, 1.5), col0=(1, 0, 0), col1=(0, 1, 0), col2=(0, 0, 1))]
    prms = [scene.make_params(width=w, height=h, bounces=1, num_aa_rays=spp, sobol_block=True, pass_seed=seed) for seed in (4, 5)]
    imgs, foreign_seen, ref = _emulate_ranks(_empty_scene(), prms, cam, lights, 2)
    assert foreign_seen >= 2
    assert float(ref[..., 3].astype(np.float64).sum()) == 2.0 * w * h and (ref[..., 3] != 2.0).any()
    for r, img in enumerate(imgs):
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "rank %d lost or doubled an out-of-block sample" % r
    # and the 1-GPU image is the oracle's (deferred out-of-block samples, same order)
    c = np.zeros((h, w, 4), np.float32)
    for prm in prms:
        c, f, _ = oracle.render_pass_tiles(_empty_scene(), prm, cam, lights, c, 0, 1, nthreads=8)
    assert np.array_equal(ref.view(np.uint32), c.view(np.uint32))


def test_rccl_single_rank_communicator(gpu_ctx, head):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy through the ABI with a 1-rank communicator: the gather must
    leave the image as rendered (this is the code path N ranks take; only the peer count differs)."""
    w, h = 96, 64
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True, pass_seed=9)
    s = capi.Scene(gpu_ctx, head["sd"])
    try:
        with pytest.raises(capi.PtgpuError, match="communicator"):
            gpu_ctx.gather_tiles_dev(w, h, 1)
        a = gpu_ctx.malloc(w * h * 16)
        capi._check(gpu_ctx.lib.ptgpu_memset(gpu_ctx.handle, a, 0, w * h * 16))
        s.render_pass_dev(prm, cam, lights, a)
        before = gpu_ctx.d2h(a, (h, w, 4), np.float32)
        gpu_ctx.comm_init(1, 0, capi.create_comm_id())
        gpu_ctx.gather_tiles_dev(w, h, a)
        gpu_ctx.sync()
        after = gpu_ctx.d2h(a, (h, w, 4), np.float32)
        gpu_ctx.comm_destroy()
        gpu_ctx.free(a)
        assert np.array_equal(before.view(np.uint32), after.view(np.uint32)) and before[..., 3].min() > 0
    finally:
        s.close()


_TILEGATHER_SCRIPT = r"""
import sys
import numpy as np
import torch  # first: this process runs on torch's own HIP runtime (as bench.py does)
sys.path.insert(0, %(root)r)
from pathtracer_amd import capi, hostapi, multigpu, scene
mesh = scene.load_ptmesh()
nodes, lst, _ = hostapi.bvh_build(mesh["pos"], sah=True)
sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
          materials=scene.mesh_materials(mesh), textures=[], nodes=nodes, list=lst)
w, h = 200, 120
cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=False, filter_width=1.5, pass_seed=4)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)  # its own non-blocking stream: not torch's
s = capi.Scene(ctx, sd)
accum = torch.zeros(h * w * 4, dtype=torch.float32, device=dev)
torch.cuda.synchronize(dev)
s.render_pass_dev(prm, cam, lights, accum.data_ptr())
ctx.sync()
before = accum.cpu().numpy().reshape(h, w, 4).copy()
tg = multigpu.TileGather(w, h, 0, 1, dev, ctx=ctx)
for _ in range(3):  # repeated: a missing join between the two streams shows as a stale or half-written buffer
    tg.gather(accum)
ctx.sync()
after = accum.cpu().numpy().reshape(h, w, 4)
assert before[..., 3].min() == 1.0 and before[..., :3].max() > 0.01
assert np.array_equal(before.view(np.uint32), after.view(np.uint32)), "the image changed on its way through pack -> copy -> unpack"
s.close()
ctx.close()
print("ok")
"""


def test_tile_gather_class_with_a_context_on_its_own_stream(tmp_path):
    """multigpu.TileGather(ctx=...): pack / unpack run on the context's stream (a default Context owns a non-blocking stream), the
    transport on torch's; the class joins the two.  One rank (the world == 1 path: pack -> device copy on torch's stream -> unpack): the
    image survives the round trip unchanged.  In a process of its own, where torch is loaded before the library."""
    script = tmp_path / "tilegather.py"
    script.write_text(_TILEGATHER_SCRIPT % dict(root=ROOT))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout[-500:] + p.stderr[-1500:]


_RANK_SCRIPT = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from pathtracer_amd import capi, hostapi, scene
rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
mesh = scene.load_ptmesh()
nodes, lst, _ = hostapi.bvh_build(mesh["pos"], sah=True)
sd = dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=hostapi.woop_from_positions(mesh["pos"]),
          materials=scene.mesh_materials(mesh), textures=[], nodes=nodes, list=lst)
ctx = capi.Context(rank)
idf = os.path.join(tmp, "comm_id.bin")
if rank == 0:
    open(idf + ".tmp", "wb").write(capi.create_comm_id())
    os.rename(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):
    assert time.time() - t0 < 60, "no communicator id from rank 0"
    time.sleep(0.05)
ctx.comm_init(world, rank, open(idf, "rb").read())
s = capi.Scene(ctx, sd)
w, h = 200, 120
cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
a = ctx.malloc(w * h * 16)
capi._check(ctx.lib.ptgpu_memset(ctx.handle, a, 0, w * h * 16))
for seed, sb in ((1, True), (2, False)):
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=3, sobol_block=sb, filter_width=1.5, pass_seed=seed)
    s.render_pass_dev(prm, cam, lights, a, block_begin=rank, block_stride=world)
    ctx.gather_tiles_dev(w, h, a)
ctx.sync()
np.save(os.path.join(tmp, "rank%%d.npy" %% rank), ctx.d2h(a, (h, w, 4), np.float32))
if rank == 0:
    capi._check(ctx.lib.ptgpu_memset(ctx.handle, a, 0, w * h * 16))
    for seed, sb in ((1, True), (2, False)):
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=3, sobol_block=sb, filter_width=1.5, pass_seed=seed)
        s.render_pass_dev(prm, cam, lights, a)
    ctx.sync()
    np.save(os.path.join(tmp, "ref.npy"), ctx.d2h(a, (h, w, 4), np.float32))
ctx.comm_destroy()
s.close()
ctx.close()
"""


def test_two_gpus_rccl_gather_equals_one_gpu_image(tmp_path):
    """Two processes, two devices, the HIP render pass on each and ptgpu_gather_tiles_dev (ncclAllGather over xGMI) between them."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 8-GPU node); the 1-GPU box runs the emulated-rank tests above")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % dict(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path)]) for r in range(2)]
    rcs = [p.wait(timeout=240) for p in procs]
    assert rcs == [0, 0]
    ref = np.load(tmp_path / "ref.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("rank%d.npy" % r)).view(np.uint32), ref.view(np.uint32)), "rank %d" % r
