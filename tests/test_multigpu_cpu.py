"""N>1 path on CPU: world_size-2 gloo.  Each rank renders its round-robin share of the 32x32 blocks (here with the CPU oracle
standing in for the device pass, which takes the same block_begin/block_stride arguments), the tiles are gathered with
pathtracer_amd.multigpu.TileGather, and every rank must end up with an image bit-identical to the single-process pass."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from pathtracer_amd import multigpu, scene
    pos = scene.grid_scene(6, 6, seed=2)
    sd = dict(pos=pos, nrm=scene.flat_normals(pos), uv=np.zeros((pos.shape[0], 3, 2), np.float32), mat_id=np.zeros(pos.shape[0], np.int32),
              woop=po.woop_from_positions(pos), materials=[scene.make_material()], textures=[])
    sd["nodes"], sd["list"], _ = po.bvh_build(pos, True)
    cam = scene.make_camera(eye=(0, 0, 2.0), forward=(0, 0, -1), fov_deg=60.0, near=0.01, far=6.0, width=w, height=h)
    lights = [scene.make_light(position=(0, 0, 1.5), col0=(1, 0, 0), col1=(0, 1, 0), col2=(0, 0, 1))]
    prm = scene.make_params(width=w, height=h, bounces=1, num_aa_rays=2, sobol_block=True, pass_seed=4)
    nb = multigpu.num_blocks(w, h)
    accum = np.zeros((h, w, 4), np.float32)
    for b in range(rank, nb, world):  # what ptgpu_render_pass(block_begin=rank, block_stride=world) covers
        po.render_pass(sd, prm, cam, lights, accum=accum, block_begin=b, block_end=b + 1)
    tg = multigpu.TileGather(w, h, rank, world, torch.device("cpu"))
    assert sum(tg.counts) == w * h
    full = tg.gather(torch.from_numpy(accum).view(-1)).view(h, w, 4).numpy()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full)
    if rank == 0:
        ref, _ = po.render_pass(sd, prm, cam, lights)
        np.save(os.path.join(out_dir, "ref.npy"), ref)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_partition_covers_image_once():
    sys.path.insert(0, ROOT)
    from pathtracer_amd import multigpu
    for (w, h, world) in ((1920, 1080, 8), (100, 70, 3), (32, 32, 2), (33, 1, 4)):
        seen = np.zeros(w * h, np.int32)
        for r in range(world):
            seen[multigpu.rank_pixel_index(r, world, w, h)] += 1
        assert (seen == 1).all(), (w, h, world)


def test_two_rank_gloo_gather_bit_identical(tmp_path):
    w, h, world = 100, 70, 2  # clipped edge blocks, 12 blocks dealt 6 / 6
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npy")
    assert ref[..., :3].max() > 0
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "rank %d image differs from the single-process pass" % r
