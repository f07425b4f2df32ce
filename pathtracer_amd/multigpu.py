"""Image-tile sharding of the path-trace pass across the GPUs of one node.

The scene is replicated; the 32x32 blocks of the reference's tiling (src/PathTraceRenderer.cpp:816-841, row-major block
index) are dealt round-robin: rank r of N renders blocks r, r+N, r+2N, ... (ptgpu_render_pass(block_begin=r,
block_stride=N)), so every rank holds a disjoint set of pixels of the accumulation image.  After a pass the tiles are
gathered over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests): every rank packs
its own pixels into one contiguous buffer, ONE all_gather moves them (each peer's slice travels over its own
point-to-point link), and the pixels are scattered back into the full image.  Nothing is summed, so the N-GPU image is
bit-identical to the 1-GPU image (a sum all-reduce would re-associate nothing here either, but would move N times the
bytes).
"""
import numpy as np
import torch
import torch.distributed as dist

BLOCK = 32


def num_blocks(width, height):
    return ((width + BLOCK - 1) // BLOCK) * ((height + BLOCK - 1) // BLOCK)


def block_rect(idx, width, height):
    bx = (idx % ((width + BLOCK - 1) // BLOCK)) * BLOCK
    by = (idx // ((width + BLOCK - 1) // BLOCK)) * BLOCK
    return bx, by, min(BLOCK, width - bx), min(BLOCK, height - by)


def rank_pixel_index(rank, world, width, height):
    """Flat pixel indices (y*width + x) of every pixel in the blocks rank `rank` of `world` owns, block after block."""
    out = []
    for idx in range(rank, num_blocks(width, height), world):
        bx, by, bw, bh = block_rect(idx, width, height)
        ys, xs = np.meshgrid(np.arange(by, by + bh), np.arange(bx, bx + bw), indexing="ij")
        out.append((ys * width + xs).reshape(-1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, np.int64)


class TileGather:
    """Precomputed pack / unpack indices for one image size and world size."""

    def __init__(self, width, height, rank, world, device, stage_on_cpu=False):
        """stage_on_cpu: run the collective on host copies of the packed tiles (gloo rehearsals of a GPU job; RCCL takes the
        device buffers directly)."""
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.stage_on_cpu = stage_on_cpu and torch.device(device).type != "cpu"
        idx = [rank_pixel_index(r, world, width, height) for r in range(world)]
        self.counts = [int(i.size) for i in idx]
        self.pad = max(self.counts) if self.counts else 0
        self.index = [torch.from_numpy(i).to(device) for i in idx]
        self.packed = torch.zeros(self.pad, 4, dtype=torch.float32, device=device)
        self.gathered = torch.zeros(world, self.pad, 4, dtype=torch.float32, device=device)

    def gather(self, accum):
        """accum: float32 tensor [height*width*4] (or any view of it) holding this rank's tiles.  Returns the same tensor
        with every rank's tiles filled in (all ranks get the full image)."""
        img = accum.view(-1, 4)
        if self.world == 1:
            return accum
        mine = self.index[self.rank]
        self.packed[: mine.numel()] = img[mine]
        if self.stage_on_cpu:
            host = torch.empty(self.gathered.shape, dtype=torch.float32)
            dist.all_gather_into_tensor(host.view(-1), self.packed.cpu().view(-1))
            self.gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(self.gathered.view(-1), self.packed.view(-1))
        for r in range(self.world):
            if r != self.rank:
                img[self.index[r]] = self.gathered[r, : self.counts[r]]
        return accum
