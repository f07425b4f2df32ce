"""Image-tile sharding of the path-trace pass across the GPUs of one node -- Python callers of the C ABI's tile
functions (include/ptgpu.h "image tiles", csrc/pt_multigpu.hip).

The scene is replicated; the 32x32 blocks of the reference's tiling (src/PathTraceRenderer.cpp:816-841, row-major block index) are
dealt round-robin: rank r of N renders blocks r, r+N, r+2N, ... (ptgpu_render_pass_dev(block_begin=r, block_stride=N)).  After a
pass every rank's blocks are exchanged with ONE all-gather; nothing is summed, so the N-GPU image is bit-identical to the 1-GPU one.
Sobol-block samples that land one pixel outside their block travel in a fixed section of each rank's slice and are added by every
rank after the exchange, per pixel in (source block, sample) order -- the order in which a single process adds them (added before
the exchange they would be overwritten by the owner's tile and lost, or summed in an order that depends on the rank count).

  DeviceTileGather  the product path: ptgpu_gather_tiles_dev = pack kernel + ncclAllGather (RCCL over xGMI) + unpack kernels on the
                    context's stream; this class only creates the communicator (id broadcast over torch.distributed).
  TileGather        the same wire format with torch.distributed as the transport: for CPU tensors (gloo tests of the protocol, where
                    the oracle stands in for the device pass) and for rehearsals of a GPU job on fewer GPUs than ranks (device pack /
                    unpack kernels around a gloo all_gather of host copies).
"""
import numpy as np
import torch
import torch.distributed as dist

BLOCK = 32
TILE_PIXELS = BLOCK * BLOCK
FOREIGN_CAPACITY = 65536  # csrc/pt_context.h
FOREIGN_FLOATS = 4 + FOREIGN_CAPACITY * 8


def num_blocks(width, height):
    return ((width + BLOCK - 1) // BLOCK) * ((height + BLOCK - 1) // BLOCK)


def block_rect(idx, width, height):
    bx = (idx % ((width + BLOCK - 1) // BLOCK)) * BLOCK
    by = (idx // ((width + BLOCK - 1) // BLOCK)) * BLOCK
    return bx, by, min(BLOCK, width - bx), min(BLOCK, height - by)


def blocks_per_rank(width, height, world):
    return (num_blocks(width, height) + world - 1) // world


def slice_floats(width, height, world):
    """Floats in one rank's slice (== ptgpu_tiles_slice_floats)."""
    return blocks_per_rank(width, height, world) * TILE_PIXELS * 4 + FOREIGN_FLOATS


def rank_pixel_index(rank, world, width, height):
    """Flat pixel indices (y*width + x) of every pixel in the blocks rank `rank` of `world` owns, block after block."""
    out = []
    for idx in range(rank, num_blocks(width, height), world):
        bx, by, bw, bh = block_rect(idx, width, height)
        ys, xs = np.meshgrid(np.arange(by, by + bh), np.arange(bx, bx + bw), indexing="ij")
        out.append((ys * width + xs).reshape(-1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, np.int64)


def rank_slot_index(rank, world, width, height):
    """(slot positions, pixel indices): pixel `pix[k]` of the image sits at float4 position `pos[k]` of rank's slice
    (block j of the rank in slot j, row-major inside the 32x32 slot)."""
    pos, pix = [], []
    for j, idx in enumerate(range(rank, num_blocks(width, height), world)):
        bx, by, bw, bh = block_rect(idx, width, height)
        ys, xs = np.meshgrid(np.arange(bh), np.arange(bw), indexing="ij")
        pos.append((j * TILE_PIXELS + ys * BLOCK + xs).reshape(-1))
        pix.append(((by + ys) * width + bx + xs).reshape(-1))
    if not pos:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(pos).astype(np.int64), np.concatenate(pix).astype(np.int64)


class DeviceTileGather:
    """ptgpu_gather_tiles_dev behind a communicator created from an id that rank 0 broadcasts over torch.distributed."""

    def __init__(self, ctx, width, height, rank, world):
        from . import capi
        self.ctx, self.width, self.height, self.rank, self.world = ctx, width, height, rank, world
        if world > 1:
            box = [capi.create_comm_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.comm_init(world, rank, box[0])

    def gather(self, accum):
        """accum: float32 CUDA tensor [height*width*4] holding this rank's blocks; enqueued on the context's stream."""
        if self.world > 1:
            self.ctx.gather_tiles_dev(self.width, self.height, accum.data_ptr())
        return accum

    def close(self):
        if self.world > 1:
            self.ctx.comm_destroy()


class TileGather:
    """The tile exchange with torch.distributed as the transport (same slice layout as the device path)."""

    def __init__(self, width, height, rank, world, device, ctx=None):
        """ctx given (a capi.Context on `device`): pack / unpack run as the library's kernels and only the all_gather goes through
        torch.distributed, staged on the host when the backend is gloo.  ctx None: CPU tensors, everything in torch.
        Stream order with ctx given: pack / unpack run on the CONTEXT's stream, the transport on torch's current stream, and the two may
        differ (a default Context owns a non-blocking stream), so gather() joins them on the host: the context's stream is drained after
        the pack and torch's stream before the unpack.  The caller's render pass and the gather are ordered by the context's stream."""
        self.width, self.height, self.rank, self.world, self.ctx = width, height, rank, world, ctx
        self.device = torch.device(device)
        self.floats = slice_floats(width, height, world)
        self.tile_f4 = blocks_per_rank(width, height, world) * TILE_PIXELS
        self.send = torch.zeros(self.floats, dtype=torch.float32, device=self.device)
        self.recv = torch.zeros(world * self.floats, dtype=torch.float32, device=self.device)
        if ctx is None:
            idx = [rank_slot_index(r, world, width, height) for r in range(world)]
            self.pos = [torch.from_numpy(p) for p, _ in idx]
            self.pix = [torch.from_numpy(q) for _, q in idx]
            self.counts = [int(q.numel()) for q in self.pix]

    def gather(self, accum, foreign=None):
        """accum: float32 tensor [height*width*4] holding this rank's blocks.  foreign (ctx None only): float32 array [n,8] of this
        rank's out-of-block samples for other ranks (oracle.render_pass_tiles).  Every rank ends with the full image."""
        if self.world == 1 and foreign is None and self.ctx is None:
            return accum
        if self.ctx is not None:
            self.ctx.tiles_pack_dev(self.width, self.height, self.world, self.rank, accum.data_ptr(), self.send.data_ptr())
            self.ctx.sync()  # `send` is complete before the transport (on torch's stream) reads it
            if self.world > 1:
                if dist.get_backend() == "gloo":
                    host = torch.empty(self.recv.shape, dtype=torch.float32)
                    dist.all_gather_into_tensor(host, self.send.cpu())
                    self.recv.copy_(host)
                else:
                    dist.all_gather_into_tensor(self.recv, self.send)
            else:
                self.recv.copy_(self.send)
            torch.cuda.current_stream(self.device).synchronize()  # `recv` is complete before the unpack (on the context's stream) reads it
            self.ctx.tiles_unpack_dev(self.width, self.height, self.world, self.rank, self.recv.data_ptr(), accum.data_ptr())
            return accum
        img = accum.view(-1, 4)
        s4 = self.send.view(-1, 4)
        s4.zero_()
        s4[self.pos[self.rank]] = img[self.pix[self.rank]]
        n = 0 if foreign is None else min(int(foreign.shape[0]), FOREIGN_CAPACITY)
        sec = self.send[self.tile_f4 * 4:]
        sec[0:1].view(torch.int32)[0] = n
        if n:
            sec[4:4 + n * 8] = torch.from_numpy(np.ascontiguousarray(foreign[:n], np.float32).reshape(-1))
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv, self.send)
        else:
            self.recv.copy_(self.send)
        r4 = self.recv.view(self.world, -1, 4)
        for r in range(self.world):
            if r != self.rank:
                img[self.pix[r]] = r4[r][self.pos[r]]
        # every rank's out-of-block samples, per pixel in (source block, sample) order, one at a time: the order of a single process
        ents = []
        for r in range(self.world):
            sec = self.recv.view(self.world, -1)[r][self.tile_f4 * 4:]
            cnt = int(sec[0:1].view(torch.int32)[0])
            ents.append(sec[4:4 + cnt * 8].view(cnt, 8))
        ent = torch.cat(ents) if ents else self.recv.new_zeros((0, 8))
        if ent.shape[0]:
            keys = ent[:, 4:7].contiguous().view(torch.int32).numpy().astype(np.int64)  # pixel, source block, sample
            for e in np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0])):
                img[int(keys[e, 0])] += ent[int(e), :4]
        return accum
