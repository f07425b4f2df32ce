// Multi-GPU image tiles behind the C ABI (include/ptgpu.h "tiles"): one process per GPU, scene replicated, the reference's 32x32 blocks
// (src/PathTraceRenderer.cpp:816-841, row-major block index) dealt round-robin -- rank r of N renders blocks r, r+N, ... -- and after a pass
// every rank's tiles are exchanged with ONE all-gather over RCCL (xGMI: each rank's slice travels over its own point-to-point link).
// Nothing is summed, so the N-GPU accumulation image equals the 1-GPU image bit for bit.  The reference has no counterpart: its worker
// threads write disjoint blocks of one shared image (src/PathTraceRenderer.cpp:727-732, :785-787).
//
// Wire format of one rank's slice (floats): blocks_per_rank x 1024 pixels x 4 (block j of the rank in slot j, row-major inside the
// block, pixels outside an edge-clipped block zero), then the out-of-block section: one float4 header (count as int bits) and
// FOREIGN_CAPACITY entries of two float4 (pt_context.h ForeignSpill) -- the Sobol-block samples of the rank's pass that landed one
// pixel outside their block (ALL of them, whoever owns the pixel).  After the exchange every rank adds every rank's entries to the full
// image, per pixel in (source block, sample) order (ptgpu_apply_spills_sorted) -- the order in which ONE process adds them -- so all
// ranks end with the image one GPU renders, bit for bit, and no sample is lost.
//
// RCCL is loaded at run time (dlopen of librccl.so.1, found through this library's RUNPATH or already in the process with torch), so
// single-GPU users of libptgpu.so do not load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "pt_context.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return PTGPU_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(PTGPU_E_UNSUPPORTED, "RCCL is not available: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        dlclose(h);
        return fail(PTGPU_E_UNSUPPORTED, "librccl.so lacks a required symbol");
    }
    g_rccl = r;
    return PTGPU_OK;
}
#define RCCL_TRY(expr)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return fail(PTGPU_E_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));         \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PTGPU_COMM_ID_BYTES, "the ABI's id is an ncclUniqueId");

constexpr int TILE = 32, TILE_PIXELS = TILE * TILE;
constexpr int FOREIGN_FLOATS = 4 + FOREIGN_CAPACITY * 8;

struct TileGrid {
    int32_t width, height, blocks_x, num_blocks, world, blocks_per_rank;
};
int make_grid(int32_t width, int32_t height, int32_t world, TileGrid* g) {
    if (width <= 0 || height <= 0 || (long long)width * height > 0x7fffffffLL) return fail(PTGPU_E_INVALID, "image size %dx%d", width, height);
    if (world < 1) return fail(PTGPU_E_INVALID, "world size %d", world);
    g->width = width; g->height = height; g->world = world;
    g->blocks_x = (width + TILE - 1) / TILE;
    g->num_blocks = g->blocks_x * ((height + TILE - 1) / TILE);
    g->blocks_per_rank = (g->num_blocks + world - 1) / world;
    return PTGPU_OK;
}
long long slice_floats(const TileGrid& g) { return (long long)g.blocks_per_rank * TILE_PIXELS * 4 + FOREIGN_FLOATS; }

// slot j of rank r <-> block r + j * world; one 256-thread workgroup per slot, four pixels per thread
__device__ __forceinline__ bool slot_pixel(const TileGrid& g, int rank, int j, int p, size_t& image_index) {
    const int idx = rank + j * g.world;
    if (idx >= g.num_blocks) return false;
    const int bx = (idx % g.blocks_x) * TILE, by = (idx / g.blocks_x) * TILE;
    const int x = bx + (p & (TILE - 1)), y = by + (p >> 5);
    if (x >= g.width || y >= g.height) return false;
    image_index = (size_t)y * g.width + x;
    return true;
}

constexpr int PACK_SECTION_GROUPS = 32;  // workgroups that move the out-of-block section

__global__ __launch_bounds__(256) void k_tiles_pack(TileGrid g, int rank, const float4* __restrict__ accum, float4* __restrict__ slice,
                                                    const ForeignSpill* __restrict__ foreign, const unsigned long long* __restrict__ counters) {
    const int j = blockIdx.x;
    if (j < g.blocks_per_rank) {
        for (int p = threadIdx.x; p < TILE_PIXELS; p += 256) {
            size_t at;
            slice[(size_t)j * TILE_PIXELS + p] = slot_pixel(g, rank, j, p, at) ? accum[at] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        return;
    }
    // the last workgroups move the out-of-block section (the pending list is emptied by k_tiles_pack_done, after this kernel)
    float4* sec = slice + (size_t)g.blocks_per_rank * TILE_PIXELS;
    unsigned long long n = counters[CTR_FOREIGN];
    if (n > (unsigned long long)FOREIGN_CAPACITY) n = FOREIGN_CAPACITY;
    const float4* src = reinterpret_cast<const float4*>(foreign);
    const int part = j - g.blocks_per_rank;
    for (int e = part * 256 + threadIdx.x; e < FOREIGN_CAPACITY * 2; e += PACK_SECTION_GROUPS * 256)
        sec[1 + e] = (unsigned long long)(e >> 1) < n ? src[e] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (part == 0 && threadIdx.x == 0) sec[0] = make_float4(__int_as_float((int)n), 0.0f, 0.0f, 0.0f);
}
__global__ void k_tiles_pack_done(unsigned long long* counters) {
    if (threadIdx.x == 0 && blockIdx.x == 0) counters[CTR_FOREIGN] = 0ull;
}

// every other rank's tiles into the full image (this rank's own pixels are already there)
__global__ __launch_bounds__(256) void k_tiles_unpack(TileGrid g, int rank, long long slice_f4, const float4* __restrict__ gathered,
                                                      float4* __restrict__ accum) {
    const int j = blockIdx.x, r = blockIdx.y;
    if (r == rank) return;
    const float4* slice = gathered + (size_t)r * slice_f4;
    for (int p = threadIdx.x; p < TILE_PIXELS; p += 256) {
        size_t at;
        if (slot_pixel(g, r, j, p, at)) accum[at] = slice[(size_t)j * TILE_PIXELS + p];
    }
}

}  // namespace

struct ptgpu_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    float* d_send = nullptr;  // this rank's slice
    float* d_recv = nullptr;  // world slices
    long long capacity_floats = 0;  // per slice
};

void ptgpu_comm_release(ptgpu_context* c) {
    if (!c || !c->comm) return;
    if (c->comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm->comm);
    if (c->comm->d_send) hipFree(c->comm->d_send);
    if (c->comm->d_recv) hipFree(c->comm->d_recv);
    delete c->comm;
    c->comm = nullptr;
}

extern "C" int ptgpu_comm_create_id(void* out_id) {
    if (!out_id) return fail(PTGPU_E_INVALID, "out_id is NULL");
    int rc = load_rccl();
    if (rc != PTGPU_OK) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof id);
    return PTGPU_OK;
}

extern "C" int ptgpu_comm_init(ptgpu_context* c, int32_t world, int32_t rank, const void* id_bytes) {
    if (!c || !id_bytes) return fail(PTGPU_E_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(PTGPU_E_INVALID, "rank %d of %d", rank, world);
    int rc = load_rccl();
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    ptgpu_comm_release(c);
    ptgpu_comm* m = new ptgpu_comm();
    m->world = world; m->rank = rank;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&m->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete m;
        return fail(PTGPU_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    c->comm = m;
    return PTGPU_OK;
}

extern "C" int ptgpu_comm_destroy(ptgpu_context* c) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    ptgpu_comm_release(c);
    return PTGPU_OK;
}

extern "C" int ptgpu_tiles_slice_floats(int32_t width, int32_t height, int32_t world, int64_t* out_floats) {
    if (!out_floats) return fail(PTGPU_E_INVALID, "out_floats is NULL");
    TileGrid g;
    int rc = make_grid(width, height, world, &g);
    if (rc != PTGPU_OK) return rc;
    *out_floats = slice_floats(g);
    return PTGPU_OK;
}

extern "C" int ptgpu_tiles_pack_dev(ptgpu_context* c, int32_t width, int32_t height, int32_t world, int32_t rank, const float* d_accum,
                                    float* d_slice) {
    if (!c || !d_accum || !d_slice) return fail(PTGPU_E_INVALID, "NULL argument");
    TileGrid g;
    int rc = make_grid(width, height, world, &g);
    if (rc != PTGPU_OK) return rc;
    if (rank < 0 || rank >= world) return fail(PTGPU_E_INVALID, "rank %d of %d", rank, world);
    HIP_TRY(hipSetDevice(c->device));
    rc = ptgpu_ensure_spill_lists(c);  // a caller may gather before its first Sobol-block pass
    if (rc != PTGPU_OK) return rc;
    hipLaunchKernelGGL(k_tiles_pack, dim3((unsigned)g.blocks_per_rank + PACK_SECTION_GROUPS), dim3(256), 0, c->stream, g, (int)rank,
                       reinterpret_cast<const float4*>(d_accum), reinterpret_cast<float4*>(d_slice), c->d_foreign, c->d_counters);
    hipLaunchKernelGGL(k_tiles_pack_done, dim3(1), dim3(64), 0, c->stream, c->d_counters);
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

extern "C" int ptgpu_tiles_unpack_dev(ptgpu_context* c, int32_t width, int32_t height, int32_t world, int32_t rank, const float* d_slices,
                                      float* d_accum) {
    if (!c || !d_accum || !d_slices) return fail(PTGPU_E_INVALID, "NULL argument");
    TileGrid g;
    int rc = make_grid(width, height, world, &g);
    if (rc != PTGPU_OK) return rc;
    if (rank < 0 || rank >= world) return fail(PTGPU_E_INVALID, "rank %d of %d", rank, world);
    HIP_TRY(hipSetDevice(c->device));
    const long long sf4 = slice_floats(g) / 4;
    if (world > 1)
        hipLaunchKernelGGL(k_tiles_unpack, dim3((unsigned)g.blocks_per_rank, (unsigned)world), dim3(256), 0, c->stream, g, (int)rank, sf4,
                           reinterpret_cast<const float4*>(d_slices), reinterpret_cast<float4*>(d_accum));
    HIP_TRY(hipGetLastError());
    // every rank's out-of-block samples, per pixel in (source block, sample) order
    const SpillLists lists{reinterpret_cast<const float4*>(d_slices) + (size_t)g.blocks_per_rank * TILE_PIXELS + 1, sf4, FOREIGN_CAPACITY, world, nullptr};
    return ptgpu_apply_spills_sorted(c, lists, d_accum, width, height, nullptr);
}

extern "C" int ptgpu_gather_tiles_dev(ptgpu_context* c, int32_t width, int32_t height, float* d_accum) {
    if (!c || !d_accum) return fail(PTGPU_E_INVALID, "NULL argument");
    if (!c->comm) return fail(PTGPU_E_INVALID, "no communicator: call ptgpu_comm_init first");
    ptgpu_comm* m = c->comm;
    TileGrid g;
    int rc = make_grid(width, height, m->world, &g);
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const long long sf = slice_floats(g);
    if (sf > m->capacity_floats) {  // grow-only exchange buffers (sized by the first gather of a render)
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (m->d_send) { HIP_TRY(hipFree(m->d_send)); m->d_send = nullptr; }
        if (m->d_recv) { HIP_TRY(hipFree(m->d_recv)); m->d_recv = nullptr; }
        m->capacity_floats = 0;
        HIP_TRY(hipMalloc(&m->d_send, (size_t)sf * sizeof(float)));
        HIP_TRY(hipMalloc(&m->d_recv, (size_t)sf * sizeof(float) * m->world));
        m->capacity_floats = sf;
    }
    rc = ptgpu_tiles_pack_dev(c, width, height, m->world, m->rank, d_accum, m->d_send);
    if (rc != PTGPU_OK) return rc;
    RCCL_TRY(g_rccl.AllGather(m->d_send, m->d_recv, (size_t)sf, ncclFloat, m->comm, c->stream));
    return ptgpu_tiles_unpack_dev(c, width, height, m->world, m->rank, m->d_recv, d_accum);
}
