// libptgpu.so -- C ABI (include/ptgpu.h) over the gfx950 kernels.  Pure HIP: no torch types, no CPU fallback.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <vector>

#include "../../include/ptgpu.h"
#include "pt_context.h"
#include "pt_traverse.hip.h"
#include "pt_raycast_step.h"
#include "pt_frame_rays.hip.h"
#include "pt_integrator.hip.h"
#include "pt_integrator_pool.hip.h"
#include "pt_lbvh.hip.h"

namespace ptd {
__device__ const uint32_t g_sobol_matrices[1024 * 52] = {
#include "_gen/sobol_matrices.inc"
};
}

static_assert(sizeof(ptgpu_flatnode) == 64, "FlatNode must be 64 bytes like the reference's");

// errors: thread-local message behind ptgpu_last_error(), shared with the other translation units through pt_context.h
static thread_local char g_err[512] = "";
int ptgpu_vfail(int code, const char* fmt, va_list ap) {
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    return code;
}

// What the raycast kernels need beyond their inputs and outputs: counter words, the deferred-ray list of the one-ray-per-lane
// kernels and the global-memory stack overflow of the persistent ones.  Either the context's grow-only buffers (grow != NULL: sized
// on demand with hipMalloc) or a fixed region that is never resized (ptgpu_raycast_batch_dev_noalloc).
struct RcWorkspace {
    ptgpu_context* grow;            // non-NULL: the buffers below are the context's and may be reallocated larger
    unsigned long long* counters;   // CTR_WORDS words
    int32_t* defer;
    size_t defer_capacity;          // rays
    uint2* stack_ovf;
    size_t stack_ovf_bytes;
    long long max_persist_blocks;   // fixed region only: the persistent grid is clamped to what stack_ovf holds
};

struct ptgpu_scene {
    ptgpu_context* ctx = nullptr;
    ptd::TravScene trav{};
    int32_t num_tris = 0, num_inner = 0, depth = 0;
    int32_t num_nodes_built = 0;  // ptgpu_scene_create_lbvh: nodes of the tree the device built
    bool has_alpha = false;
    // device allocations
    float4* d_records = nullptr;  // triangle records in reverse order, then the interior records (TravScene::records points between them)
    float *d_pos = nullptr, *d_nrm = nullptr, *d_uv = nullptr;
    int32_t* d_mat_id = nullptr;
    ptgpu_material* d_materials = nullptr;
    ptd::DevTexture* d_textures = nullptr;
    std::vector<uint8_t*> d_texels;
    int32_t num_materials = 0, num_textures = 0;
    uint64_t bytes[4] = {0, 0, 0, 0};
    // scene_probe (before the scene's first path-trace pass): traversal steps per ray of random rays inside the scene's box, and the
    // integrator's service threshold chosen from it
    float probe_steps = -1.0f;
    int32_t pt_service = 32;
};

// zero a few counter words with plain stores (the form measured to be visible to a later kernel's atomics on this stream)
__global__ void k_zero_words(unsigned long long* p, int count) {
    for (int i = threadIdx.x; i < count; i += blockDim.x) p[i] = 0ull;
}
// the public counters (words 0..2) and the spread ray-count words
__global__ void k_reset_counters(unsigned long long* p) {
    for (int i = threadIdx.x; i < 3 + RAY_SPREAD; i += blockDim.x) p[i < 3 ? i : CTR_WORDS + (i - 3)] = 0ull;
}

// tab[(pair * 3 + k) * 256 + b] = XOR of the direction numbers of dimensions (2*pair, 2*pair + 1) selected by the bits of byte b at
// index bits 8k .. 8k+7: three gathers replace up to 24 iterations of sobol::sample's bit loop (XOR is associative: same word).
__global__ void k_build_sobol_pairs(uint2* tab) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ptd::SOBOL_PAIRS * 3 * 256) return;
    const int b = e & 255, k = (e >> 8) % 3, pair = (e >> 8) / 3;
    const uint32_t* row = ptd::g_sobol_matrices + (size_t)pair * 2 * 52 + 8 * k;
    uint32_t ra = 0u, rb = 0u;
    for (int j = 0; j < 8; ++j)
        if ((b >> j) & 1) { ra ^= row[j]; rb ^= row[j + 52]; }
    tab[e] = make_uint2(ra, rb);
}

extern "C" const char* ptgpu_last_error(void) { return g_err; }
static int noalloc_scratch_address(unsigned char** out);
extern "C" int ptgpu_shutdown(ptgpu_context* c);

static int init_context(ptgpu_context* c, const hipDeviceProp_t& prop) {
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipMalloc(&c->d_counters, (CTR_WORDS + RAY_SPREAD + 1) * sizeof(unsigned long long)));  // + the device copy of the cancel word
    HIP_TRY(hipMemset(c->d_counters, 0, (CTR_WORDS + RAY_SPREAD + 1) * sizeof(unsigned long long)));
    c->d_cancel = reinterpret_cast<uint32_t*>(c->d_counters + CTR_WORDS + RAY_SPREAD);
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int entries = ptd::SOBOL_PAIRS * 3 * 256;
    HIP_TRY(hipMalloc(&c->d_sobol_pairs, (size_t)entries * sizeof(uint2)));
    hipLaunchKernelGGL(k_build_sobol_pairs, dim3((entries + 255) / 256), dim3(256), 0, c->stream, c->d_sobol_pairs);
    // this device's address of the Sobol matrices, for kernels of other translation units (pt_raycast_step.hip: the frame kernel)
    {
        void* sym = nullptr;
        HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(ptd::g_sobol_matrices)));
        c->d_sobol_matrices = static_cast<const uint32_t*>(sym);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    {   // ptgpu_cancel's stream: highest priority, so that its one-wave kernel is not queued behind the pass it cancels
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->cancel_stream, hipStreamNonBlocking, hi));
    }
    return noalloc_scratch_address(&c->scratch);  // device `device` is current
}

extern "C" int ptgpu_init(int device, ptgpu_context** out_ctx) {
    if (!out_ctx) return fail(PTGPU_E_INVALID, "out_ctx is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PTGPU_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(PTGPU_E_INVALID, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PTGPU_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    ptgpu_context* c = new ptgpu_context();
    c->device = device;
    const int rc = init_context(c, prop);
    if (rc != PTGPU_OK) {  // free what was set up, keep the message of the call that failed
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        ptgpu_shutdown(c);
        memcpy(g_err, keep, sizeof keep);
        return rc;
    }
    *out_ctx = c;
    return PTGPU_OK;
}

extern "C" int ptgpu_shutdown(ptgpu_context* c) {
    if (!c) return PTGPU_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->d_counters) hipFree(c->d_counters);
    if (c->d_rad) hipFree(c->d_rad);
    if (c->d_lights) hipFree(c->d_lights);
    ptgpu_comm_release(c);
    if (c->d_spill) hipFree(c->d_spill);
    if (c->d_sort_ws) hipFree(c->d_sort_ws);
    if (c->d_foreign) hipFree(c->d_foreign);
    if (c->d_stack_ovf) hipFree(c->d_stack_ovf);
    if (c->d_defer) hipFree(c->d_defer);
    if (c->d_sobol_pairs) hipFree(c->d_sobol_pairs);
    if (c->cancel_stream) { hipStreamSynchronize(c->cancel_stream); hipStreamDestroy(c->cancel_stream); }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return PTGPU_OK;
}

extern "C" int ptgpu_set_stream(ptgpu_context* c, void* s) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PTGPU_OK;
}
extern "C" int ptgpu_sync(ptgpu_context* c) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}
extern "C" int ptgpu_query(ptgpu_context* c, int* out_done) {
    if (!c || !out_done) return fail(PTGPU_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = hipStreamQuery(c->stream);
    if (e != hipSuccess && e != hipErrorNotReady) return fail(PTGPU_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
    *out_done = (e == hipSuccess) ? 1 : 0;
    return PTGPU_OK;
}
// Cancels every render pass enqueued on the context so far (ptgpu_render_pass[_dev] calls that have returned or are running in another
// thread): a one-wave kernel on the context's cancel stream stores the number of the last enqueued pass into the device word the
// integrator's persistent waves read whenever they draw a new chunk of samples; they draw no more, batches that are not complete are
// not accumulated, and the stream drains within milliseconds.  Passes enqueued afterwards are not affected.  Does not wait.
__global__ void k_cancel_set(uint32_t* word, uint32_t epoch) {
    if (threadIdx.x == 0) __hip_atomic_store(word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int ptgpu_cancel(ptgpu_context* c) {
    if (!c || !c->d_cancel) return fail(PTGPU_E_INVALID, "ctx is NULL or not initialised");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_cancel_set, dim3(1), dim3(64), 0, c->cancel_stream, c->d_cancel, c->pass_epoch.load(std::memory_order_acquire));
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

extern "C" int ptgpu_malloc(ptgpu_context* c, uint64_t bytes, void** out) {
    if (!c || !out) return fail(PTGPU_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 4));
    return PTGPU_OK;
}
extern "C" int ptgpu_free(ptgpu_context* c, void* p) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (p) HIP_TRY(hipFree(p));
    return PTGPU_OK;
}
extern "C" int ptgpu_mem_info(ptgpu_context* c, uint64_t* out_free, uint64_t* out_total) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (out_free) *out_free = f;
    if (out_total) *out_total = t;
    return PTGPU_OK;
}
extern "C" int ptgpu_memcpy_h2d(ptgpu_context* c, void* dev, const void* host, uint64_t bytes) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    if (!bytes) return PTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}
extern "C" int ptgpu_memcpy_d2h(ptgpu_context* c, void* host, const void* dev, uint64_t bytes) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    if (!bytes) return PTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}
extern "C" int ptgpu_memset(ptgpu_context* c, void* dev, int value, uint64_t bytes) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    if (!bytes) return PTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(dev, value, bytes, c->stream));
    return PTGPU_OK;
}

extern "C" int ptgpu_enable_counters(ptgpu_context* c, int on) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    c->count = on != 0;
    return PTGPU_OK;
}
extern "C" int ptgpu_get_counters(ptgpu_context* c, ptgpu_counters* out) {
    if (!c || !out) return fail(PTGPU_E_INVALID, "NULL argument");
    static_assert(CTR_WORDS >= 3, "public counters first");
    std::vector<unsigned long long> h(CTR_WORDS + RAY_SPREAD);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(h.data(), c->d_counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->rays = h[0]; out->node_visits = h[1]; out->tri_tests = h[2];
    for (int i = 0; i < RAY_SPREAD; ++i) out->rays += h[CTR_WORDS + i];
    return PTGPU_OK;
}
#if defined(PT_STAMPS) || defined(PT_POOL_STATS)
// Diagnostic builds only (not declared in include/ptgpu.h, absent from the shipped library): raw copy of all counter words.
extern "C" int ptgpu_debug_counters(ptgpu_context* c, unsigned long long* out32) {
    if (!c || !out32) return fail(PTGPU_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out32, c->d_counters, CTR_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PTGPU_OK;
}
#endif
extern "C" int ptgpu_reset_counters(ptgpu_context* c) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(256), 0, c->stream, c->d_counters);  // spill / defer words are cleared per use
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// scene creation: validate the reference-layout tree, build the traversal layout, upload
// ------------------------------------------------------------------------------------------------
template <class T> static int upload(ptgpu_context* c, T** dptr, const T* h, size_t n, uint64_t* acc) {
    size_t bytes = n * sizeof(T);
    HIP_TRY(hipMalloc((void**)dptr, bytes ? bytes : 4));
    if (bytes) HIP_TRY(hipMemcpy(*dptr, h, bytes, hipMemcpyHostToDevice));
    if (acc) *acc += bytes;
    return PTGPU_OK;
}

static int child_ref(const ptgpu_flatnode* nodes, const std::vector<int32_t>& inner_index, int child) {
    return nodes[child].leaf ? ~nodes[child].startIdx : inner_index[child];
}

// Where interior record k (pre-order number) goes in the record array.  The order is this library's choice: only the references inside
// the records have to follow it.  PTGPU_NODE_ORDER (read at scene creation, an experiment knob; DESIGN.md section 4 has the measurements):
//   0  depth-first pre-order, the order of the flattened tree (the default)
//   1  breadth-first
//   2..6  treelets: the top D = 2..6 levels of a subtree breadth-first in one run, then the subtrees below it, recursively
//   9  sibling pairs: both interior children of a node in one aligned 128-byte line, subtrees depth-first
// Returns slot[k]; *n_slots >= n_inner (mode 9 leaves gaps).
static std::vector<int32_t> interior_slots(const ptgpu_flatnode* nodes, const std::vector<int32_t>& inner_index, int n_inner, int mode, int* n_slots) {
    std::vector<int32_t> slot(n_inner);
    *n_slots = n_inner;
    if (mode <= 0 || n_inner == 0) {
        for (int k = 0; k < n_inner; ++k) slot[k] = k;
        return slot;
    }
    auto interior_children = [&](int f, int out[2]) {
        int n = 0;
        const int kids[2] = {f + 1, nodes[f].rightChild};
        for (int k = 0; k < 2; ++k)
            if (!nodes[kids[k]].leaf) out[n++] = kids[k];
        return n;
    };
    int next = 0;
    if (mode == 1) {
        std::vector<int> queue{0};
        for (size_t h = 0; h < queue.size(); ++h) {
            const int f = queue[h];
            slot[inner_index[f]] = next++;
            int kids[2];
            const int n = interior_children(f, kids);
            for (int k = 0; k < n; ++k) queue.push_back(kids[k]);
        }
    } else if (mode == 9) {
        slot[inner_index[0]] = 0;
        next = 2;
        std::vector<int> stack{0};
        while (!stack.empty()) {
            const int f = stack.back();
            stack.pop_back();
            int kids[2];
            const int n = interior_children(f, kids);
            if (n == 2) next = (next + 1) & ~1;
            for (int k = 0; k < n; ++k) slot[inner_index[kids[k]]] = next++;
            for (int k = n - 1; k >= 0; --k) stack.push_back(kids[k]);
        }
        *n_slots = next;
    } else {
        const int levels = mode;
        std::vector<int> roots{0}, level, below;
        while (!roots.empty()) {
            const int r = roots.back();
            roots.pop_back();
            level.assign(1, r);
            for (int l = 0; l < levels && !level.empty(); ++l) {
                below.clear();
                for (int f : level) {
                    slot[inner_index[f]] = next++;
                    int kids[2];
                    const int n = interior_children(f, kids);
                    for (int k = 0; k < n; ++k) below.push_back(kids[k]);
                }
                level.swap(below);
            }
            for (size_t k = level.size(); k-- > 0;) roots.push_back(level[k]);  // the subtrees below this treelet, leftmost next
        }
    }
    return slot;
}

// triangle / material / texture arrays of a scene description (everything but the hierarchy)
static int check_scene_arrays(const ptgpu_scene_desc* d) {
    const int32_t nn = d->num_nodes, nl = d->list_size, nt = d->num_tris;
    if (nt < 0 || nn < 0 || nl < 0) return fail(PTGPU_E_INVALID, "negative count in scene description");
    if (nt > 0 && (!d->pos || !d->nrm || !d->uv || !d->mat_id || !d->woop))
        return fail(PTGPU_E_INVALID, "triangle arrays missing");
    if (d->num_materials < 0 || (d->num_materials > 0 && !d->materials)) return fail(PTGPU_E_INVALID, "materials missing");
    if (d->num_textures < 0 || (d->num_textures > 0 && !d->textures)) return fail(PTGPU_E_INVALID, "textures missing");
    for (int i = 0; i < nt; ++i)
        if (d->mat_id[i] < 0 || d->mat_id[i] >= d->num_materials)
            return fail(PTGPU_E_INVALID, "triangle %d: material index %d out of range", i, d->mat_id[i]);
    for (int i = 0; i < d->num_materials; ++i) {
        const int32_t t[4] = {d->materials[i].tex_diffuse, d->materials[i].tex_alpha, d->materials[i].tex_normal, d->materials[i].tex_specular};
        for (int k = 0; k < 4; ++k)
            if (t[k] < -1 || t[k] >= d->num_textures) return fail(PTGPU_E_INVALID, "material %d: texture index %d out of range", i, t[k]);
    }
    for (int i = 0; i < d->num_textures; ++i)
        if (d->textures[i].width < 2 || d->textures[i].height < 2 || !d->textures[i].rgba)
            return fail(PTGPU_E_INVALID, "texture %d: needs width,height >= 2 and pixel data", i);
    return PTGPU_OK;
}

// the arrays shading reads (never touched by traversal)
static int upload_shading_arrays(ptgpu_context* c, const ptgpu_scene_desc* d, ptgpu_scene* s) {
    const int32_t nt = d->num_tris;
    int rc;
    if ((rc = upload(c, &s->d_pos, d->pos, (size_t)nt * 9, &s->bytes[2]))) return rc;
    if ((rc = upload(c, &s->d_nrm, d->nrm, (size_t)nt * 9, &s->bytes[2]))) return rc;
    if ((rc = upload(c, &s->d_uv, d->uv, (size_t)nt * 6, &s->bytes[2]))) return rc;
    if ((rc = upload(c, &s->d_mat_id, d->mat_id, (size_t)nt, &s->bytes[2]))) return rc;
    if ((rc = upload(c, &s->d_materials, d->materials, (size_t)d->num_materials, &s->bytes[2]))) return rc;
    std::vector<ptd::DevTexture> tex(d->num_textures);
    for (int i = 0; i < d->num_textures; ++i) {
        uint8_t* p = nullptr;
        if ((rc = upload(c, &p, d->textures[i].rgba, (size_t)d->textures[i].width * d->textures[i].height * 4, &s->bytes[3]))) return rc;
        s->d_texels.push_back(p);
        tex[i].rgba = p; tex[i].width = d->textures[i].width; tex[i].height = d->textures[i].height;
    }
    if ((rc = upload(c, &s->d_textures, tex.data(), tex.size(), &s->bytes[3]))) return rc;
    s->trav.uv = s->d_uv;
    s->trav.textures = s->d_textures;
    return PTGPU_OK;
}

static int build_scene(ptgpu_context* c, const ptgpu_scene_desc* d, ptgpu_scene* s) {
    const int32_t nn = d->num_nodes, nl = d->list_size, nt = d->num_tris;
    {
        const int rc0 = check_scene_arrays(d);
        if (rc0 != PTGPU_OK) return rc0;
    }

    s->num_tris = nt;
    s->num_materials = d->num_materials;
    s->num_textures = d->num_textures;
    std::vector<float4> inner, tris;
    int depth = 0;
    const char* order_env = getenv("PTGPU_NODE_ORDER");
    const int node_order = order_env ? atoi(order_env) : 0;
    if (nn > 0) {
        if (!d->nodes || !d->list || nl <= 0 || nt <= 0) return fail(PTGPU_E_INVALID, "BVH given without nodes/list/triangles");
        for (int i = 0; i < nl; ++i)
            if (d->list[i] < 0 || d->list[i] >= nt) return fail(PTGPU_E_INVALID, "list[%d] = %d out of range", i, d->list[i]);
        // Walk the tree exactly as the flattened layout promises (left child = index + 1, Bvh.cpp:85) and check that it is a
        // tree over [0, num_nodes) whose leaves tile the list in order; anything else could make a kernel loop or fault.
        std::vector<int32_t> inner_index(nn, -1);
        std::vector<uint8_t> seen(nn, 0);
        struct Item { int node, depth; };
        std::vector<Item> stack;
        stack.push_back({0, 1});
        int next_list = 0, n_inner = 0;
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            if (it.node < 0 || it.node >= nn) return fail(PTGPU_E_INVALID, "child index %d out of range", it.node);
            if (seen[it.node]) return fail(PTGPU_E_INVALID, "node %d reachable twice (not a tree)", it.node);
            seen[it.node] = 1;
            const ptgpu_flatnode& n = d->nodes[it.node];
            if (it.depth > depth) depth = it.depth;
            for (int k = 0; k < 3; ++k)  // the NaN-free slab test of the kernels relies on this
                if (n.bmin[k] != n.bmin[k] || n.bmax[k] != n.bmax[k]) return fail(PTGPU_E_INVALID, "node %d: NaN in its bounding box", it.node);
            if (n.leaf) {
                if (n.startIdx != next_list || n.endIdx < n.startIdx || n.endIdx >= nl)
                    return fail(PTGPU_E_INVALID, "leaf %d: range [%d,%d] does not continue the list at %d", it.node, n.startIdx, n.endIdx, next_list);
                next_list = n.endIdx + 1;
            } else {
                if (n.axis < 0 || n.axis > 2) return fail(PTGPU_E_INVALID, "node %d: axis %d", it.node, n.axis);
                if (n.rightChild <= it.node + 1 || n.rightChild >= nn) return fail(PTGPU_E_INVALID, "node %d: rightChild %d", it.node, n.rightChild);
                inner_index[it.node] = n_inner++;
                stack.push_back({n.rightChild, it.depth + 1});
                stack.push_back({it.node + 1, it.depth + 1});  // popped first: pre-order
            }
        }
        if (next_list != nl) return fail(PTGPU_E_INVALID, "leaves cover %d of %d list entries", next_list, nl);
        if (depth > 64) return fail(PTGPU_E_UNSUPPORTED, "tree depth %d exceeds 64 (the reference's traversal stack, RayTracer.cpp:190)", depth);

        int n_slots = n_inner;
        const std::vector<int32_t> slot = interior_slots(d->nodes, inner_index, n_inner, node_order, &n_slots);
        inner.assign((size_t)n_slots * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        for (int i = 0; i < nn; ++i) {
            if (inner_index[i] < 0) continue;
            const ptgpu_flatnode& n = d->nodes[i];
            const ptgpu_flatnode& L = d->nodes[i + 1];
            const ptgpu_flatnode& R = d->nodes[n.rightChild];
            float4* o = &inner[(size_t)slot[inner_index[i]] * 4];
            // x/y planes pairwise, z planes together: the fast slab test runs on packed fp32 pairs (pt_traverse.hip.h)
            o[0] = make_float4(L.bmin[0], L.bmin[1], L.bmax[0], L.bmax[1]);
            o[1] = make_float4(R.bmin[0], R.bmin[1], R.bmax[0], R.bmax[1]);
            o[2] = make_float4(L.bmin[2], L.bmax[2], R.bmin[2], R.bmax[2]);
            int refL = child_ref(d->nodes, inner_index, i + 1), refR = child_ref(d->nodes, inner_index, n.rightChild);
            if (refL >= 0) refL = slot[refL];
            if (refR >= 0) refR = slot[refR];
            float fl, fr, fa;
            int32_t ax = n.axis;
            memcpy(&fl, &refL, 4); memcpy(&fr, &refR, 4); memcpy(&fa, &ax, 4);
            o[3] = make_float4(fl, fr, fa, 0.0f);
        }
        tris.resize((size_t)nl * 4);
        for (int i = 0; i < nn; ++i) {
            const ptgpu_flatnode& n = d->nodes[i];
            if (!n.leaf) continue;
            for (int p = n.startIdx; p <= n.endIdx; ++p) {
                int32_t id = d->list[p], remaining = n.endIdx - p, mat = d->mat_id[id];
                int32_t atex = d->materials[mat].tex_alpha;
                if (atex >= 0) s->has_alpha = true;
                const float* w = d->woop + (size_t)id * 12;
                float4* o = &tris[(size_t)p * 4];
                o[0] = make_float4(w[0], w[1], w[2], w[3]);
                o[1] = make_float4(w[4], w[5], w[6], w[7]);
                o[2] = make_float4(w[8], w[9], w[10], w[11]);
                float f0, f1, f2, f3;
                memcpy(&f0, &id, 4); memcpy(&f1, &remaining, 4); memcpy(&f2, &atex, 4); memcpy(&f3, &mat, 4);
                o[3] = make_float4(f0, f1, f2, f3);
            }
        }
        const ptgpu_flatnode& root = d->nodes[0];
        for (int k = 0; k < 3; ++k) { s->trav.root_bmin[k] = root.bmin[k]; s->trav.root_bmax[k] = root.bmax[k]; }
        s->trav.root_ref = root.leaf ? ~root.startIdx : 0;
        s->trav.list_size = nl;
        s->num_inner = n_inner;
    } else {
        s->trav.list_size = 0;  // empty scene: every ray misses
        s->trav.root_ref = -1;
    }
    s->depth = depth;

    int rc;
    {
        // one array: triangle record p at index (nl - 1 - p), interior record i at index nl + i  (records = base + nl records)
        const size_t n_tri_rec = tris.size() / 4, n_in_rec = inner.size() / 4;
        const size_t pad = node_order > 0 ? (n_tri_rec & 1) : 0;  // experimental orders: interior record 0 on a 128-byte line boundary
        std::vector<float4> all((pad + n_tri_rec + n_in_rec) * 4);
        for (size_t p = 0; p < n_tri_rec; ++p) memcpy(&all[(pad + n_tri_rec - 1 - p) * 4], &tris[p * 4], 64);
        if (n_in_rec) memcpy(&all[(pad + n_tri_rec) * 4], inner.data(), n_in_rec * 64);
        if ((rc = upload(c, &s->d_records, all.data(), all.size(), nullptr))) return rc;
        s->bytes[0] += inner.size() * sizeof(float4);
        s->bytes[1] += tris.size() * sizeof(float4);
        s->trav.records = s->d_records + (pad + n_tri_rec) * 4;
    }
    return upload_shading_arrays(c, d, s);
}

extern "C" int ptgpu_scene_destroy(ptgpu_scene* s) {
    if (!s) return PTGPU_OK;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    void* ptrs[] = {s->d_records, s->d_pos, s->d_nrm, s->d_uv, s->d_mat_id, s->d_materials, s->d_textures};
    for (void* p : ptrs) if (p) hipFree(p);
    for (uint8_t* p : s->d_texels) if (p) hipFree(p);
    delete s;
    return PTGPU_OK;
}

extern "C" int ptgpu_scene_create(ptgpu_context* c, const ptgpu_scene_desc* d, ptgpu_scene** out) {
    if (!c || !d || !out) return fail(PTGPU_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    ptgpu_scene* s = new ptgpu_scene();
    s->ctx = c;
    int rc = build_scene(c, d, s);
    if (rc != PTGPU_OK) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        ptgpu_scene_destroy(s);
        memcpy(g_err, keep, sizeof keep);
        return rc;
    }
    *out = s;
    return PTGPU_OK;
}

extern "C" int ptgpu_scene_stats(const ptgpu_scene* s, uint64_t out_bytes[4], int32_t* out_depth) {
    if (!s) return fail(PTGPU_E_INVALID, "scene is NULL");
    if (out_bytes) for (int i = 0; i < 4; ++i) out_bytes[i] = s->bytes[i];
    if (out_depth) *out_depth = s->depth;
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
constexpr int RC_BLOCK = 256;      // persistent raycast kernel: 4 long-lived waves per workgroup
constexpr int RC_STEP_BLOCK = ptd::RC_STEP_BLOCK;  // one-ray-per-lane kernels: pt_raycast_step.hip, a translation unit of its own

// Long-lived waves with refill: wave w of W owns one chunk of RC_CHUNK consecutive rays out of every W consecutive chunks and hands
// fresh rays to its idle lanes in batches (ballot + prefix count), so lanes do not idle while the wave's slowest ray
// finishes.  The chunk stride is static: pulling chunks from a shared cursor (one word, or eight sharded ones) was
// measured 1.8-2.2x slower on the configs[1] frame.  The grid is a small multiple of what is resident at once, so the
// dispatcher still evens out cheap and expensive image regions.  Which lane traces which ray never affects a result.
// Exit: the wave's chunk index has passed the last chunk and no lane holds a ray.
constexpr int RC_CHUNK = 64;
template <bool ANYHIT, bool ALPHA, bool COUNT, int NLDS, bool OVF>
__global__ __launch_bounds__(RC_BLOCK) void k_raycast_persist(ptd::TravScene S, const float* __restrict__ orig, const float* __restrict__ dir,
                                                              long long n, int32_t* __restrict__ out_id, float* __restrict__ out_t,
                                                              float* __restrict__ out_u, float* __restrict__ out_v,
                                                              unsigned long long* counters, int refill_threshold, uint2* stack_ovf,
                                                              const int32_t* __restrict__ ray_index) {
    // ray_index != NULL: trace rays ray_index[0 .. counters[5]) (the rays k_raycast_step deferred) instead of 0 .. n
    if (ray_index) {
        const unsigned long long m = counters[5];  // written by an earlier kernel on this stream
        n = (long long)(m < (unsigned long long)n ? m : (unsigned long long)n);
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&counters[0], (unsigned long long)n);  // 1 ray = 1 raycast() call (reference src/RayTracer.cpp:192); deferred rays were counted
    }
    __shared__ uint2 lds_stack[RC_BLOCK / 64][NLDS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long total_waves = (long long)gridDim.x * (RC_BLOCK / 64);
    const long long wave_id = (long long)blockIdx.x * (RC_BLOCK / 64) + wave;
    ptd::Trav<ALPHA, COUNT, NLDS, OVF> tr;
    tr.reset_counts();
    tr.attach(&lds_stack[wave][0][lane], OVF ? stack_ovf + (size_t)wave_id * (64 - NLDS) * 64 + lane : nullptr);
    tr.alive = false;
    long long ray = -1;           // ray this lane currently owns
    long long next = 0, end = 0;  // wave-uniform: unassigned part of the wave's current chunk
    long long round = -1;  // advanced before first use
    bool exhausted = false;       // wave-uniform
    for (;;) {
        // Retire finished rays and hand out new ones in batches: doing it the moment any single lane finishes would add
        // two dependent memory round trips (hit record out, ray in) to nearly every step of the whole wave.
        const unsigned long long running = __ballot(tr.alive);
        const int idle = 64 - __popcll(running);
        if (idle >= refill_threshold || running == 0ull) {
            if (ray >= 0 && !tr.alive) {
                ptd::HitRec h = tr.result(S);
                if (out_id) out_id[ray] = h.id;
                if (out_t) out_t[ray] = h.t;
                if (out_u) out_u[ray] = h.u;
                if (out_v) out_v[ray] = h.v;
                ray = -1;
            }
            unsigned long long need = __ballot(ray < 0);
            while (need != 0ull && !exhausted) {
                if (next >= end) {
                    // round r hands this wave chunk r*W + (w + 997 r) mod W: rotating the assignment every round keeps one wave
                    // from drawing the same image columns (all cheap or all expensive) in every round
                    round += 1;
                    const long long base = (round * total_waves + (wave_id + round * 997) % total_waves) * RC_CHUNK;
                    if (base >= n) { exhausted = true; break; }
                    next = base;
                    end = next + RC_CHUNK < n ? next + RC_CHUNK : n;
                }
                int avail = (int)(end - next);
                int rank = __popcll(need & ((1ull << lane) - 1ull));  // position of this lane among the lanes that need a ray
                if ((ray < 0) && (rank < avail)) {
                    ray = ray_index ? (long long)ray_index[next + rank] : next + rank;
                    tr.begin(S, ptd::ld3(orig + ray * 3), ptd::ld3(dir + ray * 3), ANYHIT);
                }
                int taken = __popcll(need) < avail ? __popcll(need) : avail;
                next += taken;
                need = __ballot(ray < 0);
            }
        }
        if (__ballot(ray >= 0) == 0ull) break;  // nothing left in this wave (only reachable once exhausted)
        if (tr.alive) tr.step(S);
    }
    if (COUNT) {
        atomicAdd(&counters[1], (unsigned long long)tr.n_nodes);
        atomicAdd(&counters[2], (unsigned long long)tr.n_tris);
    }
}

// Kernel selection.  PTGPU_RAYCAST_IMPL (1 = step-wise, one ray per lane; 2 = persistent step-wise with refill) and
// PTGPU_RAYCAST_NLDS (16 / 24 LDS stack entries per lane) exist for A/B timing in one process (read at every launch).  Every
// variant produces identical hit records.  For trees deeper than the LDS stack the one-ray-per-lane kernel defers the (rare) rays that
// actually overflow to a persistent launch, whose bounded set of waves owns a global-memory overflow region each.
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int ensure_stack_overflow(ptgpu_context* c, size_t bytes) {
    if (bytes <= c->stack_ovf_bytes) return PTGPU_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_stack_ovf) { HIP_TRY(hipFree(c->d_stack_ovf)); c->d_stack_ovf = nullptr; c->stack_ovf_bytes = 0; }
    HIP_TRY(hipMalloc(&c->d_stack_ovf, bytes));
    c->stack_ovf_bytes = bytes;
    return PTGPU_OK;
}
// device staging for the entry points that take HOST pointers: grown when a call needs more, never shrunk, freed at shutdown
static int ensure_stage(ptgpu_context* c, size_t bytes) {
    if (bytes <= c->stage_bytes) return PTGPU_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_stage) { HIP_TRY(hipFree(c->d_stage)); c->d_stage = nullptr; c->stage_bytes = 0; }
    HIP_TRY(hipMalloc(&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return PTGPU_OK;
}
static RcWorkspace context_workspace(ptgpu_context* c) {
    return RcWorkspace{c, c->d_counters, c->d_defer, c->defer_capacity, c->d_stack_ovf, c->stack_ovf_bytes, 0};
}
static int ws_stack_overflow(RcWorkspace& ws, size_t bytes) {
    if (ws.grow) {
        int rc = ensure_stack_overflow(ws.grow, bytes);
        ws.stack_ovf = ws.grow->d_stack_ovf;
        ws.stack_ovf_bytes = ws.grow->stack_ovf_bytes;
        return rc;
    }
    return bytes <= ws.stack_ovf_bytes ? PTGPU_OK : fail(PTGPU_E_UNSUPPORTED, "fixed workspace: %zu bytes of stack overflow needed, %zu available", bytes, ws.stack_ovf_bytes);
}
static int ws_defer_list(RcWorkspace& ws, size_t n) {
    if (n <= ws.defer_capacity) return PTGPU_OK;
    if (!ws.grow) return fail(PTGPU_E_UNSUPPORTED, "fixed workspace: deferred-ray list holds %zu rays, %zu needed", ws.defer_capacity, n);
    ptgpu_context* c = ws.grow;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_defer) { HIP_TRY(hipFree(c->d_defer)); c->d_defer = nullptr; c->defer_capacity = 0; }
    HIP_TRY(hipMalloc(&c->d_defer, n * sizeof(int32_t)));
    c->defer_capacity = n;
    ws.defer = c->d_defer;
    ws.defer_capacity = n;
    return PTGPU_OK;
}

template <bool ANYHIT, bool ALPHA, bool COUNT, int NLDS, bool OVF>
static int launch_rc_persist(ptgpu_scene* s, RcWorkspace& ws, const ptd::TravScene& trav, const float* o, const float* d, long long n, int32_t* id, float* t, float* u, float* v,
                             const int32_t* ray_index = nullptr, long long expected = -1, int refill_default = 32) {
    ptgpu_context* c = s->ctx;
    long long want = ((expected >= 0 ? expected : n) + RC_CHUNK - 1) / RC_CHUNK;  // no more waves than chunks
    if (want < 1) want = 1;
    long long blocks = (want + (RC_BLOCK / 64) - 1) / (RC_BLOCK / 64);
    // a small multiple of the blocks resident at once (LDS-limited): long-lived waves, dispatcher-level balancing
    static std::atomic<int> per_cu_cache{0};  // same value whichever thread computes it first
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_raycast_persist<ANYHIT, ALPHA, COUNT, NLDS, OVF>, RC_BLOCK, 0) != hipSuccess || v < 1) v = 1;
        per_cu = v;
        per_cu_cache.store(v, std::memory_order_relaxed);
    }
    long long resident = (long long)c->num_cu * per_cu * env_int("PTGPU_RAYCAST_OVERSUB", 1);
    if (blocks > resident) blocks = resident;
    if (OVF) {
        // fixed workspace: the grid is clamped to the waves its overflow region holds (variants without overflow stacks need no clamp)
        if (!ws.grow && blocks > ws.max_persist_blocks) blocks = ws.max_persist_blocks;
        int rc = ws_stack_overflow(ws, (size_t)blocks * (RC_BLOCK / 64) * ptd::stack_overflow_bytes_per_wave<NLDS>());
        if (rc != PTGPU_OK) return rc;
    }
    hipLaunchKernelGGL((k_raycast_persist<ANYHIT, ALPHA, COUNT, NLDS, OVF>), dim3((unsigned)blocks), dim3(RC_BLOCK), 0, c->stream, trav, o, d, n, id, t,
                       u, v, ws.counters, env_int("PTGPU_RAYCAST_REFILL", refill_default), ws.stack_ovf, ray_index);
    return PTGPU_OK;
}

template <bool ANYHIT, bool ALPHA, bool COUNT>
static int launch_raycast(ptgpu_scene* s, RcWorkspace& ws, const float* o, const float* d, long long n, int32_t* id, float* t, float* u, float* v) {
    ptgpu_context* c = s->ctx;
    ptd::TravScene trav = s->trav;
    trav.alpha_on = ALPHA ? 1 : 0;
    // stack occupancy <= depth - 1 (one far sibling per ancestor level); the counting variant pushes every far child as well
    const int need = s->depth - 1;
    // LDS stack entries per lane: fewer entries = more resident waves (64 lanes x 8 B per entry).  A lane whose stack outgrows them
    // either continues in a global-memory region of its workgroup (growable workspace, RC_BEYOND_OVERFLOW) or is deferred to a second,
    // persistent launch (fixed workspace, RC_BEYOND_DEFER).  Measured closest / any-hit frames, ms (round 3, overflow regions):
    // head (1.6 MB, depth 22, coherent camera rays) 24: 0.136 / 0.174, 16: 0.146 / 0.164, 12: 0.144 / 0.166, 8: 0.144 / 0.168 (its
    // random rays 0.274 / 0.238 / 0.218 / 0.208); 262k-triangle atrium (33 MB, depth 24) 8: 1.05 / 1.52, 12: 1.07 / 1.57, 16: 1.20 / 1.80,
    // 24: 1.44 / 2.17; clouds of 50 k triangles 8: 0.53 / 0.63, 24: 0.66 / 0.80; 300 k 8: 0.95 / 0.92, 12: 0.98 / 0.96, 16: 1.08 / 1.05.
    // With deferral (what the fixed workspace still does) 12 entries were best up to depth 25, 16 beyond (8 defers too many rays).
    const bool small_scene = s->bytes[0] + s->bytes[1] <= (4u << 20) && need <= 24;
    // Scenes far beyond the caches (records > 64 MB): long-lived waves with refill at 16 idle lanes and a SHORT LDS stack (12 entries,
    // the rest in global memory: 24 waves per CU instead of 20) -- on such scenes a wave of 64 fresh rays waits for its longest one at
    // 18 % lane utilisation.  Measured, ms closest / any-hit over a 1080p frame, one ray per lane -> this: tetrahedron clouds of 1 M
    // triangles 1.48 / 1.32 -> 1.30 / 0.95, 3 M 1.77 / 1.63 -> 1.45 / 1.12, 10 M 2.09 / 1.95 -> 1.80 / 1.29; it loses below that size
    // (300 k cloud 1.04 -> 1.13 closest; 262 k-triangle atrium 1.09 / 1.67 -> 1.74 / 1.77), with 8 or 10 LDS entries (overflow
    // traffic), with 16 (fewer waves) and with a grid of 2x / 4x the resident waves.  profiles/r03_persist_short_stack.txt.
    // A fixed workspace (the entry point that allocates nothing) keeps one ray per lane: its overflow region holds 64 workgroups' stacks.
    const bool large_scene = s->bytes[0] + s->bytes[1] > (64u << 20) && ws.grow != nullptr;
    const int impl = env_int("PTGPU_RAYCAST_IMPL", large_scene ? 2 : 1);
    const size_t groups = (size_t)((n + RC_STEP_BLOCK - 1) / RC_STEP_BLOCK);
    auto region_bytes = [&](int entries) { return need > entries ? groups * (size_t)(need - entries) * 64 * sizeof(uint2) : (size_t)0; };
    const size_t region_cap = (size_t)1 << 30;  // of the context's grow-only overflow buffer
    const bool regions = ws.grow != nullptr && env_int("PTGPU_RAYCAST_DEFER", 0) == 0;
    int nlds_default;
    if (large_scene && impl == 2) nlds_default = 12;
    else if (small_scene) nlds_default = (ANYHIT && regions && region_bytes(16) <= region_cap) ? 16 : 24;
    else if (regions && region_bytes(8) <= region_cap) nlds_default = 8;
    else nlds_default = need <= 24 ? 12 : 16;
    const int nlds = env_int("PTGPU_RAYCAST_NLDS", nlds_default);
    if (nlds != 8 && nlds != 12 && nlds != 16 && nlds != 24 && !(nlds == 10 && impl == 2))  // the instantiated set: `defer` below must describe the kernel that runs
        return fail(PTGPU_E_INVALID, "PTGPU_RAYCAST_NLDS=%d: the LDS stack is built with 8, 12, 16 or 24 entries", nlds);
    const int refill = large_scene ? 16 : 32;
    if constexpr (COUNT || ALPHA) {  // rare variants: one configuration that is valid for every depth
        return launch_rc_persist<ANYHIT, ALPHA, COUNT, 16, true>(s, ws, trav, o, d, n, id, t, u, v);
    } else {
        if (impl == 2) {
            // short LDS stacks with global-memory overflow: more resident waves (12: the default of scenes beyond the caches; 8 / 10: A/B only)
            if (nlds == 8) return launch_rc_persist<ANYHIT, ALPHA, COUNT, 8, true>(s, ws, trav, o, d, n, id, t, u, v, nullptr, -1, refill);
            if (nlds == 10) return launch_rc_persist<ANYHIT, ALPHA, COUNT, 10, true>(s, ws, trav, o, d, n, id, t, u, v, nullptr, -1, refill);
            if (nlds == 12) return launch_rc_persist<ANYHIT, ALPHA, COUNT, 12, true>(s, ws, trav, o, d, n, id, t, u, v, nullptr, -1, refill);
            if (need > 24 || (need > 16 && nlds == 16)) return launch_rc_persist<ANYHIT, ALPHA, COUNT, 16, true>(s, ws, trav, o, d, n, id, t, u, v, nullptr, -1, refill);
            return nlds == 16 ? launch_rc_persist<ANYHIT, ALPHA, COUNT, 16, false>(s, ws, trav, o, d, n, id, t, u, v, nullptr, -1, refill)
                              : launch_rc_persist<ANYHIT, ALPHA, COUNT, 24, false>(s, ws, trav, o, d, n, id, t, u, v, nullptr, -1, refill);
        }
        // one ray per lane.  If the tree could need more stack than the LDS holds: with a growable workspace every workgroup of the grid
        // gets a global-memory region for its deeper entries (need - nlds entries x 512 B per 64 rays; at most 1 GiB, else as below);
        // with the fixed workspace of the entry point that allocates nothing the rays that actually overflow are deferred to a second,
        // persistent launch whose bounded set of waves owns overflow regions (they are rare: occupancy stays < 16 for almost every ray).
        int beyond = need > nlds ? ptd::RC_BEYOND_DEFER : ptd::RC_BEYOND_NEVER;
        const size_t ovf_bytes = region_bytes(nlds);
        if (beyond == ptd::RC_BEYOND_DEFER && regions && ovf_bytes <= region_cap) {
            const int rc = ws_stack_overflow(ws, ovf_bytes);
            if (rc != PTGPU_OK) return rc;
            beyond = ptd::RC_BEYOND_OVERFLOW;
        }
        const bool defer = beyond == ptd::RC_BEYOND_DEFER;
        if (defer) {
            if (n > 0x7fffffffLL) return fail(PTGPU_E_INVALID, "n too large for the deferred-ray list");
            const int rc = ws_defer_list(ws, (size_t)n);
            if (rc != PTGPU_OK) return rc;
            hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, c->stream, ws.counters + 5, 1);
        }
        ptd::launch_raycast_step(ANYHIT, nlds, beyond, c->stream, trav, o, d, n, id, t, u, v, ws.counters, ws.defer, ws.stack_ovf, need - nlds);
        if (defer) {
            // the number of deferred rays is only known on the device: a small fixed grid, the kernel reads the count itself
            const long long expected = n / 64 < 65536 ? 65536 : n / 64;
            return launch_rc_persist<ANYHIT, ALPHA, COUNT, 16, true>(s, ws, trav, o, d, n, id, t, u, v, ws.defer, expected < n ? expected : n);
        }
        return PTGPU_OK;
    }
}

static int scene_probe(ptgpu_scene* s) {
    if (s->probe_steps >= 0.0f) return PTGPU_OK;
    ptgpu_context* c = s->ctx;
    s->probe_steps = 0.0f;
    s->pt_service = 32;
    if (s->trav.list_size <= 0) return PTGPU_OK;
    constexpr int N = 8192;
    std::vector<float> h((size_t)N * 6);
    uint32_t x = 0x9E3779B9u;
    auto rnd = [&x]() { x = x * 1664525u + 1013904223u; return (float)(x >> 8) * (1.0f / 16777216.0f); };
    float ext[3], diag2 = 0.0f;
    for (int k = 0; k < 3; ++k) { ext[k] = s->trav.root_bmax[k] - s->trav.root_bmin[k]; diag2 += ext[k] * ext[k]; }
    const float diag = sqrtf(diag2);
    for (int i = 0; i < N; ++i) {
        float d[3], l2;
        do {
            l2 = 0.0f;
            for (int k = 0; k < 3; ++k) { d[k] = 2.0f * rnd() - 1.0f; l2 += d[k] * d[k]; }
        } while (l2 > 1.0f || l2 < 1e-4f);
        const float sc = diag / sqrtf(l2);
        for (int k = 0; k < 3; ++k) {
            h[(size_t)i * 3 + k] = s->trav.root_bmin[k] + rnd() * ext[k];
            h[(size_t)N * 3 + (size_t)i * 3 + k] = d[k] * sc;
        }
    }
    unsigned char* d_tmp = nullptr;
    const size_t ray_bytes = h.size() * sizeof(float), ctr_bytes = (size_t)(CTR_WORDS + RAY_SPREAD) * sizeof(unsigned long long);
    HIP_TRY(hipMalloc(&d_tmp, ray_bytes + ctr_bytes));
    float* d_rays = reinterpret_cast<float*>(d_tmp);
    unsigned long long* d_ctr = reinterpret_cast<unsigned long long*>(d_tmp + ray_bytes);
    int rc = PTGPU_OK;
    unsigned long long hc[3] = {0, 0, 0};
    hipError_t e = hipMemcpyAsync(d_rays, h.data(), ray_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_ctr, 0, ctr_bytes, c->stream);
    if (e == hipSuccess) {
        RcWorkspace ws{c, d_ctr, c->d_defer, c->defer_capacity, c->d_stack_ovf, c->stack_ovf_bytes, 0};
        rc = launch_raycast<false, false, true>(s, ws, d_rays, d_rays + (size_t)N * 3, N, nullptr, nullptr, nullptr, nullptr);
        if (rc == PTGPU_OK) e = hipMemcpyAsync(hc, d_ctr, sizeof hc, hipMemcpyDeviceToHost, c->stream);
    }
    const hipError_t e2 = hipStreamSynchronize(c->stream);  // also before the buffer is freed
    hipFree(d_tmp);
    if (rc != PTGPU_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(PTGPU_E_HIP, "scene probe failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    s->probe_steps = (float)((double)(hc[1] + hc[2]) / N);
    // 32 up to 20 steps per ray, 12 from 55 on, linear in between, in steps of 4
    float t = 32.0f - (s->probe_steps - 20.0f) * (20.0f / 35.0f);
    t = t > 32.0f ? 32.0f : (t < 12.0f ? 12.0f : t);
    s->pt_service = ((int)(t + 2.0f) / 4) * 4;
    return PTGPU_OK;
}

extern "C" int ptgpu_scene_ray_length(ptgpu_scene* s, float* out_steps_per_ray, int32_t* out_service_threshold) {
    if (!s) return fail(PTGPU_E_INVALID, "scene is NULL");
    HIP_TRY(hipSetDevice(s->ctx->device));
    const int rc = scene_probe(s);
    if (rc != PTGPU_OK) return rc;
    if (out_steps_per_ray) *out_steps_per_ray = s->probe_steps;
    if (out_service_threshold) *out_service_threshold = s->pt_service;
    return PTGPU_OK;
}

static int raycast_dispatch(ptgpu_scene* s, RcWorkspace& ws, bool alpha, bool count, const float* d_orig, const float* d_dir, int64_t n, int occlusion,
                            int32_t* d_id, float* d_t, float* d_u, float* d_v) {
    int rc;
    const int sel = (occlusion ? 4 : 0) | (alpha ? 2 : 0) | (count ? 1 : 0);
    switch (sel) {
        case 0: rc = launch_raycast<false, false, false>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        case 1: rc = launch_raycast<false, false, true>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        case 2: rc = launch_raycast<false, true, false>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        case 3: rc = launch_raycast<false, true, true>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        case 4: rc = launch_raycast<true, false, false>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        case 5: rc = launch_raycast<true, false, true>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        case 6: rc = launch_raycast<true, true, false>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
        default: rc = launch_raycast<true, true, true>(s, ws, d_orig, d_dir, n, d_id, d_t, d_u, d_v); break;
    }
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

extern "C" int ptgpu_raycast_batch_dev(ptgpu_scene* s, int use_alpha, const float* d_orig, const float* d_dir, int64_t n, int occlusion,
                                       int32_t* d_id, float* d_t, float* d_u, float* d_v) {
    if (!s) return fail(PTGPU_E_INVALID, "scene is NULL");
    if (n < 0) return fail(PTGPU_E_INVALID, "n < 0");
    if (n == 0) return PTGPU_OK;
    if (!d_orig || !d_dir) return fail(PTGPU_E_INVALID, "ray arrays are NULL");
    if ((n + RC_STEP_BLOCK - 1) / RC_STEP_BLOCK > 0x7fffffffLL) return fail(PTGPU_E_INVALID, "n too large for one launch");
    HIP_TRY(hipSetDevice(s->ctx->device));
    RcWorkspace ws = context_workspace(s->ctx);
    return raycast_dispatch(s, ws, use_alpha && s->has_alpha, s->ctx->count, d_orig, d_dir, n, occlusion, d_id, d_t, d_u, d_v);
}

// ------------------------------------------------------------------------------------------------
// The entry point that allocates nothing (include/ptgpu.h): ptgpu_raycast_batch_dev's kernels with their scratch memory -- counter
// words, deferred-ray list, stack overflow of a bounded persistent grid -- in a file-scope device array instead of the context's
// grow-only hipMalloc buffers.  The array's name is only a host-side handle: its device address is taken once per device in
// ptgpu_init (hipGetSymbolAddress with that device current) and kept in the context; every GPU holds its own copy.
// ------------------------------------------------------------------------------------------------
constexpr long long NOALLOC_MAX_RAYS = PTGPU_NOALLOC_MAX_RAYS;
constexpr long long NOALLOC_PERSIST_BLOCKS = 64;  // x 4 waves x 24 KiB of overflow stack = 6 MiB
constexpr size_t NOALLOC_OFF_DEFER = 256 * sizeof(unsigned long long);  // counter words first (CTR_WORDS of them used)
constexpr size_t NOALLOC_OFF_STACK = NOALLOC_OFF_DEFER + (size_t)NOALLOC_MAX_RAYS * sizeof(int32_t);
constexpr size_t NOALLOC_STACK_BYTES = (size_t)NOALLOC_PERSIST_BLOCKS * (RC_BLOCK / 64) * ptd::stack_overflow_bytes_per_wave<16>();
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_OFF_STACK + NOALLOC_STACK_BYTES;
static_assert(CTR_WORDS <= 256 && NOALLOC_OFF_DEFER % 256 == 0 && NOALLOC_OFF_STACK % 256 == 0, "scratch regions stay 256-byte aligned");
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

// once per device, from ptgpu_init with that device current
static int noalloc_scratch_address(unsigned char** out) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(PTGPU_E_HIP, "hipGetSymbolAddress returned a NULL scratch address");
    *out = static_cast<unsigned char*>(p);
    return PTGPU_OK;
}

extern "C" int ptgpu_raycast_batch_dev_noalloc(ptgpu_scene* s, int use_alpha, const float* d_orig, const float* d_dir, int64_t n, int occlusion,
                                               int32_t* d_id, float* d_t, float* d_u, float* d_v) {
    if (!s) return fail(PTGPU_E_INVALID, "scene is NULL");
    if (n < 0) return fail(PTGPU_E_INVALID, "n < 0");
    if (n > NOALLOC_MAX_RAYS)
        return fail(PTGPU_E_UNSUPPORTED, "ptgpu_raycast_batch_dev_noalloc takes at most %lld rays per call (got %lld): split the batch or use ptgpu_raycast_batch_dev",
                    NOALLOC_MAX_RAYS, (long long)n);
    if (n == 0) return PTGPU_OK;
    if (!d_orig || !d_dir) return fail(PTGPU_E_INVALID, "ray arrays are NULL");
    ptgpu_context* c = s->ctx;
    if (!c->scratch) return fail(PTGPU_E_HIP, "the context has no scratch address (ptgpu_init did not complete)");
    HIP_TRY(hipSetDevice(c->device));
    unsigned char* base = c->scratch;  // the device address, never the array's name
    RcWorkspace ws{nullptr,
                   reinterpret_cast<unsigned long long*>(base),
                   reinterpret_cast<int32_t*>(base + NOALLOC_OFF_DEFER), (size_t)NOALLOC_MAX_RAYS,
                   reinterpret_cast<uint2*>(base + NOALLOC_OFF_STACK), NOALLOC_STACK_BYTES,
                   NOALLOC_PERSIST_BLOCKS};
    // The scratch keeps the last call's contents.  What must start at zero is the deferred-ray count (word 5), cleared by the launcher
    // before every deferring launch; list entries and stack slots are written before they are read.  Counting variants are not
    // offered here (the context's counters live in allocated memory): rays traced by this entry point do not show in ptgpu_get_counters.
    return raycast_dispatch(s, ws, use_alpha && s->has_alpha, false, d_orig, d_dir, n, occlusion, d_id, d_t, d_u, d_v);
}

extern "C" int ptgpu_raycast_batch(ptgpu_scene* s, int use_alpha, const float* orig, const float* dir, int64_t n, int occlusion,
                                   int32_t* out_id, float* out_t, float* out_u, float* out_v) {
    if (!s) return fail(PTGPU_E_INVALID, "scene is NULL");
    if (n < 0) return fail(PTGPU_E_INVALID, "n < 0");
    if (n == 0) return PTGPU_OK;
    if (!orig || !dir) return fail(PTGPU_E_INVALID, "ray arrays are NULL");
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // one grow-only staging buffer: origins, directions, then the four output arrays (a single-ray call of the reference-signature
    // RayTracer::raycast() allocates nothing after the first)
    const size_t N = (size_t)n, slot = (N * 4 + 255) & ~(size_t)255, slot3 = (N * 12 + 255) & ~(size_t)255;
    int rc = ensure_stage(c, 2 * slot3 + 4 * slot);
    if (rc != PTGPU_OK) return rc;
    unsigned char* b = c->d_stage;
    float *d_o = (float*)b, *d_d = (float*)(b + slot3);
    int32_t* d_id = out_id ? (int32_t*)(b + 2 * slot3) : nullptr;
    float* d_t = out_t ? (float*)(b + 2 * slot3 + slot) : nullptr;
    float* d_u = out_u ? (float*)(b + 2 * slot3 + 2 * slot) : nullptr;
    float* d_v = out_v ? (float*)(b + 2 * slot3 + 3 * slot) : nullptr;
    HIP_TRY(hipMemcpyAsync(d_o, orig, N * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_d, dir, N * 12, hipMemcpyHostToDevice, st));
    rc = ptgpu_raycast_batch_dev(s, use_alpha, d_o, d_d, n, occlusion, d_id, d_t, d_u, d_v);
    if (rc != PTGPU_OK) return rc;
    if (out_id) HIP_TRY(hipMemcpyAsync(out_id, d_id, N * 4, hipMemcpyDeviceToHost, st));
    if (out_t) HIP_TRY(hipMemcpyAsync(out_t, d_t, N * 4, hipMemcpyDeviceToHost, st));
    if (out_u) HIP_TRY(hipMemcpyAsync(out_u, d_u, N * 4, hipMemcpyDeviceToHost, st));
    if (out_v) HIP_TRY(hipMemcpyAsync(out_v, d_v, N * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PTGPU_OK;
}

// ---- ray generators for BASELINE config 2 (camera_ray_nodof, shadow_ray: pt_frame_rays.hip.h) ----
__global__ void k_gen_primary(ptgpu_camera cam, int width, int height, float* __restrict__ orig, float* __restrict__ dir) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)width * height) return;
    int px = (int)(i % width), py = (int)(i / width);
    float div_x = 1.0f / (float)width, div_y = 1.0f / (float)height;
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float x = fx * div_x * 2.0f - 1.0f;
    float y = fy * div_y * -2.0f + 1.0f;
    ptd::F3 Ro, Rd;
    camera_ray_nodof(cam, x, y, Ro, Rd);
    orig[i * 3] = Ro.x; orig[i * 3 + 1] = Ro.y; orig[i * 3 + 2] = Ro.z;
    dir[i * 3] = Rd.x; dir[i * 3 + 1] = Rd.y; dir[i * 3 + 2] = Rd.z;
}

__global__ void k_gen_shadow(const float* __restrict__ nrm, const uint2* __restrict__ sobol_pairs, int num_tris, ptgpu_light L, long long n, const float* __restrict__ orig,
                             const float* __restrict__ dir, const int32_t* __restrict__ id, const float* __restrict__ t,
                             const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ so, float* __restrict__ sd,
                             int32_t* __restrict__ valid) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int tri = id[i];
    ptd::F3 o = ptd::f3(0, 0, 0), d = ptd::f3(0, 0, 0);
    if (tri >= num_tris) tri = -1;  // never index past the scene with a caller-supplied id
    if (tri >= 0) shadow_ray(nrm, sobol_pairs, ptd::g_sobol_matrices, L, (unsigned long long)i, ptd::ld3(orig + i * 3), ptd::ld3(dir + i * 3), tri, t[i], u[i], v[i], o, d);
    valid[i] = tri >= 0 ? 1 : 0;
    so[i * 3] = o.x; so[i * 3 + 1] = o.y; so[i * 3 + 2] = o.z;
    sd[i * 3] = d.x; sd[i * 3 + 1] = d.y; sd[i * 3 + 2] = d.z;
}

extern "C" int ptgpu_trace_frame_dev(ptgpu_scene* s, const ptgpu_camera* cam, const ptgpu_light* light, int32_t width, int32_t height, int32_t* d_id,
                                     float* d_t, float* d_u, float* d_v, int32_t* d_shadow_id) {
    if (!s || !cam || !light) return fail(PTGPU_E_INVALID, "NULL argument");
    if (!d_id || !d_t || !d_u || !d_v) return fail(PTGPU_E_INVALID, "the primary hit record arrays are required");
    if (width <= 0 || height <= 0 || (long long)width * height > 0x7fffffffLL) return fail(PTGPU_E_INVALID, "image size %dx%d", width, height);
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    ptd::TravScene trav = s->trav;
    trav.alpha_on = 0;
    const int tw = env_int("PTGPU_FRAME_TILE_W", 8);
    if (tw != 4 && tw != 8 && tw != 16 && tw != 32 && tw != 64)  // the instantiated set: the grid below must describe the kernel that runs
        return fail(PTGPU_E_INVALID, "PTGPU_FRAME_TILE_W=%d: the frame kernel is built for tiles 4, 8, 16, 32 or 64 pixels wide", tw);
    const int th = 64 / tw;
    const int tiles_x = (width + tw - 1) / tw, tiles_y = (height + th - 1) / th;
    const dim3 grid((unsigned)(tiles_x * tiles_y));
    // LDS stack entries per lane.  16 + the rest of the stack (tree depth - 1 - 16 entries, usually a handful) in a global-memory region
    // per workgroup: 20 resident waves per CU instead of the 13 that 24 entries allow -- the 1080p head frame 0.258 -> 0.247 ms
    // (12 / 10 / 8 entries: 0.247 / 0.245 / 0.246, for more overflow memory), and no limit on the tree's depth.  PTGPU_FRAME_NLDS=24
    // and the other tile widths select the all-LDS form (trees up to depth 25) for A/B timing.
    const int fn = env_int("PTGPU_FRAME_NLDS", 16);
    if (fn != 16 && fn != 24) return fail(PTGPU_E_INVALID, "PTGPU_FRAME_NLDS=%d: the frame kernel is built with 16 or 24 LDS stack entries", fn);
    const int ovf_entries = s->depth - 1 > 16 ? s->depth - 1 - 16 : 0;
    const size_t ovf_bytes = (size_t)grid.x * (size_t)(ovf_entries > 0 ? ovf_entries : 1) * 64 * sizeof(uint2);
    // the overflow buffer is grow-only: beyond 4 GiB (an 8K frame on a 40-level tree) the all-LDS form takes over where the tree allows it
    const bool regions_fit = ovf_bytes <= ((size_t)4 << 30);
    if (fn == 16 && tw == 8 && !regions_fit && s->depth - 1 > 24)
        return fail(PTGPU_E_UNSUPPORTED, "ptgpu_trace_frame_dev: a %dx%d frame on a tree of depth %d needs %zu bytes of stack overflow regions; use "
                    "ptgpu_gen_primary_dev / ptgpu_raycast_batch_dev / ptgpu_gen_shadow_dev", width, height, s->depth, ovf_bytes);
    const bool with_regions = fn == 16 && tw == 8 && regions_fit;
    if (with_regions) {
        const int rc = ensure_stack_overflow(c, ovf_bytes);
        if (rc != PTGPU_OK) return rc;
    } else if (s->depth - 1 > 24) {
        return fail(PTGPU_E_UNSUPPORTED, "ptgpu_trace_frame_dev: tree depth %d needs more than the 24 LDS stack entries of the all-LDS frame kernels "
                    "(PTGPU_FRAME_NLDS / PTGPU_FRAME_TILE_W select them)", s->depth);
    }
    ptd::launch_trace_frame(c->stream, grid, with_regions, tw, trav, s->d_nrm, c->d_sobol_pairs, c->d_sobol_matrices, *cam, *light, (int)width, (int)height,
                            tiles_x, d_id, d_t, d_u, d_v, d_shadow_id, c->d_counters, c->d_stack_ovf, ovf_entries);
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

extern "C" int ptgpu_gen_primary_dev(ptgpu_context* c, const ptgpu_camera* cam, int32_t width, int32_t height, float* d_orig, float* d_dir) {
    if (!c || !cam || !d_orig || !d_dir) return fail(PTGPU_E_INVALID, "NULL argument");
    if (width <= 0 || height <= 0) return fail(PTGPU_E_INVALID, "image size %dx%d", width, height);
    HIP_TRY(hipSetDevice(c->device));
    long long n = (long long)width * height;
    hipLaunchKernelGGL(k_gen_primary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, *cam, width, height, d_orig, d_dir);
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

extern "C" int ptgpu_gen_shadow_dev(ptgpu_scene* s, const ptgpu_light* light, int64_t n, const float* d_orig, const float* d_dir,
                                    const int32_t* d_id, const float* d_t, const float* d_u, const float* d_v, float* d_sh_orig,
                                    float* d_sh_dir, int32_t* d_sh_valid) {
    if (!s || !light) return fail(PTGPU_E_INVALID, "NULL argument");
    if (n < 0) return fail(PTGPU_E_INVALID, "n < 0");
    if (n == 0) return PTGPU_OK;
    if (!d_orig || !d_dir || !d_id || !d_t || !d_u || !d_v || !d_sh_orig || !d_sh_dir || !d_sh_valid)
        return fail(PTGPU_E_INVALID, "NULL device array");
    HIP_TRY(hipSetDevice(s->ctx->device));
    if (n > 0x7fffffffLL) return fail(PTGPU_E_INVALID, "n too large (the Sobol index of a shadow ray is its ray index, 31 bits)");
    hipLaunchKernelGGL(k_gen_shadow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->ctx->stream, s->d_nrm, s->ctx->d_sobol_pairs, s->num_tris, *light, (long long)n,
                       d_orig, d_dir, d_id, d_t, d_u, d_v, d_sh_orig, d_sh_dir, d_sh_valid);
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// PathTraceRenderer: one pass of pathTraceBlock over a set of 32x32 blocks (reference src/PathTraceRenderer.cpp:663-790)
//   k_integrate        per-sample radiance (tracePath) -> rad[padded sample id]
//   k_accum_sobol      m_sobolBlock == true:  pixel += (color, 1) / NumAaRays in sample order i (:727-732)
//   k_accum_gauss      m_sobolBlock == false: Gaussian-weighted per-pixel mean over k, saturate, pixel += (:743-787)
//   k_spill_*          Sobol-block samples whose pixel lies outside their own block (the position rounds up to the block's far edge)
//   k_resolve          updatePicture (:855-871)
// ------------------------------------------------------------------------------------------------
#ifndef PT_BLOCK_THREADS
#define PT_BLOCK_THREADS 64  // one wave per workgroup (measured 64 / 128 / 256: 30.3 / 30.7 / 31.0 ms per 16 spp pass)
#endif
constexpr int PT_BLOCK = PT_BLOCK_THREADS;
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for (launch_bounds 2nd argument): <= 128 VGPRs.
                        // Measured on the configs[2] frame: 4 waves + 16 LDS stack entries 32.0 ms vs 3 waves + 24 entries 34.6 ms per 16 spp
#endif

template <bool COUNT, int NLDS, bool OVF, bool ALPHA, bool EXT = false>
#ifdef PT_WAVES_PER_EU
// Diagnostic build: force the register allocator to a given occupancy.  5 waves/SIMD = 96 VGPRs + 236 B of scratch per lane: 27.3 ms per
// 16 spp pass against 25.3 ms for the 4-wave / 128-VGPR build below (whose run time scales almost linearly with waves: 2 / 3 / 4 waves
// per SIMD = 40.9 / 32.5 / 25.5 ms, measured by padding the workgroup's LDS) -- the kernel is latency-bound, but the state a lane
// carries across a bounce does not fit 96 registers together with the hit block's temporaries.
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(PT_WAVES_PER_EU, PT_WAVES_PER_EU))) void k_integrate(
#else
__global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES) void k_integrate(
#endif
ptd::TravScene TS, ptd::ShadeScene SS, ptgpu_params prm, ptgpu_camera cam,
                                                        const ptgpu_light* __restrict__ lights, ptd::PassDesc pd, float4* __restrict__ rad,
                                                        unsigned long long* counters, int service_threshold, uint2* stack_ovf) {
    __shared__ uint2 lds_stack[PT_BLOCK / 64][NLDS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wave_id = (size_t)blockIdx.x * (PT_BLOCK / 64) + wave;
    ptd::integrate<COUNT, NLDS, OVF, PT_BLOCK, ALPHA, EXT>(TS, SS, prm, cam, lights, pd, rad, counters, &lds_stack[wave][0][lane],
                                               OVF ? stack_ovf + wave_id * (64 - NLDS) * 64 + lane : nullptr, service_threshold);
}

// the pooled form (csrc/pt_integrator_pool.hip.h): one large workgroup per CU, path state in LDS slots, trace waves + shade waves
template <bool ALPHA>
__global__ __launch_bounds__(ptd::POOL_THREADS) void k_integrate_pool(ptd::TravScene TS, ptd::ShadeScene SS, ptgpu_params prm, ptgpu_camera cam,
                                                                      const ptgpu_light* __restrict__ lights, ptd::PassDesc pd, float4* __restrict__ rad,
                                                                      unsigned long long* counters, uint2* stack_ovf, ptd::PoolCold* cold, int knobs) {
    extern __shared__ __align__(16) unsigned char pool_smem[];
    ptd::integrate_pool<ptd::POOL_NLDS, ALPHA>(TS, SS, prm, cam, lights, pd, rad, counters, pool_smem, stack_ovf, cold, knobs);
}

__device__ __forceinline__ void block_rect(const ptgpu_params& prm, const ptd::PassDesc& pd, int jj, int& bx, int& by, int& bw, int& bh) {
    const int idx = pd.block_begin + (pd.j0 + jj) * pd.block_stride;
    bx = (idx % pd.blocks_x) * 32; by = (idx / pd.blocks_x) * 32;
    bw = (bx + 32 < prm.width) ? 32 : prm.width - bx;
    bh = (by + 32 < prm.height) ? 32 : prm.height - by;
}

__global__ __launch_bounds__(256) void k_accum_sobol(ptgpu_params prm, ptd::PassDesc pd, const float4* __restrict__ rad, float* __restrict__ accum,
                                                     ForeignSpill* __restrict__ spill, unsigned long long spill_capacity, unsigned long long* counters) {
    __shared__ float4 acc[1024];
    __shared__ int owner[1024];
    if (ptd::pass_cancelled(pd)) return;  // the batch's radiance may be incomplete: none of it is added (workgroup-uniform)
    const int jj = blockIdx.x, tid = threadIdx.x;
    int bx, by, bw, bh;
    block_rect(prm, pd, jj, bx, by, bw, bh);
    const int n = bw * bh * prm.num_aa_rays;
    const float scale = 1.0f / (float)prm.num_aa_rays;
    float4* img = reinterpret_cast<float4*>(accum);
    for (int p = tid; p < 1024; p += 256) {
        const int lx = p & 31, ly = p >> 5;
        acc[p] = (lx < bw && ly < bh) ? img[(size_t)(by + ly) * prm.width + bx + lx] : make_float4(0, 0, 0, 0);
        owner[p] = 0x7fffffff;
    }
    __syncthreads();
    const float4* src = rad + (size_t)jj * pd.samples_per_block;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        bool pending = false;
        float4 r = make_float4(0, 0, 0, 0);
        int lp = 0;
        if (i < n) {
            r = src[i];
            const int gp = __float_as_int(r.w);
            const int gx = gp % prm.width, gy = gp / prm.width;
            if (gx >= bx && gx < bx + bw && gy >= by && gy < by + bh) {
                lp = (gy - by) * 32 + (gx - bx);
                pending = true;
            } else {
                // one pixel outside this block (the position rounded up to the block's far edge): kept for the end of the pass
                const unsigned long long slot = atomicAdd(&counters[CTR_SPILL], 1ull);
                if (slot < spill_capacity)
                    spill[slot] = ForeignSpill{r.x * scale, r.y * scale, r.z * scale, 1.0f * scale, gp, pd.block_begin + (pd.j0 + jj) * pd.block_stride, i, 0};
                else atomicAdd(&counters[CTR_DROPPED], 1ull);
            }
        }
        // add in sample order: among the pending samples of one pixel the lowest thread (= lowest i) goes first
        while (__syncthreads_or(pending ? 1 : 0)) {
            if (pending) atomicMin(&owner[lp], tid);
            __syncthreads();
            if (pending && owner[lp] == tid) {
                float4 a = acc[lp];
                a.x += r.x * scale; a.y += r.y * scale; a.z += r.z * scale; a.w += 1.0f * scale;
                acc[lp] = a;
                pending = false;
                owner[lp] = 0x7fffffff;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int p = tid; p < 1024; p += 256) {
        const int lx = p & 31, ly = p >> 5;
        if (lx < bw && ly < bh) img[(size_t)(by + ly) * prm.width + bx + lx] = acc[p];
    }
}

__device__ __forceinline__ bool owns_pixel(const TileOwner& o, int gp, int width) {
    const int idx = ((gp / width) >> 5) * o.blocks_x + ((gp % width) >> 5);
    return idx >= o.block_begin && idx < o.block_end && (idx - o.block_begin) % o.block_stride == 0;
}

// ------------------------------------------------------------------------------------------------
// Out-of-block samples (pt_context.h): added after the block sums, per pixel in (source block, sample) order, whatever order the lists
// were filled in -- so one GPU, N GPUs and the sequential CPU loop produce the same bits even when several of them meet in one pixel.
//   sort 1 (stable radix): key = invalid | source block | sample          sort 2 (stable radix): key = pixel
//   then one thread per run of equal pixels walks its run and adds (or, for a pixel the call does not own, appends to the foreign list)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ const ForeignSpill* spill_entry(const SpillLists& L, uint32_t idx) {
    return reinterpret_cast<const ForeignSpill*>(L.first + (size_t)(idx / (uint32_t)L.per_list) * L.list_stride_f4) + (idx % (uint32_t)L.per_list);
}
__device__ __forceinline__ uint32_t spill_list_count(const SpillLists& L, uint32_t list) {
    long long n = L.local_count ? (long long)*L.local_count : (long long)__float_as_int((L.first + (size_t)list * L.list_stride_f4 - 1)->x);
    return (uint32_t)(n < 0 ? 0 : (n > L.per_list ? L.per_list : n));
}
constexpr unsigned long long SPILL_KEY_INVALID = 1ull << 53;  // above the source block (21 bits at 32..52) and the sample (30 bits)
__global__ __launch_bounds__(256) void k_spill_keys(SpillLists L, uint32_t m, int num_pixels, unsigned long long* __restrict__ key, uint32_t* __restrict__ idx) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= m) return;
    unsigned long long k = SPILL_KEY_INVALID;
    if (e % (uint32_t)L.per_list < spill_list_count(L, e / (uint32_t)L.per_list)) {
        const ForeignSpill* s = spill_entry(L, e);
        if (s->pixel >= 0 && s->pixel < num_pixels)
            k = ((unsigned long long)((uint32_t)s->src_block & 0x1fffffu) << 32) | (unsigned long long)((uint32_t)s->sample & 0x3fffffffu);
    }
    key[e] = k;
    idx[e] = e;
}
__global__ __launch_bounds__(256) void k_spill_pixel_keys(SpillLists L, uint32_t m, const unsigned long long* __restrict__ key_sorted,
                                                          const uint32_t* __restrict__ idx_sorted, uint32_t* __restrict__ pixel_key) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= m) return;
    pixel_key[e] = (key_sorted[e] & SPILL_KEY_INVALID) ? 0xffffffffu : (uint32_t)spill_entry(L, idx_sorted[e])->pixel;
}
__global__ __launch_bounds__(256) void k_spill_apply(SpillLists L, uint32_t m, const uint32_t* __restrict__ pixel_sorted, const uint32_t* __restrict__ idx_sorted,
                                                     float* __restrict__ accum, int width, bool filter, TileOwner own, ForeignSpill* __restrict__ foreign,
                                                     unsigned long long* counters) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= m) return;
    const uint32_t gp = pixel_sorted[e];
    if (gp == 0xffffffffu || (e > 0 && pixel_sorted[e - 1] == gp)) return;  // not the first entry of its pixel
    const bool mine = !filter || owns_pixel(own, (int)gp, width);
    float4* px = reinterpret_cast<float4*>(accum) + gp;
    float4 a = mine ? *px : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t j = e; j < m && pixel_sorted[j] == gp; ++j) {
        const ForeignSpill s = *spill_entry(L, idx_sorted[j]);
        if (mine) { a.x += s.r; a.y += s.g; a.z += s.b; a.w += s.w; }
        else {
            // a pixel of a block this call does not render: waits in the foreign list for ptgpu_tiles_pack_dev (which sorts again)
            const unsigned long long slot = atomicAdd(&counters[CTR_FOREIGN], 1ull);
            if (slot < (unsigned long long)FOREIGN_CAPACITY) foreign[slot] = s;
            else atomicAdd(&counters[CTR_DROPPED], 1ull);
        }
    }
    if (mine) *px = a;
}
// tiles sharded over ranks: every out-of-block sample of the pass goes to the foreign list (all ranks add all of them after the exchange)
__global__ __launch_bounds__(256) void k_spill_to_foreign(const ForeignSpill* __restrict__ spill, unsigned long long spill_capacity,
                                                          ForeignSpill* __restrict__ foreign, unsigned long long* counters) {
    unsigned long long n = counters[CTR_SPILL];
    if (n > spill_capacity) n = spill_capacity;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * 256ull) {
        const unsigned long long slot = atomicAdd(&counters[CTR_FOREIGN], 1ull);
        if (slot < (unsigned long long)FOREIGN_CAPACITY) foreign[slot] = spill[e];
        else atomicAdd(&counters[CTR_DROPPED], 1ull);
    }
}

int ptgpu_apply_spills_sorted(ptgpu_context* c, const SpillLists& lists, float* d_accum, int32_t width, int32_t height, const TileOwner* own) {
    const long long m64 = (long long)lists.per_list * lists.num_lists;
    if (m64 <= 0) return PTGPU_OK;
    if (m64 > SPILL_CAPACITY_MAX * 8) return fail(PTGPU_E_UNSUPPORTED, "%lld out-of-block sample slots to sort", m64);
    const uint32_t m = (uint32_t)m64;
    hipStream_t st = c->stream;
    // workspace: key[m] u64 x2, idx[m] u32 x3, pixel key[m] u32 x2, rocprim temporary storage
    size_t t1 = 0, t2 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)m, 0u, 54u, st));
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t2, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)m, 0u, 32u, st));
    const size_t temp = ((t1 > t2 ? t1 : t2) + 255) & ~(size_t)255, a8 = ((size_t)m * 8 + 255) & ~(size_t)255, a4 = ((size_t)m * 4 + 255) & ~(size_t)255;
    const size_t need = 2 * a8 + 5 * a4 + temp;
    if (need > c->sort_ws_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c->d_sort_ws) { HIP_TRY(hipFree(c->d_sort_ws)); c->d_sort_ws = nullptr; c->sort_ws_bytes = 0; }
        HIP_TRY(hipMalloc(&c->d_sort_ws, need));
        c->sort_ws_bytes = need;
    }
    unsigned char* w = c->d_sort_ws;
    unsigned long long *key = (unsigned long long*)w, *key2 = (unsigned long long*)(w + a8);
    uint32_t *idx = (uint32_t*)(w + 2 * a8), *idx2 = idx + a4 / 4, *idx3 = idx2 + a4 / 4, *pk = idx3 + a4 / 4, *pk2 = pk + a4 / 4;
    void* tmp = w + 2 * a8 + 5 * a4;
    const dim3 grid((m + 255u) / 256u), block(256);
    hipLaunchKernelGGL(k_spill_keys, grid, block, 0, st, lists, m, (int)((long long)width * height), key, idx);
    size_t tb = temp;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, key, key2, idx, idx2, (size_t)m, 0u, 54u, st));
    hipLaunchKernelGGL(k_spill_pixel_keys, grid, block, 0, st, lists, m, key2, idx2, pk);
    tb = temp;
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, pk, pk2, idx2, idx3, (size_t)m, 0u, 32u, st));
    hipLaunchKernelGGL(k_spill_apply, grid, block, 0, st, lists, m, pk2, idx3, d_accum, (int)width, own != nullptr, own ? *own : TileOwner{0, 0, 1, 1},
                       c->d_foreign, c->d_counters);
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

// Upper bound of the out-of-block samples of one pass of `samples` samples, for sizing the pass's list on the host: a coordinate
// c = b0 + bw * sobol() lands on b0 + bw when sobol() >= 1 - ulp(c) / (2 bw); ulp(c) <= 2^-11 for images up to 8192 pixels wide, bw = 32
// except in clipped edge blocks => below 2^-16 per sample for both coordinates together.  Four times that, plus a floor.
static size_t spill_bound(long long samples) {
    long long b = (samples >> 14) + 4096;
    return (size_t)(b > SPILL_CAPACITY_MAX ? SPILL_CAPACITY_MAX : b);
}

__global__ __launch_bounds__(256) void k_accum_gauss(ptgpu_params prm, ptd::PassDesc pd, const float4* __restrict__ rad, float* __restrict__ accum) {
    if (ptd::pass_cancelled(pd)) return;  // as in k_accum_sobol
    const int jj = blockIdx.x;
    int bx, by, bw, bh;
    block_rect(prm, pd, jj, bx, by, bw, bh);
    float4* img = reinterpret_cast<float4*>(accum);
    for (int i = threadIdx.x; i < bw * bh; i += 256) {
        const int pixel_x = bx + (i % bw), pixel_y = by + (i / bw);
        const float4* src = rad + (size_t)jj * pd.samples_per_block + (size_t)i * prm.num_aa_rays;
        float cx = 0.0f, cy = 0.0f, cz = 0.0f, cw = 0.0f;
        // a pixel's samples are contiguous: eight at a time (one 128-byte line when aligned), all loads issued before the first sum, so a
        // line is used up while it is there (one load per trip re-fetched lines that 16 waves per CU had pushed out of the 32 KB L1);
        // the sums stay in sample order
        int k = 0;
        for (; k + 8 <= prm.num_aa_rays; k += 8) {
            float4 r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = src[k + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float g = r[q].w;
                cx += g * r[q].x; cy += g * r[q].y; cz += g * r[q].z; cw += g * 1.0f;
            }
        }
        for (; k < prm.num_aa_rays; ++k) {
            const float4 r = src[k];
            const float g = r.w;
            cx += g * r.x; cy += g * r.y; cz += g * r.z; cw += g * 1.0f;
        }
        cx = cx / cw; cy = cy / cw; cz = cz / cw; cw = cw / cw;
        // saturate(), reference :36-49
        const float Pr = 0.299f, Pg = 0.587f, Pb = 0.114f;
        const float amount = prm.attenuation;
        const float P = sqrtf(cx * cx * Pr + cy * cy * Pg + cz * cz * Pb);
        cx = P + (cx - P) * amount; cy = P + (cy - P) * amount; cz = P + (cz - P) * amount;
        cx = cx < 0.0f ? 0.0f : (cx > 1.0f ? 1.0f : cx);
        cy = cy < 0.0f ? 0.0f : (cy > 1.0f ? 1.0f : cy);
        cz = cz < 0.0f ? 0.0f : (cz > 1.0f ? 1.0f : cz);
        cw = cw < 0.0f ? 0.0f : (cw > 1.0f ? 1.0f : cw);
        float4 a = img[(size_t)pixel_y * prm.width + pixel_x];
        a.x += cx; a.y += cy; a.z += cz; a.w += cw;
        img[(size_t)pixel_y * prm.width + pixel_x] = a;
    }
}

__global__ void k_resolve(const float4* __restrict__ accum, long long n, float4* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 D = accum[i];
    if (D.w != 0.0f) {
        float r = 1.0f / D.w;
        D = make_float4(D.x * r, D.y * r, D.z * r, D.w * r);
    }
    const float g = 1.0f / 2.2f;
    out[i] = make_float4(ptd::ptm_pow(D.x, g), ptd::ptm_pow(D.y, g), ptd::ptm_pow(D.z, g), D.w);
}

static int check_params(const ptgpu_params* p) {
    if (p->width <= 0 || p->height <= 0) return fail(PTGPU_E_INVALID, "image size %dx%d", p->width, p->height);
    if ((long long)p->width * p->height > 0x7fffffffLL) return fail(PTGPU_E_INVALID, "image too large");
    if (p->num_aa_rays < 1 || p->num_aa_rays > 65536) return fail(PTGPU_E_INVALID, "num_aa_rays %d out of range [1,65536]", p->num_aa_rays);
    if (p->whitted && p->num_light_rays < 1) return fail(PTGPU_E_INVALID, "num_light_rays must be >= 1 in Whitted mode");
    if (p->bounces < 0 && !(p->termination_prob > 0.0f)) return fail(PTGPU_E_INVALID, "termination_prob must be > 0 with Russian roulette");
    if (!p->sobol_block && !(p->filter_width > 0.0f)) return fail(PTGPU_E_INVALID, "filter_width must be > 0");
    return PTGPU_OK;
}

// light table of the context: re-uploaded only when it changed (synchronous, rare)
static int upload_lights(ptgpu_context* c, const ptgpu_light* lights, int32_t num_lights) {
    if ((int)c->h_lights.size() == num_lights && memcmp(c->h_lights.data(), lights, sizeof(ptgpu_light) * num_lights) == 0) return PTGPU_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_lights) { HIP_TRY(hipFree(c->d_lights)); c->d_lights = nullptr; }
    c->h_lights.clear();
    HIP_TRY(hipMalloc(&c->d_lights, sizeof(ptgpu_light) * num_lights));
    HIP_TRY(hipMemcpy(c->d_lights, lights, sizeof(ptgpu_light) * num_lights, hipMemcpyHostToDevice));
    c->h_lights.assign(lights, lights + num_lights);
    return PTGPU_OK;
}

template <bool COUNT, int NLDS, bool OVF, bool ALPHA, bool EXT = false>
static int launch_integrate(ptgpu_scene* s, const ptd::TravScene& trav, const ptd::ShadeScene& ss, const ptgpu_params* prm, const ptgpu_camera* cam,
                            const ptd::PassDesc& pd, float4* rad = nullptr) {
    ptgpu_context* c = s->ctx;
    static std::atomic<int> per_cu_cache{0};  // same value whichever thread computes it first
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_integrate<COUNT, NLDS, OVF, ALPHA, EXT>, PT_BLOCK, 0) != hipSuccess || v < 1) v = 1;
        per_cu = v;
        per_cu_cache.store(v, std::memory_order_relaxed);
    }
    const long long total = EXT ? (long long)pd.num_ext : (long long)pd.num_blocks * pd.samples_per_block;
    long long blocks = (total + PT_BLOCK - 1) / PT_BLOCK;
    if (blocks < 1) blocks = 1;
    const long long resident = (long long)c->num_cu * per_cu * env_int("PTGPU_PT_OVERSUB", 4);
    if (blocks > resident) blocks = resident;
    if (OVF) {
        int rc = ensure_stack_overflow(c, (size_t)blocks * (PT_BLOCK / 64) * ptd::stack_overflow_bytes_per_wave<NLDS>());
        if (rc != PTGPU_OK) return rc;
    }
    hipLaunchKernelGGL((k_integrate<COUNT, NLDS, OVF, ALPHA, EXT>), dim3((unsigned)blocks), dim3(PT_BLOCK), (size_t)env_int("PTGPU_PT_EXTRA_LDS", 0), c->stream, trav, ss, *prm, *cam, c->d_lights, pd, rad ? rad : c->d_rad,
                       c->d_counters, (env_int("PTGPU_PT_SERVICE", s->pt_service) & 0xff) | (env_int("PTGPU_PT_CTLMIN", 16) << 8), c->d_stack_ovf);
    return PTGPU_OK;
}

template <bool ALPHA>
static int launch_integrate_pool(ptgpu_scene* s, const ptd::TravScene& trav, const ptd::ShadeScene& ss, const ptgpu_params* prm, const ptgpu_camera* cam,
                                 const ptd::PassDesc& pd) {
    ptgpu_context* c = s->ctx;
    const size_t lds = ptd::pool_lds_bytes();
    static std::atomic<int> attr_set{0};
    if (!attr_set.load(std::memory_order_relaxed)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_integrate_pool<ALPHA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.store(1, std::memory_order_relaxed);
    }
    const long long total = (long long)pd.num_blocks * pd.samples_per_block;
    long long blocks = (total + ptd::POOL_CHUNK - 1) / ptd::POOL_CHUNK;
    if (blocks < 1) blocks = 1;
    if (blocks > c->num_cu) blocks = c->num_cu;  // one workgroup per CU (LDS-bound), resident for the whole launch
    // one allocation: the trace waves' stack overflow regions, then the slots' shade-only state
    const size_t ovf_bytes = (size_t)blocks * ptd::POOL_TW * ptd::stack_overflow_bytes_per_wave<ptd::POOL_NLDS>();
    int rc = ensure_stack_overflow(c, ovf_bytes + (size_t)blocks * ptd::pool_cold_bytes_per_group());
    if (rc != PTGPU_OK) return rc;
    hipLaunchKernelGGL((k_integrate_pool<ALPHA>), dim3((unsigned)blocks), dim3(ptd::POOL_THREADS), lds, c->stream, trav, ss, *prm, *cam, c->d_lights, pd, c->d_rad,
                       c->d_counters, c->d_stack_ovf, reinterpret_cast<ptd::PoolCold*>(reinterpret_cast<unsigned char*>(c->d_stack_ovf) + ovf_bytes), env_int("PTGPU_POOL_REFILL", 16) | (env_int("PTGPU_POOL_LOW", 32) << 8) | ((env_int("PTGPU_POOL_WATCHDOG_LOG2", ptd::POOL_WATCHDOG_LOG2) & 63) << 16));
    return PTGPU_OK;
}

extern "C" int ptgpu_render_pass_dev(ptgpu_scene* s, const ptgpu_params* prm, const ptgpu_camera* cam, const ptgpu_light* lights, int32_t num_lights,
                                     int32_t block_begin, int32_t block_end, int32_t block_stride, float* d_accum) {
    if (!s || !prm || !cam || !lights || !d_accum) return fail(PTGPU_E_INVALID, "NULL argument");
    int rc = check_params(prm);
    if (rc != PTGPU_OK) return rc;
    if (num_lights < 1 || num_lights > 1024) return fail(PTGPU_E_INVALID, "num_lights %d out of range [1,1024]", num_lights);
    const int blocks_x = (prm->width + 31) / 32, blocks_y = (prm->height + 31) / 32;
    const int nblocks = blocks_x * blocks_y;
    if (block_end < 0 || block_end > nblocks) block_end = nblocks;
    if (block_begin < 0 || block_stride < 1) return fail(PTGPU_E_INVALID, "block range [%d,%d) stride %d", block_begin, block_end, block_stride);
    if (block_begin >= block_end) return PTGPU_OK;
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    rc = upload_lights(c, lights, num_lights);
    if (rc != PTGPU_OK) return rc;
    rc = ptgpu_ensure_spill_lists(c);
    if (rc != PTGPU_OK) return rc;
    rc = scene_probe(s);  // first pass on this scene only
    if (rc != PTGPU_OK) return rc;
    const int count_all = (block_end - block_begin + block_stride - 1) / block_stride;
    // this pass's list of out-of-block samples holds `bound` entries (more are counted as dropped)
    const size_t bound = spill_bound((long long)count_all * 1024LL * prm->num_aa_rays);
    if (prm->sobol_block) {
        const size_t want = bound;
        if (want > c->spill_capacity) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_spill) { HIP_TRY(hipFree(c->d_spill)); c->d_spill = nullptr; c->spill_capacity = 0; }
            HIP_TRY(hipMalloc(&c->d_spill, want * sizeof(ForeignSpill)));
            c->spill_capacity = want;
        }
    }
    // the spill list is per pass (all batches of this call append to one list, applied after the last block sum); the dropped-sample
    // count is NOT: it stands until ptgpu_spill_status / ptgpu_render_pass has reported it, so a full list in pass k of a sequence of
    // _dev calls cannot go unnoticed
    if (prm->sobol_block) hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, c->stream, c->d_counters + CTR_SPILL, 1);
    const bool pooled = env_int("PTGPU_PT_POOL", 0) != 0;  // the pooled form of the integrator (csrc/pt_integrator_pool.hip.h), off unless asked for
    // the pooled integrator's abort word is per pass too, whichever integrator this pass runs (ptgpu_spill_status reads it after every pass)
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, c->stream, c->d_counters + ptd::CTR_POOL_ABORT, 1);
    const int count = (block_end - block_begin + block_stride - 1) / block_stride;  // blocks of this call
    const long long spb = 1024LL * prm->num_aa_rays;
    // samples per integrator launch: every launch ends with a tail of waves draining, so few, large launches (the radiance buffer is
    // 16 B per sample: 4 GiB at 256 Mi, out of 288 GB).  configs[2], ms per pass: 16 Mi 512, 64 Mi 466, 256 Mi 459, 512 Mi 457.
    long long batch_samples = (long long)env_int("PTGPU_PT_BATCH_MSAMPLES", 256) << 20;
    if (batch_samples > (1LL << 30)) batch_samples = 1LL << 30;  // sample ids of one launch are 31-bit in the kernels
    if (spb > (1LL << 30)) return fail(PTGPU_E_UNSUPPORTED, "num_aa_rays %d: more than 2^30 samples per block", prm->num_aa_rays);
    long long batch_blocks = batch_samples / spb;
    if (batch_blocks < 1) batch_blocks = 1;
    if (batch_blocks > count) batch_blocks = count;
    const size_t need = (size_t)(batch_blocks * spb);
    if (need > c->rad_capacity) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rad) { HIP_TRY(hipFree(c->d_rad)); c->d_rad = nullptr; c->rad_capacity = 0; }
        HIP_TRY(hipMalloc(&c->d_rad, need * sizeof(float4)));
        c->rad_capacity = need;
    }
    ptd::TravScene trav = s->trav;
    trav.alpha_on = (prm->use_alpha && s->has_alpha) ? 1 : 0;
    ptd::ShadeScene ss;
    ss.pos = s->d_pos; ss.nrm = s->d_nrm; ss.uv = s->d_uv; ss.materials = s->d_materials; ss.textures = s->d_textures;
    ss.sobol_pairs = c->d_sobol_pairs;
    const int stack_need = s->depth - 1, nlds = env_int("PTGPU_PT_NLDS", 16);
    const uint32_t epoch = c->pass_epoch.fetch_add(1u, std::memory_order_acq_rel) + 1u;  // this pass's number, for ptgpu_cancel
    for (int j0 = 0; j0 < count; j0 += (int)batch_blocks) {
        ptd::PassDesc pd;
        pd.block_begin = block_begin; pd.block_stride = block_stride; pd.j0 = j0;
        pd.num_blocks = (count - j0 < batch_blocks) ? count - j0 : (int)batch_blocks;
        pd.blocks_x = blocks_x; pd.samples_per_block = (int32_t)spb; pd.num_lights = num_lights;
        pd.inv_prob = 1.0f / prm->termination_prob;
        pd.ext_xy = nullptr; pd.ext_base = nullptr; pd.ext_rng = nullptr; pd.num_ext = 0;
        pd.cancel = c->d_cancel; pd.epoch = epoch;
        // the alpha-cutout code is only compiled into the variants a scene with alpha maps takes (fewer registers for all others)
#define PT_LAUNCH(COUNT_, NLDS_, OVF_) (trav.alpha_on ? launch_integrate<COUNT_, NLDS_, OVF_, true>(s, trav, ss, prm, cam, pd) \
                                                      : launch_integrate<COUNT_, NLDS_, OVF_, false>(s, trav, ss, prm, cam, pd))
        const int abs_bounces = prm->bounces < 0 ? -prm->bounces : prm->bounces;
        if (pooled && !c->count && !prm->whitted && abs_bounces < 60000)
            rc = trav.alpha_on ? launch_integrate_pool<true>(s, trav, ss, prm, cam, pd) : launch_integrate_pool<false>(s, trav, ss, prm, cam, pd);
        else if (c->count) rc = launch_integrate<true, 16, true, true>(s, trav, ss, prm, cam, pd);
        else if (nlds == 12) rc = PT_LAUNCH(false, 12, true);
        else if (stack_need > 24 || (stack_need > 16 && nlds == 16)) rc = PT_LAUNCH(false, 16, true);
        else if (nlds == 16) rc = PT_LAUNCH(false, 16, false);
        else rc = PT_LAUNCH(false, 24, false);
#undef PT_LAUNCH
        if (rc != PTGPU_OK) return rc;
        if (prm->sobol_block) {
            hipLaunchKernelGGL(k_accum_sobol, dim3(pd.num_blocks), dim3(256), 0, c->stream, *prm, pd, c->d_rad, d_accum, c->d_spill,
                               (unsigned long long)bound, c->d_counters);
        } else {
            hipLaunchKernelGGL(k_accum_gauss, dim3(pd.num_blocks), dim3(256), 0, c->stream, *prm, pd, c->d_rad, d_accum);
        }
        HIP_TRY(hipGetLastError());
    }
    if (prm->sobol_block) {
        // the pass's out-of-block samples, once, after its last block sum
        if (block_stride > 1) {
            // tiles sharded over ranks: all of them travel with this rank's tiles and every rank adds every rank's after the exchange
            hipLaunchKernelGGL(k_spill_to_foreign, dim3((unsigned)((bound + 255) / 256 < 1024 ? (bound + 255) / 256 : 1024)), dim3(256), 0, c->stream,
                               c->d_spill, (unsigned long long)bound, c->d_foreign, c->d_counters);
            HIP_TRY(hipGetLastError());
        } else {
            const TileOwner own{block_begin, block_end, block_stride, blocks_x};
            const SpillLists lists{reinterpret_cast<const float4*>(c->d_spill), 0, (int32_t)bound, 1, c->d_counters + CTR_SPILL};
            rc = ptgpu_apply_spills_sorted(c, lists, d_accum, prm->width, prm->height, &own);
            if (rc != PTGPU_OK) return rc;
        }
    }
    return PTGPU_OK;
}

extern "C" int ptgpu_spill_status(ptgpu_context* c, uint64_t out[3]) {
    if (!c || !out) return fail(PTGPU_E_INVALID, "NULL argument");
    unsigned long long h[5];
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(h, c->d_counters + CTR_SPILL, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[CTR_FOREIGN - CTR_SPILL] < (unsigned long long)FOREIGN_CAPACITY ? h[CTR_FOREIGN - CTR_SPILL] : FOREIGN_CAPACITY;
    if (h[1]) hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, c->stream, c->d_counters + CTR_DROPPED, 1);  // reported: start counting anew
    if (h[ptd::CTR_POOL_ABORT - CTR_SPILL] != 0)
        return fail(PTGPU_E_INTERNAL, "the pooled integrator (PTGPU_PT_POOL=1) gave a launch up after a wave waited 2^%d ticks: the pass's image is incomplete",
                    env_int("PTGPU_POOL_WATCHDOG_LOG2", ptd::POOL_WATCHDOG_LOG2) & 63);
    return PTGPU_OK;
}

int ptgpu_ensure_spill_lists(ptgpu_context* c) {
    if (!c->d_foreign) HIP_TRY(hipMalloc(&c->d_foreign, sizeof(ForeignSpill) * FOREIGN_CAPACITY));
    return PTGPU_OK;
}

extern "C" int ptgpu_render_pass(ptgpu_scene* s, const ptgpu_params* prm, const ptgpu_camera* cam, const ptgpu_light* lights, int32_t num_lights,
                                 int32_t block_begin, int32_t block_end, int32_t block_stride, float* accum) {
    if (!s || !prm || !accum) return fail(PTGPU_E_INVALID, "NULL argument");
    int rc = check_params(prm);
    if (rc != PTGPU_OK) return rc;
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)prm->width * prm->height * 16;
    rc = ensure_stage(c, bytes);
    if (rc != PTGPU_OK) return rc;
    float* d = reinterpret_cast<float*>(c->d_stage);
    HIP_TRY(hipMemcpyAsync(d, accum, bytes, hipMemcpyHostToDevice, c->stream));
    rc = ptgpu_render_pass_dev(s, prm, cam, lights, num_lights, block_begin, block_end, block_stride, d);
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipMemcpyAsync(accum, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long h[4];  // CTR_DROPPED .. CTR_POOL_ABORT
    static_assert(ptd::CTR_POOL_ABORT == CTR_DROPPED + 3, "one copy covers both words");
    HIP_TRY(hipMemcpy(h, c->d_counters + CTR_DROPPED, sizeof h, hipMemcpyDeviceToHost));
    if (h[3]) return fail(PTGPU_E_INTERNAL, "the pooled integrator (PTGPU_PT_POOL=1) gave a launch up (watchdog): the pass's image is incomplete");
    if (h[0]) {
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, c->stream, c->d_counters + CTR_DROPPED, 1);
        return fail(PTGPU_E_UNSUPPORTED, "%llu out-of-block samples did not fit the spill lists and were dropped (more than %d waiting for one tile exchange?)",
                    h[0], FOREIGN_CAPACITY);
    }
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-sample entry points: the reference's static PathTraceRenderer functions (src/PathTraceRenderer.hpp:97-102) for arbitrary
// arguments, traced / evaluated by the SAME device code the render pass runs (ptd::integrate, get_texture_parameters,
// get_light_to_sample) -- so tracePath is checked sample by sample, not only through pixel sums.
// ------------------------------------------------------------------------------------------------
extern "C" int ptgpu_trace_samples_dev(ptgpu_scene* s, const ptgpu_params* prm, const ptgpu_camera* cam, const ptgpu_light* lights, int32_t num_lights,
                                       int64_t n, const float* d_xy, const int32_t* d_sampler_base, const uint32_t* d_rng_state, float* d_out) {
    if (!s || !prm || !cam || !lights) return fail(PTGPU_E_INVALID, "NULL argument");
    int rc = check_params(prm);
    if (rc != PTGPU_OK) return rc;
    if (num_lights < 1 || num_lights > 1024) return fail(PTGPU_E_INVALID, "num_lights %d out of range [1,1024]", num_lights);
    if (n < 0 || n > (1LL << 30)) return fail(PTGPU_E_INVALID, "n = %lld out of range [0, 2^30]", (long long)n);
    if (n == 0) return PTGPU_OK;
    if (!d_xy || !d_sampler_base || !d_rng_state || !d_out) return fail(PTGPU_E_INVALID, "NULL device array");
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    rc = upload_lights(c, lights, num_lights);
    if (rc != PTGPU_OK) return rc;
    ptd::TravScene trav = s->trav;
    trav.alpha_on = (prm->use_alpha && s->has_alpha) ? 1 : 0;
    ptd::ShadeScene ss;
    ss.pos = s->d_pos; ss.nrm = s->d_nrm; ss.uv = s->d_uv; ss.materials = s->d_materials; ss.textures = s->d_textures;
    ss.sobol_pairs = c->d_sobol_pairs;
    ptd::PassDesc pd{};
    pd.block_stride = 1; pd.blocks_x = (prm->width + 31) / 32; pd.samples_per_block = 1024 * prm->num_aa_rays; pd.num_lights = num_lights;
    pd.inv_prob = 1.0f / prm->termination_prob;
    pd.ext_xy = d_xy; pd.ext_base = d_sampler_base; pd.ext_rng = d_rng_state; pd.num_ext = (uint32_t)n;
    // one configuration that is valid for every tree depth (16 LDS stack entries + global overflow)
    rc = trav.alpha_on ? launch_integrate<false, 16, true, true, true>(s, trav, ss, prm, cam, pd, reinterpret_cast<float4*>(d_out))
                       : launch_integrate<false, 16, true, false, true>(s, trav, ss, prm, cam, pd, reinterpret_cast<float4*>(d_out));
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

extern "C" int ptgpu_trace_samples(ptgpu_scene* s, const ptgpu_params* prm, const ptgpu_camera* cam, const ptgpu_light* lights, int32_t num_lights,
                                   int64_t n, const float* xy, const int32_t* sampler_base, const uint32_t* rng_state, float* out) {
    if (!s) return fail(PTGPU_E_INVALID, "scene is NULL");
    if (n < 0 || n > (1LL << 30)) return fail(PTGPU_E_INVALID, "n = %lld out of range [0, 2^30]", (long long)n);
    if (n == 0) return PTGPU_OK;
    if (!xy || !sampler_base || !rng_state || !out) return fail(PTGPU_E_INVALID, "NULL array");
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)n, b8 = (N * 8 + 255) & ~(size_t)255, b4 = (N * 4 + 255) & ~(size_t)255;
    int rc = ensure_stage(c, 2 * b8 + b4 + N * 16);
    if (rc != PTGPU_OK) return rc;
    unsigned char* b = c->d_stage;
    float* d_xy = (float*)b;
    uint32_t* d_rng = (uint32_t*)(b + b8);
    int32_t* d_base = (int32_t*)(b + 2 * b8);
    float* d_out = (float*)(b + 2 * b8 + b4);
    HIP_TRY(hipMemcpyAsync(d_xy, xy, N * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_rng, rng_state, N * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_base, sampler_base, N * 4, hipMemcpyHostToDevice, c->stream));
    rc = ptgpu_trace_samples_dev(s, prm, cam, lights, num_lights, n, d_xy, d_base, d_rng, d_out);
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, N * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}

// getTextureParameters (reference src/PathTraceRenderer.cpp:75-181) for n hits (triangle, u, v)
__global__ void k_texture_parameters(ptd::ShadeScene SS, ptgpu_params prm, const int32_t* __restrict__ mat_id, int num_tris, long long n,
                                     const int32_t* __restrict__ tri, const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ out9) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = tri[i];
    ptd::F3 diffuse = ptd::f3(0, 0, 0), nrm = diffuse, specular = diffuse;
    if (t >= 0 && t < num_tris) ptd::get_texture_parameters(SS, prm, t, mat_id[t], u[i], v[i], diffuse, nrm, specular);
    float* o = out9 + i * 9;
    o[0] = diffuse.x; o[1] = diffuse.y; o[2] = diffuse.z; o[3] = nrm.x; o[4] = nrm.y; o[5] = nrm.z; o[6] = specular.x; o[7] = specular.y; o[8] = specular.z;
}

// getLightToSample (reference src/PathTraceRenderer.cpp:183-210) for n positions, each with its own random-stream state
__global__ void k_light_to_sample(const ptgpu_light* __restrict__ lights, int num_lights, long long n, const float* __restrict__ pos,
                                  const uint32_t* __restrict__ rng_state, int32_t* __restrict__ out2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ptd::Rng R;
    R.key = rng_state[2 * i]; R.ctr = rng_state[2 * i + 1];
    out2[2 * i] = ptd::get_light_to_sample(ptd::ld3(pos + 3 * i), lights, num_lights, R);
    out2[2 * i + 1] = (int32_t)R.ctr;
}

// sobol::sample(index, dimension) (reference src/sobol.h:39-49) for n (index, dimension) pairs, through the device functions the kernels
// draw their samples with: the byte-indexed pair table + bit loop of sobol32x2 for indices below 2^32 (every index the kernels form), the
// reference's own 52-step loop beyond.
__global__ void k_sobol_sample(const uint2* __restrict__ pairs, long long n, const unsigned long long* __restrict__ index,
                               const uint32_t* __restrict__ dimension, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long idx = index[i];
    const uint32_t dim = dimension[i];
    if (idx >> 32) {
        uint32_t result = 0u;
        for (unsigned k = dim * 52u; idx; idx >>= 1, ++k)
            if (idx & 1ull) result ^= ptd::g_sobol_matrices[k];
        out[i] = (float)result * (1.f / 4294967296.f);
    } else {
        float a, b;
        ptd::sobol32x2(pairs, (uint32_t)idx, dim & ~1u, a, b);
        out[i] = (dim & 1u) ? b : a;
    }
}

extern "C" int ptgpu_sobol_sample_batch(ptgpu_context* c, int64_t n, const uint64_t* index, const uint32_t* dimension, float* out) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    if (n < 0 || n > (1LL << 28)) return fail(PTGPU_E_INVALID, "n = %lld out of range", (long long)n);
    if (n == 0) return PTGPU_OK;
    if (!index || !dimension || !out) return fail(PTGPU_E_INVALID, "NULL array");
    for (int64_t i = 0; i < n; ++i)
        if (dimension[i] >= 1024u) return fail(PTGPU_E_INVALID, "dimension[%lld] = %u: the table has 1024 dimensions (sobol.h:40 asserts)", (long long)i, dimension[i]);
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)n, b8 = (N * 8 + 255) & ~(size_t)255, b4 = (N * 4 + 255) & ~(size_t)255;
    int rc = ensure_stage(c, b8 + 2 * b4);
    if (rc != PTGPU_OK) return rc;
    unsigned char* b = c->d_stage;
    HIP_TRY(hipMemcpyAsync(b, index, N * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b + b8, dimension, N * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_sobol_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_sobol_pairs, (long long)n,
                       (const unsigned long long*)b, (const uint32_t*)(b + b8), (float*)(b + b8 + b4));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, b + b8 + b4, N * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}

extern "C" int ptgpu_texture_parameters_batch(ptgpu_scene* s, const ptgpu_params* prm, int64_t n, const int32_t* tri, const float* u, const float* v,
                                              float* out_diffuse_normal_specular) {
    if (!s || !prm) return fail(PTGPU_E_INVALID, "NULL argument");
    if (n < 0 || n > (1LL << 28)) return fail(PTGPU_E_INVALID, "n = %lld out of range", (long long)n);
    if (n == 0) return PTGPU_OK;
    if (!tri || !u || !v || !out_diffuse_normal_specular) return fail(PTGPU_E_INVALID, "NULL array");
    ptgpu_context* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)n, b4 = (N * 4 + 255) & ~(size_t)255;
    int rc = ensure_stage(c, 3 * b4 + N * 36);
    if (rc != PTGPU_OK) return rc;
    unsigned char* b = c->d_stage;
    HIP_TRY(hipMemcpyAsync(b, tri, N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b + b4, u, N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b + 2 * b4, v, N * 4, hipMemcpyHostToDevice, c->stream));
    ptd::ShadeScene ss;
    ss.pos = s->d_pos; ss.nrm = s->d_nrm; ss.uv = s->d_uv; ss.materials = s->d_materials; ss.textures = s->d_textures; ss.sobol_pairs = c->d_sobol_pairs;
    hipLaunchKernelGGL(k_texture_parameters, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ss, *prm, s->d_mat_id, s->num_tris, (long long)n,
                       (const int32_t*)b, (const float*)(b + b4), (const float*)(b + 2 * b4), (float*)(b + 3 * b4));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_diffuse_normal_specular, b + 3 * b4, N * 36, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}

extern "C" int ptgpu_light_to_sample_batch(ptgpu_context* c, const ptgpu_light* lights, int32_t num_lights, int64_t n, const float* pos,
                                           const uint32_t* rng_state, int32_t* out_index_and_counter) {
    if (!c || !lights) return fail(PTGPU_E_INVALID, "NULL argument");
    if (num_lights < 1 || num_lights > 1024) return fail(PTGPU_E_INVALID, "num_lights %d out of range [1,1024]", num_lights);
    if (n < 0 || n > (1LL << 28)) return fail(PTGPU_E_INVALID, "n = %lld out of range", (long long)n);
    if (n == 0) return PTGPU_OK;
    if (!pos || !rng_state || !out_index_and_counter) return fail(PTGPU_E_INVALID, "NULL array");
    HIP_TRY(hipSetDevice(c->device));
    int rc = upload_lights(c, lights, num_lights);
    if (rc != PTGPU_OK) return rc;
    const size_t N = (size_t)n, b12 = (N * 12 + 255) & ~(size_t)255, b8 = (N * 8 + 255) & ~(size_t)255;
    rc = ensure_stage(c, b12 + 2 * b8);
    if (rc != PTGPU_OK) return rc;
    unsigned char* b = c->d_stage;
    HIP_TRY(hipMemcpyAsync(b, pos, N * 12, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(b + b12, rng_state, N * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_light_to_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_lights, (int)num_lights, (long long)n,
                       (const float*)b, (const uint32_t*)(b + b12), (int32_t*)(b + b12 + b8));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_index_and_counter, b + b12 + b8, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}

extern "C" int ptgpu_resolve_dev(ptgpu_context* c, const float* d_accum, int32_t width, int32_t height, float* d_out_rgba) {
    if (!c || !d_accum || !d_out_rgba) return fail(PTGPU_E_INVALID, "NULL argument");
    if (width <= 0 || height <= 0) return fail(PTGPU_E_INVALID, "image size %dx%d", width, height);
    HIP_TRY(hipSetDevice(c->device));
    const long long n = (long long)width * height;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, reinterpret_cast<const float4*>(d_accum), n,
                       reinterpret_cast<float4*>(d_out_rgba));
    HIP_TRY(hipGetLastError());
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// rtIntersect.inl intersect_triangle1 (reference src/rtIntersect.inl:215-259): two-sided Moller-Trumbore on raw vertex
// positions.  Dead code in the reference (included by RayTracer.cpp:7, never called -- the live test is Woop),
// provided as a batch kernel so that the test north_star names exists on the device, pinned bit-exactly on goldens produced by
// the reference's own source (tests/golden/rtintersect_golden.json).
// ------------------------------------------------------------------------------------------------
__global__ void k_intersect_triangle1(const float* __restrict__ orig, const float* __restrict__ dir, const float* __restrict__ v0,
                                      const float* __restrict__ v1, const float* __restrict__ v2, long long n, int32_t* __restrict__ hit,
                                      float* __restrict__ ot, float* __restrict__ ou, float* __restrict__ ov) {
    using namespace ptd;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float EPSILON = 0.000000001f;
    F3 o = ld3(orig + i * 3), d = ld3(dir + i * 3), a = ld3(v0 + i * 3);
    F3 edge1 = ld3(v1 + i * 3) - a, edge2 = ld3(v2 + i * 3) - a;
    F3 pvec = cross(d, edge2);
    float det = dot(edge1, pvec);
    float u = 0.0f, v = 0.0f, t = 0.0f;
    F3 tvec, qvec = f3(0, 0, 0);
    bool ok = true;
    if (det > EPSILON) {
        tvec = o - a;
        u = dot(tvec, pvec);
        if (u < 0.0f || u > det) ok = false;
        if (ok) {
            qvec = cross(tvec, edge1);
            v = dot(d, qvec);
            if (v < 0.0f || u + v > det) ok = false;
        }
    } else if (det < -EPSILON) {
        tvec = o - a;
        u = dot(tvec, pvec);
        if (u > 0.0f || u < det) ok = false;
        if (ok) {
            qvec = cross(tvec, edge1);
            v = dot(d, qvec);
            if (v > 0.0f || u + v < det) ok = false;
        }
    } else
        ok = false;
    if (ok) {
        float inv_det = 1.f / det;
        t = dot(edge2, qvec) * inv_det;
        u *= inv_det;
        v *= inv_det;
    }
    hit[i] = ok ? 1 : 0;
    ot[i] = ok ? t : 0.0f; ou[i] = ok ? u : 0.0f; ov[i] = ok ? v : 0.0f;
}

extern "C" int ptgpu_intersect_triangle1_batch(ptgpu_context* c, const float* orig, const float* dir, const float* v0, const float* v1,
                                               const float* v2, int64_t n, int32_t* out_hit, float* out_t, float* out_u, float* out_v) {
    if (!c) return fail(PTGPU_E_INVALID, "ctx is NULL");
    if (n < 0) return fail(PTGPU_E_INVALID, "n < 0");
    if (n == 0) return PTGPU_OK;
    if (!orig || !dir || !v0 || !v1 || !v2 || !out_hit || !out_t || !out_u || !out_v) return fail(PTGPU_E_INVALID, "NULL array");
    HIP_TRY(hipSetDevice(c->device));
    float* d_in = nullptr;
    float* d_out = nullptr;
    hipError_t e = hipMalloc(&d_in, (size_t)n * 15 * 4);
    if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * 4 * 4);
    const float* src[5] = {orig, dir, v0, v1, v2};
    for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipMemcpyAsync(d_in + (size_t)k * n * 3, src[k], (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_intersect_triangle1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, d_in + n * 3, d_in + n * 6,
                           d_in + n * 9, d_in + n * 12, (long long)n, reinterpret_cast<int32_t*>(d_out), d_out + n, d_out + 2 * n, d_out + 3 * n);
        e = hipGetLastError();
    }
    void* dst[4] = {out_hit, out_t, out_u, out_v};
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipMemcpyAsync(dst[k], d_out + (size_t)k * n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_in) hipFree(d_in);
    if (d_out) hipFree(d_out);
    if (e != hipSuccess) return fail(PTGPU_E_HIP, "intersect_triangle1 batch: %s", hipGetErrorString(e));
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side BVH builder (pt_lbvh.hip.h): host arrays in, reference-layout tree out
// ------------------------------------------------------------------------------------------------
namespace {
// what the device builder leaves on the device (freed by release())
struct LbvhOnDevice {
    std::vector<void*> owned;
    ptgpu_flatnode* d_nodes = nullptr;  // reference layout, pre-order
    int32_t* d_list = nullptr;          // triangle index per list entry
    int32_t num_nodes = 0, depth = 0;
    void release() { for (void* p : owned) hipFree(p); owned.clear(); }
};
}  // namespace

// num_tris >= 2.  The triangle positions come from the host (`pos`) or, if d_pos_ready is not NULL, are on the device already.
static int lbvh_build_on_device(ptgpu_context* c, const float* pos, const float* d_pos_ready, int n, int max_tris_in_leaf, LbvhOnDevice& R) {
    hipStream_t st = c->stream;
    auto cleanup = [&]() { R.release(); };
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
        R.owned.push_back(p);
        return p;
    };
#define LB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(PTGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
#define LB_ALLOC(var, type, count) type* var = (type*)dalloc((size_t)(count) * sizeof(type)); if (!var) { cleanup(); return fail(PTGPU_E_HIP, "hipMalloc failed in the BVH builder"); }
    const int n_int = n - 1;
    const float* d_pos = d_pos_ready;
    if (!d_pos) {
        LB_ALLOC(d_pos_own, float, (size_t)n * 9)
        LB_TRY(hipMemcpyAsync(d_pos_own, pos, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice, st));
        d_pos = d_pos_own;
    }
    LB_ALLOC(d_lo, float4, n)
    LB_ALLOC(d_hi, float4, n)
    LB_ALLOC(d_bounds, uint32_t, 8)
    LB_ALLOC(d_keys, unsigned long long, n)
    LB_ALLOC(d_keys2, unsigned long long, n)
    LB_ALLOC(d_vals, int32_t, n)
    LB_ALLOC(d_vals2, int32_t, n)
    LB_ALLOC(d_out, int32_t, 4)
    const uint32_t h_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    LB_TRY(hipMemcpyAsync(d_bounds, h_bounds, sizeof h_bounds, hipMemcpyHostToDevice, st));
    LB_TRY(hipMemsetAsync(d_out, 0, 4 * sizeof(int32_t), st));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ptl::k_lbvh_prims, dim3(blocks), dim3(256), 0, st, d_pos, n, d_lo, d_hi, d_bounds);
    hipLaunchKernelGGL(ptl::k_lbvh_morton, dim3(blocks), dim3(256), 0, st, d_lo, d_hi, n, d_bounds, d_keys, d_vals);
    size_t temp_bytes = 0;
    LB_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0u, 63u, st));
    LB_ALLOC(d_temp, char, temp_bytes)
    LB_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0u, 63u, st));
    uint32_t hb[8];
    LB_TRY(hipMemcpyAsync(hb, d_bounds, sizeof hb, hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    if (hb[6]) { cleanup(); return fail(PTGPU_E_INVALID, "a triangle has a vertex that is not finite"); }
    LB_ALLOC(d_child, int2, n_int)
    LB_ALLOC(d_range, int2, n_int)
    LB_ALLOC(d_parent, int32_t, n_int)
    LB_ALLOC(d_leaf_parent, int32_t, n)
    LB_ALLOC(d_box_lo, float4, n_int)
    LB_ALLOC(d_box_hi, float4, n_int)
    LB_ALLOC(d_size, int32_t, n_int)
    LB_ALLOC(d_arrived, uint32_t, n_int)
    LB_ALLOC(d_nodes, ptgpu_flatnode, (size_t)2 * n)
    LB_TRY(hipMemsetAsync(d_arrived, 0, (size_t)n_int * sizeof(uint32_t), st));
    ptl::Tree T;
    T.n = n; T.keys = d_keys2; T.child = d_child; T.range = d_range; T.parent = d_parent; T.leaf_parent = d_leaf_parent;
    T.box_lo = d_box_lo; T.box_hi = d_box_hi; T.size = d_size; T.arrived = d_arrived;
    hipLaunchKernelGGL(ptl::k_lbvh_hierarchy, dim3((unsigned)((n_int + 255) / 256)), dim3(256), 0, st, T);
    hipLaunchKernelGGL(ptl::k_lbvh_fit, dim3(blocks), dim3(256), 0, st, T, d_vals2, d_lo, d_hi, (int)max_tris_in_leaf);
    hipLaunchKernelGGL(ptl::k_lbvh_emit, dim3((unsigned)(((size_t)n_int + n + 255) / 256)), dim3(256), 0, st, T, d_vals2, d_lo, d_hi, (int)max_tris_in_leaf,
                       d_nodes, d_out);
    LB_TRY(hipGetLastError());
    int32_t h_size = 0, h_out[4];
    LB_TRY(hipMemcpyAsync(&h_size, d_size, sizeof(int32_t), hipMemcpyDeviceToHost, st));  // size of the root's subtree
    LB_TRY(hipMemcpyAsync(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    if (h_size < 1 || h_size > 2 * n) { cleanup(); return fail(PTGPU_E_HIP, "BVH builder produced %d nodes for %d triangles", h_size, n); }
    R.d_nodes = d_nodes; R.d_list = d_vals2; R.num_nodes = h_size; R.depth = h_out[0];
#undef LB_TRY
#undef LB_ALLOC
    return PTGPU_OK;
}

extern "C" int ptgpu_bvh_build_lbvh(ptgpu_context* c, const float* pos, int32_t num_tris, int32_t max_tris_in_leaf, ptgpu_flatnode* out_nodes,
                                    int32_t* out_num_nodes, int32_t* out_list, int32_t* out_depth) {
    if (!c || !out_num_nodes) return fail(PTGPU_E_INVALID, "NULL argument");
    if (num_tris < 0) return fail(PTGPU_E_INVALID, "num_tris < 0");
    if (max_tris_in_leaf < 1) return fail(PTGPU_E_INVALID, "max_tris_in_leaf %d < 1", max_tris_in_leaf);
    if (num_tris > 0 && (!pos || !out_nodes || !out_list)) return fail(PTGPU_E_INVALID, "NULL array");
    *out_num_nodes = 0;
    if (out_depth) *out_depth = 0;
    if (num_tris == 0) return PTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (num_tris == 1) {
        for (int k = 0; k < 9; ++k)
            if (!(std::fabs(pos[k]) <= 3.402823466e+38f)) return fail(PTGPU_E_INVALID, "a triangle has a vertex that is not finite");
        ptgpu_flatnode f;
        memset(&f, 0, sizeof f);
        for (int k = 0; k < 3; ++k) {
            f.bmin[k] = std::min(pos[k], std::min(pos[3 + k], pos[6 + k]));
            f.bmax[k] = std::max(pos[k], std::max(pos[3 + k], pos[6 + k]));
        }
        f.leaf = 1; f.triangles = 1;
        out_nodes[0] = f;
        out_list[0] = 0;
        *out_num_nodes = 1;
        if (out_depth) *out_depth = 1;
        return PTGPU_OK;
    }
    LbvhOnDevice R;
    int rc = lbvh_build_on_device(c, pos, nullptr, num_tris, max_tris_in_leaf, R);
    if (rc != PTGPU_OK) return rc;
    hipError_t e = hipMemcpy(out_list, R.d_list, (size_t)num_tris * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_nodes, R.d_nodes, (size_t)R.num_nodes * sizeof(ptgpu_flatnode), hipMemcpyDeviceToHost);
    *out_num_nodes = R.num_nodes;
    if (out_depth) *out_depth = R.depth;
    R.release();
    if (e != hipSuccess) return fail(PTGPU_E_HIP, "copying the built tree to the host failed: %s", hipGetErrorString(e));
    return PTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// Scene straight from the device-built tree: the FlatNode array and the index list never visit the host.  The traversal records are
// written by two kernels from the same fields build_scene() reads on the host, so the scene equals the one ptgpu_scene_create makes from
// ptgpu_bvh_build_lbvh's output, record for record.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scene_inner_flags(const ptgpu_flatnode* __restrict__ nodes, int nn, int32_t* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nn) flag[i] = nodes[i].leaf ? 0 : 1;
}
// records = base + list_size records: interior node with interior index k at records[k], triangle record p at records[~p]
__global__ __launch_bounds__(256) void k_scene_records(const ptgpu_flatnode* __restrict__ nodes, int nn, const int32_t* __restrict__ inner_index,
                                                       const int32_t* __restrict__ list, const float* __restrict__ woop,
                                                       const int32_t* __restrict__ mat_id, const ptgpu_material* __restrict__ materials,
                                                       float4* __restrict__ records) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const ptgpu_flatnode n = nodes[i];
    if (n.leaf) {
        for (int p = n.startIdx; p <= n.endIdx; ++p) {
            const int32_t id = list[p], remaining = n.endIdx - p, mat = mat_id[id], atex = materials[mat].tex_alpha;
            const float* w = woop + (size_t)id * 12;
            float4* o = records + (ptrdiff_t)(~p) * 4;
            o[0] = make_float4(w[0], w[1], w[2], w[3]);
            o[1] = make_float4(w[4], w[5], w[6], w[7]);
            o[2] = make_float4(w[8], w[9], w[10], w[11]);
            o[3] = make_float4(__int_as_float(id), __int_as_float(remaining), __int_as_float(atex), __int_as_float(mat));
        }
    } else {
        const ptgpu_flatnode L = nodes[i + 1], R = nodes[n.rightChild];
        const int refL = L.leaf ? ~L.startIdx : inner_index[i + 1], refR = R.leaf ? ~R.startIdx : inner_index[n.rightChild];
        float4* o = records + (ptrdiff_t)inner_index[i] * 4;
        o[0] = make_float4(L.bmin[0], L.bmin[1], L.bmax[0], L.bmax[1]);
        o[1] = make_float4(R.bmin[0], R.bmin[1], R.bmax[0], R.bmax[1]);
        o[2] = make_float4(L.bmin[2], L.bmax[2], R.bmin[2], R.bmax[2]);
        o[3] = make_float4(__int_as_float(refL), __int_as_float(refR), __int_as_float(n.axis), 0.0f);
    }
}

static int build_scene_lbvh(ptgpu_context* c, const ptgpu_scene_desc* d, int max_tris_in_leaf, ptgpu_scene* s) {
    const int nt = d->num_tris;
    s->num_tris = nt;
    s->num_materials = d->num_materials;
    s->num_textures = d->num_textures;
    int rc = upload_shading_arrays(c, d, s);
    if (rc != PTGPU_OK) return rc;
    for (int i = 0; i < nt && !s->has_alpha; ++i)
        if (d->materials[d->mat_id[i]].tex_alpha >= 0) s->has_alpha = true;
    LbvhOnDevice R;
    rc = lbvh_build_on_device(c, nullptr, s->d_pos, nt, max_tris_in_leaf, R);
    if (rc != PTGPU_OK) return rc;
    if (R.depth > 64) { R.release(); return fail(PTGPU_E_UNSUPPORTED, "tree depth %d exceeds 64 (the reference's traversal stack, RayTracer.cpp:190)", R.depth); }
    const int nn = R.num_nodes;
    hipStream_t st = c->stream;
    int32_t *d_flag = nullptr, *d_index = nullptr;
    float* d_woop = nullptr;
    void* d_temp = nullptr;
    auto fail_hip = [&](const char* what, hipError_t e) {
        if (d_flag) hipFree(d_flag);
        if (d_index) hipFree(d_index);
        if (d_woop) hipFree(d_woop);
        if (d_temp) hipFree(d_temp);
        R.release();
        return fail(PTGPU_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc(&d_flag, (size_t)nn * sizeof(int32_t))) != hipSuccess) return fail_hip("hipMalloc", e);
    if ((e = hipMalloc(&d_index, (size_t)nn * sizeof(int32_t))) != hipSuccess) return fail_hip("hipMalloc", e);
    if ((e = hipMalloc(&d_woop, (size_t)nt * 12 * sizeof(float))) != hipSuccess) return fail_hip("hipMalloc", e);
    if ((e = hipMemcpyAsync(d_woop, d->woop, (size_t)nt * 12 * sizeof(float), hipMemcpyHostToDevice, st)) != hipSuccess) return fail_hip("hipMemcpyAsync", e);
    const unsigned nb = (unsigned)((nn + 255) / 256);
    hipLaunchKernelGGL(k_scene_inner_flags, dim3(nb), dim3(256), 0, st, R.d_nodes, nn, d_flag);
    size_t temp_bytes = 0;
    if ((e = rocprim::exclusive_scan(nullptr, temp_bytes, d_flag, d_index, 0, (size_t)nn, rocprim::plus<int32_t>(), st)) != hipSuccess) return fail_hip("exclusive_scan", e);
    if ((e = hipMalloc(&d_temp, temp_bytes ? temp_bytes : 4)) != hipSuccess) return fail_hip("hipMalloc", e);
    if ((e = rocprim::exclusive_scan(d_temp, temp_bytes, d_flag, d_index, 0, (size_t)nn, rocprim::plus<int32_t>(), st)) != hipSuccess) return fail_hip("exclusive_scan", e);
    // interior count = index of the last node + its flag; the root's box and kind
    int32_t last[2] = {0, 0};
    ptgpu_flatnode root;
    if ((e = hipMemcpyAsync(&last[0], d_index + (nn - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return fail_hip("hipMemcpyAsync", e);
    if ((e = hipMemcpyAsync(&last[1], d_flag + (nn - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return fail_hip("hipMemcpyAsync", e);
    if ((e = hipMemcpyAsync(&root, R.d_nodes, sizeof root, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail_hip("hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail_hip("hipStreamSynchronize", e);
    const int n_inner = last[0] + last[1], nl = nt;
    if ((e = hipMalloc((void**)&s->d_records, ((size_t)nl + n_inner) * 64)) != hipSuccess) return fail_hip("hipMalloc", e);
    s->trav.records = s->d_records + (size_t)nl * 4;
    hipLaunchKernelGGL(k_scene_records, dim3(nb), dim3(256), 0, st, R.d_nodes, nn, d_index, R.d_list, d_woop, s->d_mat_id, s->d_materials,
                       const_cast<float4*>(s->trav.records));
    if ((e = hipGetLastError()) != hipSuccess) return fail_hip("k_scene_records", e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail_hip("hipStreamSynchronize", e);
    for (int k = 0; k < 3; ++k) { s->trav.root_bmin[k] = root.bmin[k]; s->trav.root_bmax[k] = root.bmax[k]; }
    s->trav.root_ref = root.leaf ? ~root.startIdx : 0;
    s->trav.list_size = nl;
    s->num_inner = n_inner;
    s->depth = R.depth;
    s->num_nodes_built = nn;
    s->bytes[0] += (size_t)n_inner * 64;
    s->bytes[1] += (size_t)nl * 64;
    hipFree(d_flag); hipFree(d_index); hipFree(d_woop); hipFree(d_temp);
    R.release();
    return PTGPU_OK;
}

extern "C" int ptgpu_scene_create_lbvh(ptgpu_context* c, const ptgpu_scene_desc* d, int32_t max_tris_in_leaf, ptgpu_scene** out, int32_t* out_num_nodes) {
    if (!c || !d || !out) return fail(PTGPU_E_INVALID, "NULL argument");
    if (d->num_nodes != 0 || d->list_size != 0 || d->nodes || d->list)
        return fail(PTGPU_E_INVALID, "ptgpu_scene_create_lbvh builds the hierarchy itself: nodes / list of the description must be empty");
    if (max_tris_in_leaf < 1) return fail(PTGPU_E_INVALID, "max_tris_in_leaf %d < 1", max_tris_in_leaf);
    int rc = check_scene_arrays(d);
    if (rc != PTGPU_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (d->num_tris < 2) {  // nothing to build on the device: the host path with the one-node (or empty) tree
        ptgpu_scene_desc d2 = *d;
        ptgpu_flatnode node;
        int32_t list0 = 0, nn = 0, depth = 0;
        rc = ptgpu_bvh_build_lbvh(c, d->pos, d->num_tris, max_tris_in_leaf, &node, &nn, &list0, &depth);
        if (rc != PTGPU_OK) return rc;
        d2.num_nodes = nn; d2.nodes = nn ? &node : nullptr; d2.list_size = nn ? 1 : 0; d2.list = nn ? &list0 : nullptr;
        if (out_num_nodes) *out_num_nodes = nn;
        return ptgpu_scene_create(c, &d2, out);
    }
    ptgpu_scene* s = new ptgpu_scene();
    s->ctx = c;
    rc = build_scene_lbvh(c, d, max_tris_in_leaf, s);
    if (rc != PTGPU_OK) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        ptgpu_scene_destroy(s);
        memcpy(g_err, keep, sizeof keep);
        return rc;
    }
    if (out_num_nodes) *out_num_nodes = s->num_nodes_built;
    *out = s;
    return PTGPU_OK;
}
