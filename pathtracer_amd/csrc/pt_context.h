// Library-internal state shared by the translation units of libptgpu.so (ptgpu.hip, pt_multigpu.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <vector>

#include "../../include/ptgpu.h"

// sets the thread-local message behind ptgpu_last_error() and returns `code`
int ptgpu_vfail(int code, const char* fmt, va_list ap);
__attribute__((format(printf, 2, 3))) static inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    const int rc = ptgpu_vfail(code, fmt, ap);
    va_end(ap);
    return rc;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PTGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Counter words of a context (unsigned long long each):
//   [0] rays  [1] node visits  [2] triangle tests                      the public counters (ptgpu_get_counters)
//   [3] entries in the spill list of the pass in flight    [4] samples dropped because a list was full (last pass)
//   [5] deferred rays of the raycast launch in flight      [6] entries pending in the foreign-spill list
//   [8..28] diagnostic stamps (PT_STAMPS builds)
constexpr int CTR_WORDS = 32;
// Short kernels with one wave per workgroup count their rays per wave; tens of thousands of atomics on ONE word serialise (measured:
// 32,400 same-address adds took 0.43 ms, three times the kernel), so those go to word CTR_WORDS + (workgroup & (RAY_SPREAD - 1)) and
// ptgpu_get_counters folds the spread words into the ray count.
constexpr int RAY_SPREAD = 1024;
constexpr int CTR_SPILL = 3, CTR_DROPPED = 4, CTR_DEFER = 5, CTR_FOREIGN = 6;
// Out-of-block samples (Sobol-block mode): a sample position bx + bw * sobol() can round up to bx + bw, one pixel outside its block --
// at pixel coordinates in the thousands that takes sobol() > 1 - 4e-6, so a 4K frame at 1024 spp has some 30,000 of them per pass.
// The reference adds them to the neighbouring block's pixel while another thread works on that block (a race); here they are kept in
// a list and added after the pass's block sums, per pixel in (source block, sample) order -- the order of the sequential CPU loop.
// One list entry (also the wire format of the tile exchange, pt_multigpu.hip):
struct ForeignSpill {
    float r, g, b, w;          // what the sample adds to its pixel, already divided by NumAaRays
    int32_t pixel;             // y * width + x
    int32_t src_block, sample; // the block that traced it and its index i inside that block's sample loop: the sort key
    int32_t pad;
};
static_assert(sizeof(ForeignSpill) == 32, "two float4");
// Entries per rank and exchange that wait for ptgpu_tiles_pack_dev: with tiles sharded over ranks EVERY out-of-block sample travels (so
// that all ranks add all of them in one canonical order); 2 MiB of each rank's slice.  configs[3] (4K, 1024 spp) on two ranks: ~17,000.
constexpr int FOREIGN_CAPACITY = 65536;
// the pass's own list grows with the pass (host bound: samples * 2^-14 + 4096 entries, see spill_bound)
constexpr long long SPILL_CAPACITY_MAX = 1ll << 24;

// Several lists of ForeignSpill entries addressed as one array: entry (idx % per_list) of list (idx / per_list).
struct SpillLists {
    const float4* first;          // entry 0 of list 0
    long long list_stride_f4;     // float4s between two lists' first entries
    int32_t per_list, num_lists;  // capacity of one list; entries beyond a list's count are ignored
    // list r holds min(count, per_list) entries, count = *local_count (one list: a counter word) or the int in the float4 before its first entry
    const unsigned long long* local_count;
};
// which 32x32 blocks a render call accumulates (block_begin, block_begin + stride, ... < block_end)
struct TileOwner {
    int32_t block_begin, block_end, block_stride, blocks_x;
};
// ptgpu.hip: adds the lists' entries to d_accum, per pixel in (src_block, sample) order whatever the order of the lists (two stable
// radix sorts + one sequential walk per pixel).  own != NULL: only entries whose pixel lies in one of `own`'s blocks; the others are
// appended to the context's foreign list.  max_entries = per_list * num_lists.
int ptgpu_apply_spills_sorted(ptgpu_context* c, const SpillLists& lists, float* d_accum, int32_t width, int32_t height, const TileOwner* own);

struct ptgpu_comm;  // pt_multigpu.hip

struct ptgpu_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool count = false;
    unsigned long long* d_counters = nullptr;  // CTR_WORDS words, see above
    int num_cu = 256;
    // render workspace (grow-only): per-sample radiance of one batch of blocks, light table, spill lists
    float4* d_rad = nullptr;
    size_t rad_capacity = 0;  // samples
    ptgpu_light* d_lights = nullptr;
    std::vector<ptgpu_light> h_lights;  // what d_lights holds
    ForeignSpill* d_spill = nullptr;   // out-of-block samples of the pass in flight (grow-only, sized per pass)
    size_t spill_capacity = 0;
    ForeignSpill* d_foreign = nullptr; // FOREIGN_CAPACITY entries waiting for ptgpu_tiles_pack_dev
    unsigned char* d_sort_ws = nullptr;  // keys, indices and rocprim's temporary storage of ptgpu_apply_spills_sorted (grow-only)
    size_t sort_ws_bytes = 0;
    int32_t* d_defer = nullptr;    // indices of the rays a one-ray-per-lane launch gave up for lack of LDS stack
    size_t defer_capacity = 0;
    uint2* d_stack_ovf = nullptr;  // global-memory overflow of the per-lane traversal stacks (trees deeper than the LDS stack)
    size_t stack_ovf_bytes = 0;
    const uint32_t* d_sobol_matrices = nullptr;  // device address of ptd::g_sobol_matrices on this device
    uint2* d_sobol_pairs = nullptr;  // byte-indexed Sobol table for the integrator (pt_integrator.hip.h sobol32x2), built at init
    unsigned char* d_stage = nullptr;  // grow-only device staging of the host-pointer entry points (rays in, hit records out, images)
    size_t stage_bytes = 0;
    unsigned char* scratch = nullptr;  // THIS device's address of the file-scope scratch array (ptgpu_raycast_batch_dev_noalloc)
    ptgpu_comm* comm = nullptr;        // tile-gather communicator (ptgpu_comm_init), owned
    // ptgpu_cancel: d_cancel (the last word of d_counters) = number of the last cancelled render pass, stored by a one-wave kernel on
    // cancel_stream; the kernels of pass number pass_epoch compare it with their own number.
    uint32_t* d_cancel = nullptr;
    hipStream_t cancel_stream = nullptr;
    std::atomic<uint32_t> pass_epoch{0};
};

// pt_multigpu.hip
int ptgpu_ensure_spill_lists(ptgpu_context* c);
void ptgpu_comm_release(ptgpu_context* c);
