// Pooled form of the path-trace integrator (pt_integrator.hip.h): the same per-sample arithmetic, regrouped across the waves of a workgroup.
//
// In ptd::integrate a lane owns a path sample for its whole life, so while the sample waits for a service block (hit set-up, new sample)
// the lane idles through the wave's traversal steps: 37 of 64 lanes hold a running ray on the configs[2] frame.  Here the path state lives
// in LDS "slots", and the waves of one large workgroup specialise:
//   * TRACE waves keep only rays in registers (ptd::Trav, their stacks in LDS).  A lane whose ray ends settles it if that is cheap
//     (shadow ray: add the pending radiance, launch the pending bounce ray in place), otherwise hands the slot to a queue and takes the
//     next ray from the ray queue -- so nearly every lane of a traversal step holds a ray;
//   * SHADE waves take up to 64 slots of ONE kind from a queue (closest hits to shade; ended or empty slots that want a new sample)
//     and run that block with all lanes on it, then put the slots' next rays on the ray queue.
// Queues are rings of slot numbers in LDS, claimed with one LDS atomic per wave (ballot + prefix rank inside the wave).  A slot is in
// exactly one place at any time (a queue, a trace lane, a shade lane, or retired), so its state needs no lock; queue hand-over is
// release/acquire on the ring entry.  Which lane or wave touches a sample never enters its arithmetic (see pt_integrator.hip.h), so the
// per-sample radiance is bit-identical to ptd::integrate's.
// Covers tracePath (reference src/PathTraceRenderer.cpp:236-551) in the block modes; Whitted, the counting build and caller-given samples
// stay on ptd::integrate.
#pragma once
#include "pt_integrator.hip.h"

#ifndef PT_POOL_TW
#define PT_POOL_TW 12  // trace waves per workgroup
#endif
#ifndef PT_POOL_HW
#define PT_POOL_HW 4   // shade waves per workgroup
#endif
#ifndef PT_POOL_SLOTS
#define PT_POOL_SLOTS 1020
#endif
#ifndef PT_POOL_NLDS
#define PT_POOL_NLDS 8
#endif

// Diagnostic build only (-DPT_POOL_STATS): per-wave sums added into counters[8..]; see tools/pool_stats.py for the words.
#ifdef PT_POOL_STATS
#define PS_NOW() ((long long)__builtin_amdgcn_s_memtime())
#define PS_ADD(k, v) (ps_stat[k] += (long long)(v))
#else
#define PS_NOW() 0ll
#define PS_ADD(k, v) do {} while (0)
#endif

namespace ptd {

constexpr int POOL_TW = PT_POOL_TW, POOL_HW = PT_POOL_HW, POOL_SLOTS = PT_POOL_SLOTS, POOL_NLDS = PT_POOL_NLDS;
constexpr int POOL_THREADS = (POOL_TW + POOL_HW) * 64;
constexpr int POOL_QCAP = 1024;  // ring capacity, a power of two >= POOL_SLOTS
constexpr int POOL_CHUNK = 256;  // samples a workgroup draws from one place of the launch's range before it moves on
static_assert(POOL_SLOTS < 1024 && POOL_QCAP == 1024, "a ring entry is (slot + 1) in 10 bits | the lap of its place in 6 bits");
// ring entry for place `pos` of a queue (pos counts up for ever; place pos & 1023): the lap tag lets a reader that waits at place pos tell
// its own entry from the one written a lap earlier that another reader has yet to take
__device__ __forceinline__ uint16_t pool_entry(int slot, uint32_t pos) { return (uint16_t)((uint32_t)(slot + 1) | (((pos >> 10) & 63u) << 10)); }
__device__ __forceinline__ bool pool_entry_is(uint16_t e, uint32_t pos) { return e != 0 && (uint32_t)(e >> 10) == ((pos >> 10) & 63u); }
enum : int { PQ_RAY = 0, PQ_HIT = 1, PQ_NEW = 2 };

// One path sample between two blocks.  Fields as PathState's; a ray waiting in the ray queue is described by the same fields:
//   primary ray   (stage ST_EXT, mode 0):  origin O, direction rd
//   shadow ray    (stage ST_SH):           origin O + 0.0001 N, direction rd
//   bounce ray    (stage ST_EXT, mode 1|2): origin O + 0.0001 N, direction far * V
struct alignas(16) PoolSlot {   // the part the trace lanes read and write, in LDS
    float O[3]; uint32_t flags;  // flags: stage (bits 0-2) | mode (3-4) | li0 (5-15) | bounce (16-31)
    float N[3]; uint32_t bits;   // PB_PENDING | PB_BACKFACING
    float V[3]; int32_t pmin;    // pmin, hu, hv: the closest hit a trace lane found (O is then already the hit point)
    float rd[3]; float hu;
    float c[3]; float hv;
    float Ei[3]; float normalization;
};
static_assert(sizeof(PoolSlot) == 96, "PoolSlot layout");
struct alignas(16) PoolCold {   // the part only the shade waves touch, in global memory (one array of POOL_SLOTS per workgroup)
    float thr[3]; float aux;
    float diffuse[3]; float gloss;
    uint32_t rkey, rctr; int32_t sid, base;
};
static_assert(sizeof(PoolCold) == 48, "PoolCold layout");
constexpr size_t pool_cold_bytes_per_group() { return (size_t)POOL_SLOTS * sizeof(PoolCold); }

struct PoolCtl {
    uint32_t head[3], tail[3];
    uint32_t retired;  // slots that found the sample range exhausted
    uint32_t cursor;   // samples this workgroup has drawn
    uint32_t abort;    // a wave saw no progress for the watchdog's ticks: every wave of the workgroup leaves (the launch's result is invalid;
                       // counters[CTR_POOL_ABORT] tells the host).  Every wait in this file ends on it AND carries its own time bound (the
                       // per-lane ring waits raise it themselves), so a launch cannot hang the device.
    uint32_t watchdog_log2;        // the bound, as a power of two of s_memtime ticks (knobs bits 16-21; POOL_WATCHDOG_LOG2 unless a test lowers it)
    unsigned long long* counters;  // the context's counter words, for the waits that have no other way to report
};
constexpr int POOL_WATCHDOG_LOG2 = 31;
constexpr long long POOL_WATCHDOG_TICKS = 1ll << POOL_WATCHDOG_LOG2;
constexpr int CTR_POOL_ABORT = 7;
__device__ __forceinline__ bool pool_aborted(const PoolCtl* ctl) { return *(const volatile uint32_t*)&ctl->abort != 0u; }
__device__ __forceinline__ long long pool_watchdog_ticks(const PoolCtl* ctl) { return 1ll << (*(const volatile uint32_t*)&ctl->watchdog_log2 & 63u); }
// called by a wave that has waited too long: raises the flag and leaves the queue state where the host can read it (diagnostic words 8..)
__device__ __forceinline__ void pool_raise_abort(PoolCtl* ctl, unsigned long long* counters, int where) {
    if ((threadIdx.x & 63) == 0 && atomicExch(&ctl->abort, 1u) == 0u) {
        atomicAdd(&counters[CTR_POOL_ABORT], 1ull);
        if (atomicAdd(&counters[8], 1ull) == 0ull) {
            counters[9] = (unsigned long long)blockIdx.x | ((unsigned long long)(threadIdx.x >> 6) << 32);
            counters[10] = (unsigned long long)where;
            for (int q = 0; q < 3; ++q) { counters[11 + 2 * q] = ctl->head[q]; counters[12 + 2 * q] = ctl->tail[q]; }
            counters[17] = ctl->retired; counters[18] = ctl->cursor;
        }
    }
}

// the same from ONE lane that has waited too long inside a ring wait (whichever lane it is)
__device__ __forceinline__ void pool_raise_abort_lane(PoolCtl* ctl, int where) {
    if (atomicExch(&ctl->abort, 1u) == 0u) {
        unsigned long long* counters = ctl->counters;
        atomicAdd(&counters[CTR_POOL_ABORT], 1ull);
        if (atomicAdd(&counters[8], 1ull) == 0ull) {
            counters[9] = (unsigned long long)blockIdx.x | ((unsigned long long)(threadIdx.x >> 6) << 32);
            counters[10] = (unsigned long long)where;
        }
    }
}
// wait until ring place e is empty (want_empty) or filled; gives up on the abort flag or after the watchdog's ticks
__device__ __forceinline__ uint16_t pool_ring_wait(PoolCtl* ctl, volatile uint16_t* e, bool want_empty, int where) {
    uint16_t v = *e;
    if ((v == 0) == want_empty) return v;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime(), bound = pool_watchdog_ticks(ctl);
    while (((v = *e) == 0) != want_empty && !pool_aborted(ctl)) {
        __builtin_amdgcn_s_sleep(1);
        if ((long long)__builtin_amdgcn_s_memtime() - t0 > bound) { pool_raise_abort_lane(ctl, where); break; }
    }
    return v;
}

constexpr size_t pool_lds_bytes() {
    return (size_t)POOL_TW * POOL_NLDS * 64 * sizeof(uint2) + (size_t)POOL_SLOTS * sizeof(PoolSlot) + 3 * POOL_QCAP * sizeof(uint16_t) + sizeof(PoolCtl) + 64;
}

__device__ __forceinline__ uint32_t pool_flags(int stage, int mode, int li0, int bounce) {
    return (uint32_t)stage | ((uint32_t)mode << 3) | ((uint32_t)li0 << 5) | ((uint32_t)bounce << 16);
}
__device__ __forceinline__ uint32_t vload(const uint32_t* p) { return *(const volatile uint32_t*)p; }

// append the slots of the lanes with `pred` to queue q
__device__ __forceinline__ void pool_push(PoolCtl* ctl, uint16_t* ring, int q, bool pred, int slot, int lane) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    const int n = __popcll(m), rank = __popcll(m & ((1ull << lane) - 1ull));
    const int leader = __ffsll((long long)m) - 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the slot's state before its number
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&ctl->tail[q], (uint32_t)n);
    base = (uint32_t)__shfl((int)base, leader);
    if (pred) {
        volatile uint16_t* e = ring + q * POOL_QCAP + ((base + (uint32_t)rank) & (POOL_QCAP - 1));
        pool_ring_wait(ctl, e, true, 6);  // a reader that claimed this place a lap ago has not emptied it yet
        *e = pool_entry(slot, base + (uint32_t)rank);
    }
}

// hand up to popcount(want) slots of queue q to the lanes of `want` (lowest lanes first); returns how many (wave-uniform), 0 if fewer than
// min_take are queued
__device__ __forceinline__ int pool_pop(PoolCtl* ctl, uint16_t* ring, int q, unsigned long long want, int min_take, int lane, int& slot) {
    const int n_want = __popcll(want);
    uint32_t h = 0;
    int n = 0;
    if (lane == 0) {
        for (;;) {
            if (pool_aborted(ctl)) { n = 0; break; }
            h = vload(&ctl->head[q]);
            const int avail = (int)(vload(&ctl->tail[q]) - h);
            if (avail < min_take || avail <= 0) { n = 0; break; }
            n = avail < n_want ? avail : n_want;
            if (atomicCAS(&ctl->head[q], h, h + (uint32_t)n) == h) break;
        }
    }
    n = __shfl(n, 0);
    h = (uint32_t)__shfl((int)h, 0);
    if (n == 0) return 0;
    const int rank = __popcll(want & ((1ull << lane) - 1ull));
    if (((want >> lane) & 1ull) && rank < n) {
        volatile uint16_t* e = ring + q * POOL_QCAP + ((h + (uint32_t)rank) & (POOL_QCAP - 1));
        const uint16_t v = pool_ring_wait(ctl, e, false, 7);  // the writer has reserved this place and is about to fill it
        *e = 0;
        slot = (int)(v & 1023u) - 1;  // -1 if the wait was given up: the lane holds no slot, and every wave is leaving
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return n;
}

// ---- trace waves ----------------------------------------------------------------------------------------------------------------
// two pushes with their LDS round trips overlapped (both counters bumped, then both places checked, then both written)
__device__ __forceinline__ void pool_push2(PoolCtl* ctl, uint16_t* ring, int qa, bool pa, int qb, bool pb, int slot, int lane) {
    const unsigned long long ma = __ballot(pa), mb = __ballot(pb);
    if ((ma | mb) == 0ull) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int la = ma ? __ffsll((long long)ma) - 1 : -1, lb = mb ? __ffsll((long long)mb) - 1 : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the slots' state before their numbers
    uint32_t base = 0;
    if (lane == la) base = atomicAdd(&ctl->tail[qa], (uint32_t)__popcll(ma));
    else if (lane == lb) base = atomicAdd(&ctl->tail[qb], (uint32_t)__popcll(mb));  // la != lb: a lane pushes to one queue only
    const uint32_t base_a = (uint32_t)__shfl((int)base, la < 0 ? 0 : la), base_b = (uint32_t)__shfl((int)base, lb < 0 ? 0 : lb);
    if (pa || pb) {
        const uint32_t pos = pa ? base_a + (uint32_t)__popcll(ma & below) : base_b + (uint32_t)__popcll(mb & below);
        volatile uint16_t* e = ring + (pa ? qa : qb) * POOL_QCAP + (pos & (POOL_QCAP - 1));
        pool_ring_wait(ctl, e, true, 8);
        *e = pool_entry(slot, pos);
    }
}

// The ray queue is read through TICKETS: a lane without a ray takes the next place of the ring (one atomic per wave, no test whether
// anything is queued) and looks at that place once per loop iteration until a slot number appears in it -- no wave ever waits inside
// the queue code, and a ray pushed by a shade wave goes to the lane that has waited longest.
template <int NLDS, bool ALPHA>
__device__ __forceinline__ void pool_trace_wave(const TravScene& TS, const ptgpu_camera& cam, PoolSlot* slots, PoolCtl* ctl, uint16_t* ring,
                                                uint2* lds_lane, uint2* gl_lane, unsigned long long* counters, int refill_threshold) {
    const int lane = threadIdx.x & 63;
    Trav<ALPHA, false, NLDS, true> tr;
    tr.reset_counts();
    tr.attach(lds_lane, gl_lane);
    tr.alive = false;
    int slot = -1;
    uint32_t fl = 0u;          // the slot's flags word (kept from the time the ray was taken)
    int ticket = -1;           // place of the ray ring this lane waits at, or -1
    uint32_t n_rays = 0;
    long long t_progress = (long long)__builtin_amdgcn_s_memtime();
    const long long t_progress0 = t_progress;
#ifdef PT_POOL_STATS
    long long ps_stat[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long ps_t_begin = PS_NOW();
#endif
    uint32_t iteration = 0;
    for (;;) {
#ifdef PT_POOL_DEBUG
        if ((long long)__builtin_amdgcn_s_memtime() - t_progress0 > (1ll << 32)) { pool_raise_abort(ctl, counters, 5); break; }
#endif
        if ((++iteration & 255u) == 0u) {  // a ray that never ends (it cannot, on a validated tree) must not hold the device either
            if (__builtin_amdgcn_readfirstlane((int)pool_aborted(ctl))) break;
            if ((long long)__builtin_amdgcn_s_memtime() - t_progress > pool_watchdog_ticks(ctl)) { pool_raise_abort(ctl, counters, 3); break; }
        }
        const unsigned long long alive_mask = __ballot(tr.alive);
        const int n_fin = __popcll(__ballot(slot >= 0 && !tr.alive));
        // has a waiting lane's ray arrived?
        uint16_t arrived = 0;
        if (ticket >= 0) arrived = *(volatile uint16_t*)(ring + PQ_RAY * POOL_QCAP + (ticket & (POOL_QCAP - 1)));
        if (!pool_entry_is(arrived, (uint32_t)ticket)) arrived = 0;
        const int n_arrived = __popcll(__ballot(arrived != 0));
        const int n_free = __popcll(__ballot(slot < 0 && ticket < 0));
        if (alive_mask == 0ull && n_fin + n_arrived + n_free == 0) {
            // nothing to do but wait for a ray, at the lowest priority and mostly asleep: three trace waves share a SIMD with one shade wave,
            // and the rays they wait for come from that shade wave
            if (__builtin_amdgcn_readfirstlane(vload(&ctl->retired)) >= (uint32_t)POOL_SLOTS) break;
            if (__builtin_amdgcn_readfirstlane((int)pool_aborted(ctl))) break;
            if ((long long)__builtin_amdgcn_s_memtime() - t_progress > pool_watchdog_ticks(ctl)) { pool_raise_abort(ctl, counters, 1); break; }
            const long long ps_t1 = PS_NOW();
            __builtin_amdgcn_s_sleep(32);
            PS_ADD(4, PS_NOW() - ps_t1); PS_ADD(5, 1);
            continue;
        }
        if (alive_mask == 0ull || n_fin + n_arrived + n_free >= refill_threshold) {
            const long long ps_t0 = PS_NOW();
            PS_ADD(1, 1); PS_ADD(2, n_fin); PS_ADD(3, n_arrived);
            __builtin_amdgcn_s_setprio(2);
            if (n_fin + n_arrived > 0) t_progress = (long long)__builtin_amdgcn_s_memtime();
            bool to_hit = false, to_new = false, go = false, go_any = false;
            F3 go_o = f3(0.0f, 0.0f, 0.0f), go_d = go_o;
            // (1) rays that ended
            if (slot >= 0 && !tr.alive) {
                PoolSlot& S = slots[slot];
                if (tr.anyhit) {
                    // shadow ray of a path-tracing hit: reference :346-376 / :416-450 / :500-534 reduced to their outcome (as in integrate)
                    const float4 sO = *reinterpret_cast<const float4*>(S.O), sN = *reinterpret_cast<const float4*>(S.N), sV = *reinterpret_cast<const float4*>(S.V),
                                 sc = *reinterpret_cast<const float4*>(S.c), sE = *reinterpret_cast<const float4*>(S.Ei);
                    const uint32_t bits = __float_as_uint(sN.w);
                    const bool unoccluded = tr.pmin < 0;
                    if (unoccluded) { S.Ei[0] = sE.x + sc.x; S.Ei[1] = sE.y + sc.y; S.Ei[2] = sE.z + sc.z; }
                    if (!(bits & PB_PENDING)) { fl = (fl & ~7u) | (uint32_t)ST_FIN; to_new = true; }
                    else if (unoccluded && (bits & PB_BACKFACING)) {
                        S.N[0] = sN.x * -1.0f; S.N[1] = sN.y * -1.0f; S.N[2] = sN.z * -1.0f;
                        fl = (fl & ~7u) | (uint32_t)ST_REBOUNCE;
                        to_hit = true;
                    } else {  // the pending bounce ray, in this lane
                        fl = (fl & ~7u) | (uint32_t)ST_EXT;
                        go = true; go_any = false;
                        go_o = f3(sO.x, sO.y, sO.z) + 0.0001f * f3(sN.x, sN.y, sN.z);
                        go_d = cam.far_ * f3(sV.x, sV.y, sV.z);
                    }
                    S.flags = fl;
                } else {  // closest-hit ray
                    S.pmin = tr.pmin;
                    if (tr.pmin >= 0) {
                        const F3 P = tr.o + tr.tmin * tr.d;
                        S.O[0] = P.x; S.O[1] = P.y; S.O[2] = P.z;
                        S.hu = tr.umin; S.hv = tr.vmin;
                        to_hit = true;
                    } else to_new = true;
                }
            }
            pool_push2(ctl, ring, PQ_HIT, to_hit, PQ_NEW, to_new, slot, lane);
            if (to_hit || to_new) slot = -1;
            // (2) rays that arrived
            if (arrived != 0) {
                *(volatile uint16_t*)(ring + PQ_RAY * POOL_QCAP + (ticket & (POOL_QCAP - 1))) = 0;
                ticket = -1;
                slot = (int)(arrived & 1023u) - 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (arrived != 0) {
                const PoolSlot& S = slots[slot];
                const float4 sO = *reinterpret_cast<const float4*>(S.O), sN = *reinterpret_cast<const float4*>(S.N), sV = *reinterpret_cast<const float4*>(S.V),
                             sr = *reinterpret_cast<const float4*>(S.rd);
                fl = __float_as_uint(sO.w);
                const int stage = (int)(fl & 7u), mode = (int)((fl >> 3) & 3u);
                const F3 O = f3(sO.x, sO.y, sO.z);
                go = true; go_any = stage == ST_SH;
                if (stage == ST_EXT && mode == 0) { go_o = O; go_d = f3(sr.x, sr.y, sr.z); }
                else {
                    go_o = O + 0.0001f * f3(sN.x, sN.y, sN.z);
                    if (stage == ST_SH) go_d = f3(sr.x, sr.y, sr.z);
                    else go_d = cam.far_ * f3(sV.x, sV.y, sV.z);
                }
            }
            if (go) {
                tr.begin(TS, go_o, go_d, go_any);
                n_rays += 1;
            }
            // (3) tickets for the lanes that have neither a ray nor a place in the ring
            {
                const bool want = slot < 0 && ticket < 0;
                const unsigned long long wm = __ballot(want);
                if (wm != 0ull) {
                    const int leader = __ffsll((long long)wm) - 1;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(&ctl->head[PQ_RAY], (uint32_t)__popcll(wm));
                    base = (uint32_t)__shfl((int)base, leader);
                    if (want) ticket = (int)((base + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull))) & 0x3fffffffu);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            PS_ADD(0, PS_NOW() - ps_t0);
            if (__ballot(tr.alive) == 0ull) continue;  // rays that ended at once (missed the scene's box) are settled, or the wave waits, above
        }
#ifdef PT_POOL_STATS
        const long long ps_t2 = PS_NOW();
        PS_ADD(7, 1); PS_ADD(8, __popcll(__ballot(tr.alive)));
        {
            const int nn_ = __popcll(__ballot(tr.alive && tr.at_node())), nt_ = __popcll(__ballot(tr.alive && !tr.at_node()));
            PS_ADD(9, nn_ >= nt_ ? nn_ : nt_);
        }
#endif
        __builtin_amdgcn_s_setprio(PT_PRIO_STEP);
        {   // a phase of steps of one kind, as in integrate
            const bool node_lane = tr.alive && tr.at_node(), tri_lane = tr.alive && !tr.at_node();
            if (__popcll(__ballot(node_lane)) * PT_PHASE_BIAS_N >= __popcll(__ballot(tri_lane)) * PT_PHASE_BIAS_T) {
                if (tr.alive && tr.at_node()) tr.step_node(TS);
                if (tr.alive && tr.at_node()) tr.step_node(TS);
                if (tr.alive && tr.at_node()) tr.step_node(TS);
            } else {
                if (tr.alive && !tr.at_node()) tr.step_triangle(TS);
                if (tr.alive && !tr.at_node()) tr.step_triangle(TS);
                if (tr.alive && !tr.at_node()) tr.step_triangle(TS);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PS_ADD(6, PS_NOW() - ps_t2);
    }
#ifdef PT_POOL_STATS
    PS_ADD(10, PS_NOW() - ps_t_begin);
    if (lane == 0) for (int k = 0; k < 11; ++k) atomicAdd(&counters[8 + k], (unsigned long long)ps_stat[k]);
#endif
    unsigned long long wr = n_rays;
    for (int off = 32; off > 0; off >>= 1) wr += __shfl_down(wr, off);
    if (lane == 0 && wr) atomicAdd(&counters[0], wr);
}

// ---- shade waves ----------------------------------------------------------------------------------------------------------------
struct PoolPath {  // a slot in registers
    F3 O, N, V, rd, c, Ei, thr, diffuse;
    float gloss, normalization, hu, hv, aux;
    uint32_t bits;
    int pmin, stage, mode, li0, bounce, sid, base;
    Rng R;
};
__device__ __forceinline__ void pool_load(const PoolSlot& S, const PoolCold& C, PoolPath& p) {
    const float4 g = *reinterpret_cast<const float4*>(C.thr), h = *reinterpret_cast<const float4*>(C.diffuse);
    const uint4 r = *reinterpret_cast<const uint4*>(&C.rkey);
    const float4 a = *reinterpret_cast<const float4*>(S.O), b = *reinterpret_cast<const float4*>(S.N), c = *reinterpret_cast<const float4*>(S.V),
                 d = *reinterpret_cast<const float4*>(S.rd), e = *reinterpret_cast<const float4*>(S.c), f = *reinterpret_cast<const float4*>(S.Ei);
    p.O = f3(a.x, a.y, a.z); const uint32_t fl = __float_as_uint(a.w);
    p.stage = (int)(fl & 7u); p.mode = (int)((fl >> 3) & 3u); p.li0 = (int)((fl >> 5) & 2047u); p.bounce = (int)(fl >> 16);
    p.N = f3(b.x, b.y, b.z); p.bits = __float_as_uint(b.w);
    p.V = f3(c.x, c.y, c.z); p.pmin = __float_as_int(c.w);
    p.rd = f3(d.x, d.y, d.z); p.hu = d.w;
    p.c = f3(e.x, e.y, e.z); p.hv = e.w;
    p.Ei = f3(f.x, f.y, f.z); p.normalization = f.w;
    p.thr = f3(g.x, g.y, g.z); p.aux = g.w;
    p.diffuse = f3(h.x, h.y, h.z); p.gloss = h.w;
    p.R.key = r.x; p.R.ctr = r.y; p.sid = (int)r.z; p.base = (int)r.w;
}
__device__ __forceinline__ void pool_store(PoolSlot& S, PoolCold& C, const PoolPath& p) {
    *reinterpret_cast<float4*>(C.thr) = make_float4(p.thr.x, p.thr.y, p.thr.z, p.aux);
    *reinterpret_cast<float4*>(C.diffuse) = make_float4(p.diffuse.x, p.diffuse.y, p.diffuse.z, p.gloss);
    *reinterpret_cast<uint4*>(&C.rkey) = make_uint4(p.R.key, p.R.ctr, (uint32_t)p.sid, (uint32_t)p.base);
    *reinterpret_cast<float4*>(S.O) = make_float4(p.O.x, p.O.y, p.O.z, __uint_as_float(pool_flags(p.stage, p.mode, p.li0, p.bounce)));
    *reinterpret_cast<float4*>(S.N) = make_float4(p.N.x, p.N.y, p.N.z, __uint_as_float(p.bits));
    *reinterpret_cast<float4*>(S.V) = make_float4(p.V.x, p.V.y, p.V.z, __int_as_float(p.pmin));
    *reinterpret_cast<float4*>(S.rd) = make_float4(p.rd.x, p.rd.y, p.rd.z, p.hu);
    *reinterpret_cast<float4*>(S.c) = make_float4(p.c.x, p.c.y, p.c.z, p.hv);
    *reinterpret_cast<float4*>(S.Ei) = make_float4(p.Ei.x, p.Ei.y, p.Ei.z, p.normalization);
}

// loop control of tracePath between two rays (integrate's `advance`): 1 = fixed-bounce loop test (:389), 2 = roulette entry (:466-468),
// 3 = roulette loop test (:471)
__device__ __forceinline__ void pool_advance(PoolPath& ps, const ptgpu_params& prm, int bounces, int next_ctl, float terminate) {
    if (next_ctl == 1) {
        if (ps.bounce < bounces) { ps.mode = 1; ps.stage = ST_BOUNCE; return; }
        next_ctl = 2;
    }
    if (next_ctl == 2) {
        if (prm.bounces < 0) { terminate = ps.R.getF32(0.0f, 1.0f); next_ctl = 3; }
        else { ps.stage = ST_FIN; return; }
    }
    if (terminate > prm.termination_prob) { ps.mode = 2; ps.stage = ST_BOUNCE; }
    else ps.stage = ST_FIN;
}

template <int NLDS, bool ALPHA>
__device__ __forceinline__ void pool_shade_wave(const TravScene& TS, const ShadeScene& SS, const ptgpu_params& prm, const ptgpu_camera& cam,
                                                const ptgpu_light* __restrict__ lights, const PassDesc& pd, float4* __restrict__ rad,
                                                PoolSlot* slots, PoolCold* cold, PoolCtl* ctl, uint16_t* ring, int low_water, unsigned long long* counters) {
    const int lane = threadIdx.x & 63;
    const int bounces = prm.bounces < 0 ? -prm.bounces : prm.bounces;
    const float PI = 3.14159265358979323846f;
    const uint32_t total = (uint32_t)pd.num_blocks * (uint32_t)pd.samples_per_block;
    long long t_progress = (long long)__builtin_amdgcn_s_memtime();
    const long long t_progress0 = t_progress;
#ifdef PT_POOL_STATS
    long long ps_stat[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long ps_t_begin = PS_NOW();
#endif
    for (;;) {
        const long long ps_t0 = PS_NOW();
#ifdef PT_POOL_DEBUG
        if ((long long)__builtin_amdgcn_s_memtime() - t_progress0 > (1ll << 32)) { pool_raise_abort(ctl, counters, 4); break; }
#endif
        if (__builtin_amdgcn_readfirstlane((int)pool_aborted(ctl))) break;
        if (__builtin_amdgcn_readfirstlane(vload(&ctl->retired)) >= (uint32_t)POOL_SLOTS) break;
        const int hq = __builtin_amdgcn_readfirstlane((int)(vload(&ctl->tail[PQ_HIT]) - vload(&ctl->head[PQ_HIT])));
        const int nq = __builtin_amdgcn_readfirstlane((int)(vload(&ctl->tail[PQ_NEW]) - vload(&ctl->head[PQ_NEW])));
        int job = -1;
        if (hq >= 64) job = PQ_HIT;
        else if (nq >= 64) job = PQ_NEW;
        else if (__builtin_amdgcn_readfirstlane((int)(vload(&ctl->tail[PQ_RAY]) - vload(&ctl->head[PQ_RAY]))) < low_water) {  // the trace waves are about to run dry
            if (hq > 0 && hq >= nq) job = PQ_HIT;
            else if (nq > 0) job = PQ_NEW;
        }
        if (job < 0) {
            if (__builtin_amdgcn_readfirstlane((int)pool_aborted(ctl))) break;
            if ((long long)__builtin_amdgcn_s_memtime() - t_progress > pool_watchdog_ticks(ctl)) { pool_raise_abort(ctl, counters, 2); break; }
            __builtin_amdgcn_s_sleep(8); PS_ADD(0, PS_NOW() - ps_t0); PS_ADD(1, 1);
            continue;
        }
        t_progress = (long long)__builtin_amdgcn_s_memtime();
        int slot = -1;
        if (pool_pop(ctl, ring, job, ~0ull, 1, lane, slot) == 0) continue;
        PS_ADD(job == PQ_HIT ? 3 : 6, 1); PS_ADD(job == PQ_HIT ? 4 : 7, __popcll(__ballot(slot >= 0)));
        PoolPath ps;
        ps.stage = ST_IDLE;
        if (slot >= 0) pool_load(slots[slot], cold[slot], ps);
        bool to_ray = false, to_hit = false, to_new = false;
        if (job == PQ_HIT) {
            __builtin_amdgcn_s_setprio(PT_PRIO_HIT);
            if (slot >= 0) {
                const bool hit_lane = ps.stage == ST_EXT;
                bool want_bounce = ps.stage == ST_BOUNCE || ps.stage == ST_REBOUNCE;
                if (hit_lane) {
                    // hit set-up: reference :300-344 (first hit), :403-414 / :486-498 (bounce hits)
                    const float4 rec = tri_record(TS, ps.pmin)[3];
                    const int hit_tri = __float_as_int(rec.x), hit_mat = __float_as_int(rec.w);
                    const bool first = ps.mode == 0;
                    if (first) {
                        ps.gloss = SS.materials[hit_mat].glossiness;
                        ps.normalization = (ps.gloss + 2.0f) / (4.0f * PI * (2.0f - ptm_pow(2.0f, -ps.gloss / 2.0f)));
                        ps.bounce = 0;
                        ps.thr = f3(1.0f, 1.0f, 1.0f);
                        ps.V = normalize(ps.rd);
                    }
                    // ps.O is the hit point already (trace lane: o + tmin * d)
                    F3 specular;
                    get_texture_parameters(SS, prm, hit_tri, hit_mat, ps.hu, ps.hv, ps.diffuse, ps.N, specular);
                    const int li = get_light_to_sample(ps.O, lights, pd.num_lights, ps.R);
                    if (first) ps.li0 = li;
                    const ptgpu_light& Lt = lights[li];
                    // AreaLight::sample, src/AreaLight.cpp:69-85
                    int dim = 4 + 4 * sobol_bounce_dim(ps.bounce);
                    int r = ps.R.getS32(1, 10000);
                    float sa, sb;
                    sobol32x2(SS.sobol_pairs, (uint32_t)(ps.base + r), (unsigned)dim, sa, sb);
                    float lx = (2.0f * sa - 1.0f) * Lt.size[0];
                    float ly = (2.0f * sb - 1.0f) * Lt.size[1];
                    F3 Pl = light_col(Lt, 3) + light_col(Lt, 0) * lx + light_col(Lt, 1) * ly;
                    const float pdf = 1.0f / (4.0f * Lt.size[0] * Lt.size[1]);
                    const F3 L = Pl - ps.O;
                    const ptgpu_light& L0 = lights[ps.li0];
                    const F3 emission = f3(L0.E[0], L0.E[1], L0.E[2]) * prm.intensity;  // set once per path in the reference (:341)
                    ps.rd = L;  // the shadow ray: origin O + 0.0001 N
                    {   // what the light adds if the shadow ray finds nothing: reference :346-376 / :416-450 / :500-534
                        F3 D = normalize(L);
                        F3 Nl = -light_col(Lt, 2);
                        const bool back = dot(ps.N, ps.V) > 0.0f;
                        const F3 Nf = back ? ps.N * -1.0f : ps.N;
                        float cos_theta = sel_max(0.0f, dot(Nf, D));
                        float cos_theta_l = sel_max(0.0f, dot(Nl, -D));
                        float d = length(L);
                        float inv_d = sel_min(1.0f / (d * d), 100000.0f);
                        float inv_pdf = sel_min(1.0f / pdf, 100000.0f);
                        F3 H = normalize(D - ps.V);
                        F3 color = ps.diffuse;
                        color = color + specular * ptm_pow(sel_max(0.0f, dot(Nf, H)), ps.gloss) * ps.normalization;
                        const float invPi = 0.31830988618379f;
                        if (ps.mode == 0) ps.c = color * invPi * emission * cos_theta * cos_theta_l * inv_d * inv_pdf;
                        else if (ps.mode == 1) ps.c = color * invPi * ps.thr * emission * cos_theta * cos_theta_l * inv_d * inv_pdf;
                        else ps.c = color * invPi * ps.thr * emission * cos_theta * cos_theta_l * inv_d * inv_pdf * pd.inv_prob;
                        ps.bits = back ? PB_BACKFACING : 0u;
                    }
                    // loop control after the shade (random draws in the reference's order)
                    if (ps.mode == 2) { const float terminate = ps.R.getF32(0.0f, 1.0f); ps.bounce += 1; pool_advance(ps, prm, bounces, 3, terminate); }
                    else { if (ps.mode == 1) ps.bounce += 1; pool_advance(ps, prm, bounces, 1, 0.0f); }
                    want_bounce = ps.stage == ST_BOUNCE;
                }
                if (want_bounce) {
                    // reference :391-401 / :473-484; drawCosineWeightedDirection :57-73 (RNG peeked, not advanced)
                    int dim = 2 + 4 * sobol_bounce_dim(ps.bounce);
                    int rnd = ps.R.peekS32(1, 1234);
                    float sa, sb;
                    sobol32x2(SS.sobol_pairs, (uint32_t)(ps.base + rnd), (unsigned)dim, sa, sb);
                    float a = 2.0f * sa - 1.0f, b = 2.0f * sb - 1.0f;
                    float x, y;
                    map_to_disk(a, b, x, y);
                    float z = sqrtf(1 - x * x - y * y);
                    ps.V = normalize(basis_times(ps.N, f3(x, y, z)));
                    if (ps.stage != ST_REBOUNCE) ps.thr = ps.thr * ps.diffuse;
                    if (hit_lane) ps.bits |= PB_PENDING;  // the shadow ray goes first
                    else ps.stage = ST_EXT;               // bounce ray: O + 0.0001 N, far * V
                }
                if (hit_lane) ps.stage = ST_SH;
                to_ray = true;
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
            if (slot >= 0) {
                // closest-hit ray that missed
                if (ps.stage == ST_EXT) {
                    if (ps.mode == 1) pool_advance(ps, prm, bounces, 2, 0.0f);  // inside the fixed-bounce loop: `break`, then the roulette section
                    else ps.stage = ST_FIN;                                      // primary miss (Ei = 0), or miss inside the roulette loop
                }
                if (ps.stage == ST_FIN) {
                    float4 o;
                    o.x = ps.Ei.x; o.y = ps.Ei.y; o.z = ps.Ei.z; o.w = ps.aux;
                    rad[ps.sid] = o;
                    ps.stage = ST_IDLE;
                    ps.sid = -1;
                }
                if (ps.stage == ST_BOUNCE) to_hit = true;
            }
            // new samples for the idle slots
            const unsigned long long need = __ballot(slot >= 0 && ps.stage == ST_IDLE);
            if (need != 0ull) {
                uint32_t cbase = 0;
                if (lane == 0) cbase = atomicAdd(&ctl->cursor, (uint32_t)__popcll(need));
                cbase = (uint32_t)__shfl((int)cbase, 0);
                const bool cancelled = pass_cancelled(pd);  // ptgpu_cancel: the slots retire as if the range were exhausted
                if (slot >= 0 && ps.stage == ST_IDLE) {
                    const uint32_t cc = cbase + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                    // round r of this workgroup draws chunk r * G + (g + 997 r) mod G of the launch's range (rotated so that a workgroup does
                    // not draw the same image region in every round)
                    const uint32_t r = cc / POOL_CHUNK, off = cc % POOL_CHUNK, G = gridDim.x;
                    const unsigned long long chunk = (unsigned long long)r * G + (blockIdx.x + (r % G) * 997u) % G;
                    const unsigned long long sid64 = chunk * POOL_CHUNK + off;
                    if (sid64 >= (unsigned long long)total || cancelled) {
                        ps.stage = ST_FIN;  // (transient) retired: every later draw of this workgroup lies beyond the range too
                    } else {
                        int n_width = prm.width, n_height = prm.height, n_spb = pd.samples_per_block, n_blocks_x = pd.blocks_x, n_aa = prm.num_aa_rays;
                        float n_filter = prm.filter_width;
                        const float div_x = 1.0f / (float)n_width, div_y = 1.0f / (float)n_height;
                        const int sid = (int)sid64;
                        bool valid;
                        float px = 0.0f, py = 0.0f;
                        int sbase;
                        const int jj = (int)((uint32_t)sid / (uint32_t)n_spb);
                        const int s = sid - jj * n_spb;
                        const int idx = pd.block_begin + (pd.j0 + jj) * pd.block_stride;
                        const int bx = (idx % n_blocks_x) * 32, by = (idx / n_blocks_x) * 32;
                        const int bw = (bx + 32 < n_width) ? 32 : n_width - bx;
                        const int bh = (by + 32 < n_height) ? 32 : n_height - by;
                        const int base = block_sobol_base(idx, prm.pass_seed);
                        if (prm.sobol_block) {  // reference :706-727
                            const int i = s;
                            valid = i < bw * bh * n_aa;
                            float sa, sb;
                            sobol32x2(SS.sobol_pairs, (uint32_t)(base + i), 0u, sa, sb);
                            px = (float)bx + (float)bw * sa;
                            py = (float)by + (float)bh * sb;
                            sbase = base + i;
                            ps.R.seed((uint32_t)idx, prm.pass_seed, (uint32_t)i, 0u);
                            int pixel_x = (int)floorf(px); if (pixel_x > n_width - 1) pixel_x = n_width - 1;
                            int pixel_y = (int)floorf(py); if (pixel_y > n_height - 1) pixel_y = n_height - 1;
                            ps.aux = __int_as_float(pixel_y * n_width + pixel_x);  // global pixel index; accumulate checks the block
                        } else {  // reference :743-779
                            const int i = s / n_aa, k = s - i * n_aa;
                            valid = i < bw * bh;
                            const int pixel_x = bx + (i % bw), pixel_y = by + (i / bw);
                            float sa, sb;
                            sobol32x2(SS.sobol_pairs, (uint32_t)(base + i + k), 0u, sa, sb);
                            float a = n_filter * (sa - 0.5f) + 0.5f;
                            float b = n_filter * (sb - 0.5f) + 0.5f;
                            px = (float)pixel_x + a;
                            py = (float)pixel_y + b;
                            if (px < 0.0f) px *= -1.0f;
                            if (py < 0.0f) py *= -1.0f;
                            if (px > (float)n_width) px -= a;
                            if (py > (float)n_height) py -= b;
                            sbase = base + i;
                            ps.R.seed((uint32_t)idx, prm.pass_seed, (uint32_t)i, (uint32_t)k);
                            const float fs = 2.0f / n_filter;
                            ps.aux = gauss(fs * (a - 0.5f), fs * (b - 0.5f));
                        }
                        if (valid) {
                            float x = px * div_x * 2.0f - 1.0f;
                            float y = py * div_y * -2.0f + 1.0f;
                            // camera ray, reference :244-292
                            F4 Roh = mul44(cam.invP, x, y, 0.0f, 1.0f);
                            float iw0 = 1.0f / Roh.w;
                            F3 Ro = f3(Roh.x * iw0, Roh.y * iw0, Roh.z * iw0);
                            F4 Rdh = mul44(cam.invP, x, y, 1.0f, 1.0f);
                            float iw1 = 1.0f / Rdh.w;
                            F3 Rd = f3(Rdh.x * iw1, Rdh.y * iw1, Rdh.z * iw1) - Ro;
                            if (prm.depth_of_field) {
                                F3 Rn = normalize(Rd);
                                F3 Pn = f3(cam.forward[0], cam.forward[1], cam.forward[2]);
                                float Pd = prm.focal_distance / sel_max(0.01f, dot(Rn, Pn));  // :272
                                F3 Rf = Ro + Pd * Rn;
                                float xr, yr;
                                float a = ps.R.getF32(-1.0f, 1.0f);
                                float b = ps.R.getF32(-1.0f, 1.0f);
                                map_to_disk(a, b, xr, yr);
                                F3 up = f3(cam.orient_col0[0], cam.orient_col0[1], cam.orient_col0[2]);
                                F3 right = f3(cam.orient_col1[0], cam.orient_col1[1], cam.orient_col1[2]);
                                Ro = Ro + prm.aperture * (yr * up + xr * right);
                                Rd = cam.far_ * normalize(Rf - Ro);
                            }
                            ps.sid = sid;
                            ps.base = sbase;
                            ps.mode = 0;
                            ps.bounce = 0;
                            ps.li0 = 0;
                            ps.bits = 0u;
                            ps.Ei = f3(0.0f, 0.0f, 0.0f);
                            ps.O = Ro;
                            ps.rd = Rd;
                            ps.stage = ST_EXT;
                            to_ray = true;
                        } else {
                            to_new = true;  // padded sample id: nothing to trace, the slot draws again
                        }
                    }
                }
            }
            const int n_retired = __popcll(__ballot(slot >= 0 && ps.stage == ST_FIN));
            if (n_retired > 0 && lane == 0) atomicAdd(&ctl->retired, (uint32_t)n_retired);
        }
        if (slot >= 0 && (to_ray || to_hit || to_new)) pool_store(slots[slot], cold[slot], ps);
        pool_push(ctl, ring, PQ_RAY, to_ray, slot, lane);
        pool_push(ctl, ring, PQ_HIT, to_hit, slot, lane);
        pool_push(ctl, ring, PQ_NEW, to_new, slot, lane);
        PS_ADD(job == PQ_HIT ? 2 : 5, PS_NOW() - ps_t0);
    }
#ifdef PT_POOL_STATS
    PS_ADD(8, PS_NOW() - ps_t_begin);
    if (lane == 0) for (int k = 0; k < 9; ++k) atomicAdd(&counters[19 + k], (unsigned long long)ps_stat[k]);
#endif
}

// One workgroup of POOL_THREADS threads; smem = pool_lds_bytes() of dynamic LDS.  stack_ovf: overflow stack space for gridDim.x * POOL_TW waves.
template <int NLDS, bool ALPHA>
__device__ __forceinline__ void integrate_pool(const TravScene& TS, const ShadeScene& SS, const ptgpu_params& prm, const ptgpu_camera& cam,
                                               const ptgpu_light* __restrict__ lights, const PassDesc& pd, float4* __restrict__ rad,
                                               unsigned long long* counters, unsigned char* smem, uint2* stack_ovf, PoolCold* cold_all, int knobs) {
    uint2* stacks = reinterpret_cast<uint2*>(smem);
    PoolSlot* slots = reinterpret_cast<PoolSlot*>(stacks + (size_t)POOL_TW * NLDS * 64);
    uint16_t* ring = reinterpret_cast<uint16_t*>(slots + POOL_SLOTS);
    PoolCtl* ctl = reinterpret_cast<PoolCtl*>(ring + 3 * POOL_QCAP);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 3 * POOL_QCAP; i += POOL_THREADS) ring[i] = (i >= PQ_NEW * POOL_QCAP && i < PQ_NEW * POOL_QCAP + POOL_SLOTS) ? (uint16_t)(i - PQ_NEW * POOL_QCAP + 1) : (uint16_t)0;
    for (int i = tid; i < POOL_SLOTS; i += POOL_THREADS) {
        slots[i].flags = pool_flags(ST_IDLE, 0, 0, 0);
        cold_all[(size_t)blockIdx.x * POOL_SLOTS + i].sid = -1;
    }
    if (tid == 0) {
        ctl->head[0] = ctl->head[1] = ctl->head[2] = 0u;
        ctl->tail[PQ_RAY] = 0u; ctl->tail[PQ_HIT] = 0u; ctl->tail[PQ_NEW] = (uint32_t)POOL_SLOTS;
        ctl->retired = 0u;
        ctl->cursor = 0u;
        ctl->abort = 0u;
        ctl->watchdog_log2 = (uint32_t)((knobs >> 16) & 63);
        ctl->counters = counters;
    }
    __syncthreads();
    if (wave < POOL_TW) {
        const size_t gw = (size_t)blockIdx.x * POOL_TW + wave;
        pool_trace_wave<NLDS, ALPHA>(TS, cam, slots, ctl, ring, stacks + (size_t)wave * NLDS * 64 + lane, stack_ovf + gw * (64 - NLDS) * 64 + lane, counters,
                                     knobs & 0xff);
    } else {
        pool_shade_wave<NLDS, ALPHA>(TS, SS, prm, cam, lights, pd, rad, slots, cold_all + (size_t)blockIdx.x * POOL_SLOTS, ctl, ring, (knobs >> 8) & 0xff, counters);
    }
}

}  // namespace ptd
