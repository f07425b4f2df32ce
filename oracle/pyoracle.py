"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (CPU restatement of the reference's hot path) and, where present,
oracle/_ref/libptref.so (the reference's own sobol.cpp / rtIntersect.inl compiled where they lie).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(pathtracer_amd/) never does.  The POD structs are layout-identical to include/ptgpu.h, so the ctypes
classes of pathtracer_amd.capi are reused for them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd.capi import (FLATNODE_DTYPE, Camera, Counters, Light, Material, Params, SceneDesc,
                                 make_scene_desc)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libptref.so")
_P = C.c_void_p
_lib = None
_ref = None


def build():
    """Compile the oracle (and oracle/_ref when /root/reference exists). Building the checker is not using it."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.pto_sobol_sample.restype = C.c_float
        L.pto_sobol_sample.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.pto_sobol_matrices.restype = C.POINTER(C.c_uint32)
        L.pto_gauss.restype = C.c_float
        L.pto_gauss.argtypes = [C.c_float, C.c_float, C.c_int]
        L.pto_rng_u32.restype = C.c_uint32
        L.pto_rng_u32.argtypes = [C.c_uint32] * 5
        L.pto_map_to_disk.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.pto_bvh_build.restype = C.c_int32
        L.pto_bvh_build.argtypes = [C.c_int32, _P, C.c_int, _P, _P, _P]
        L.pto_raycast_batch.argtypes = [C.POINTER(SceneDesc), C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P,
                                        C.POINTER(Counters), C.c_int]
        L.pto_gen_primary.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, _P, _P]
        L.pto_gen_shadow.argtypes = [C.POINTER(SceneDesc), C.POINTER(Light), C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]
        L.pto_render_pass.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params), C.POINTER(Camera), _P, C.c_int, _P,
                                      C.POINTER(Counters), C.c_int, C.c_int, C.c_int, C.c_int]
        L.pto_render_pass2.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params), C.POINTER(Camera), _P, C.c_int, _P,
                                       C.POINTER(Counters), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                       C.POINTER(C.c_int)]
        L.pto_resolve.argtypes = [_P, C.c_int, C.c_int, _P, C.c_int]
        L.pto_trace_path.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params), C.POINTER(Camera), _P, C.c_int, C.c_float,
                                     C.c_float, C.c_int, _P, C.c_int, _P, C.POINTER(Counters)]
        L.pto_trace_samples.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params), C.POINTER(Camera), _P, C.c_int, C.c_int64, _P, _P, _P,
                                        C.c_int, _P, C.c_int]
        L.pto_rng_key.restype = C.c_uint32
        L.pto_rng_key.argtypes = [C.c_uint32] * 4
        L.pto_texture_parameters.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params), C.c_int64, _P, _P, _P, _P]
        L.pto_light_to_sample.argtypes = [_P, C.c_int, C.c_int64, _P, _P, _P]
        L.pto_math_probe.argtypes = [C.c_int, _P, _P, _P, C.c_int, C.c_int]
        L.pto_woop_from_vertices.argtypes = [_P, _P, _P, _P]
        L.pto_intersect_woop.argtypes = [_P, _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.pto_intersect_triangle1.argtypes = [_P] * 5 + [C.POINTER(C.c_float)] * 3
        L.pto_box_intersect.argtypes = [_P, _P, _P, _P, C.c_float]
        L.pto_form_basis.argtypes = [_P, _P]
        L.pto_saturate.argtypes = [_P, C.c_float, _P]
        L.pto_light_sample.argtypes = [C.POINTER(Light), C.c_int, C.c_int, _P, C.POINTER(C.c_float), _P]
        _lib = L
    return _lib


def ref():
    """The reference-built checker library, or None when it was never built (no /root/reference at build time)."""
    global _ref
    if _ref is None and os.path.exists(REF_LIB):
        R = C.CDLL(REF_LIB)
        R.ptref_sobol_sample.restype = C.c_float
        R.ptref_sobol_sample.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        R.ptref_sobol_matrices.restype = C.POINTER(C.c_uint32)
        for f in (R.ptref_intersect_triangle1, R.ptref_intersect_triangle2):
            f.argtypes = [_P] * 5 + [C.POINTER(C.c_float)] * 3
        R.ptref_tri_box_overlap.argtypes = [_P] * 5
        _ref = R
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sobol(index, dim, scramble=0):
    return lib().pto_sobol_sample(int(index), int(dim), int(scramble))


def sobol_matrices():
    return np.ctypeslib.as_array(lib().pto_sobol_matrices(), shape=(1024 * 52,)).copy()


def woop_from_positions(pos):
    """pos [n,3,3] -> woop [n,12] via the tri_data restatement (reference src/RTTriangle.hpp:14-20)."""
    pos = _f32(pos).reshape(-1, 3, 3)
    out = np.empty((pos.shape[0], 12), dtype=np.float32)
    L = lib()
    for i in range(pos.shape[0]):
        L.pto_woop_from_vertices(pos[i, 0].ctypes.data, pos[i, 1].ctypes.data, pos[i, 2].ctypes.data, out[i].ctypes.data)
    return out


def bvh_build(pos, sah=True):
    """Serial restatement of BVH::BVH (reference src/Bvh.cpp:11-274). Returns (nodes, list, stats dict)."""
    pos = _f32(pos).reshape(-1, 3, 3)
    n = pos.shape[0]
    nodes = np.zeros(max(2 * n, 1), dtype=FLATNODE_DTYPE)
    lst = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(6, dtype=np.int32)
    cnt = lib().pto_bvh_build(n, pos.ctypes.data, int(bool(sah)), lst.ctypes.data, nodes.ctypes.data, st.ctypes.data)
    stats = dict(total_splits=int(st[0]), bad_splits=int(st[1]), leaves=int(st[2]), true_depth=int(st[3]),
                 max_leaf_tris=int(st[4]), depth_field=int(st[5]), nodes=int(cnt))
    return nodes[:cnt].copy(), lst[:n].copy(), stats


def raycast(sd, orig, direction, occlusion=False, use_alpha=True, nthreads=1):
    desc, keep = make_scene_desc(sd)
    o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
    n = o.shape[0]
    out = dict(id=np.empty(n, np.int32), t=np.empty(n, np.float32), u=np.empty(n, np.float32), v=np.empty(n, np.float32))
    ctr = Counters()
    lib().pto_raycast_batch(C.byref(desc), int(use_alpha), o.ctypes.data, d.ctypes.data, n, int(occlusion),
                            out["id"].ctypes.data, out["t"].ctypes.data, out["u"].ctypes.data, out["v"].ctypes.data,
                            C.byref(ctr), nthreads)
    out["counters"] = dict(rays=ctr.rays, node_visits=ctr.node_visits, tri_tests=ctr.tri_tests)
    return out


def gen_primary(cam, width, height):
    o = np.empty((width * height, 3), np.float32)
    d = np.empty((width * height, 3), np.float32)
    lib().pto_gen_primary(C.byref(cam), width, height, o.ctypes.data, d.ctypes.data)
    return o, d


def gen_shadow(sd, light, orig, direction, hit):
    desc, keep = make_scene_desc(sd)
    o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
    n = o.shape[0]
    so, sdir, valid = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.int32)
    lib().pto_gen_shadow(C.byref(desc), C.byref(light), n, o.ctypes.data, d.ctypes.data, hit["id"].ctypes.data,
                         hit["t"].ctypes.data, hit["u"].ctypes.data, hit["v"].ctypes.data, so.ctypes.data,
                         sdir.ctypes.data, valid.ctypes.data)
    return so, sdir, valid


def render_pass(sd, params, cam, lights, accum=None, nthreads=1, libm=False, block_begin=0, block_end=-1, block_stride=1):
    """One pass of pathTraceBlock over blocks block_begin, block_begin + block_stride, ... < block_end accumulated into accum [h,w,4]."""
    desc, keep = make_scene_desc(sd)
    if accum is None:
        accum = np.zeros((params.height, params.width, 4), np.float32)
    larr = (Light * len(lights))(*lights)
    ctr = Counters()
    lib().pto_render_pass2(C.byref(desc), C.byref(params), C.byref(cam), C.cast(larr, _P), len(lights), accum.ctypes.data,
                           C.byref(ctr), nthreads, int(libm), block_begin, block_end, block_stride, None, 0, None)
    return accum, dict(rays=ctr.rays, node_visits=ctr.node_visits, tri_tests=ctr.tri_tests)


def render_pass_tiles(sd, params, cam, lights, accum, block_begin, block_stride, nthreads=1, foreign_cap=65536):
    """One pass over this rank's blocks (block_begin, block_begin + stride, ...).  Returns (accum, foreign [n,8] float32, counters):
    foreign = with block_stride > 1 ALL out-of-block samples of the pass (they travel with the tiles), with block_stride == 1 those
    outside this call's blocks; in the device's ForeignSpill layout (r g b w, pixel, source block, sample, pad)."""
    desc, keep = make_scene_desc(sd)
    larr = (Light * len(lights))(*lights)
    ctr = Counters()
    foreign = np.zeros((foreign_cap, 8), np.float32)
    nf = C.c_int(0)
    lib().pto_render_pass2(C.byref(desc), C.byref(params), C.byref(cam), C.cast(larr, _P), len(lights), accum.ctypes.data,
                           C.byref(ctr), nthreads, 0, block_begin, -1, block_stride, foreign.ctypes.data, foreign_cap, C.byref(nf))
    return accum, foreign[: nf.value].copy(), dict(rays=ctr.rays, node_visits=ctr.node_visits, tri_tests=ctr.tri_tests)


def rng_key(k0, k1, k2, k3):
    """Key of the counter-based random stream seeded with four words (the stand-in for FW::Random(seed))."""
    return lib().pto_rng_key(int(k0), int(k1), int(k2), int(k3))


def trace_samples(sd, params, cam, lights, xy, sampler_base, rng_state, libm=False, nthreads=1):
    """tracePath / traceWhitted per sample -> (radiance [n,3], stream counters after the call [n] uint32)."""
    desc, keep = make_scene_desc(sd)
    xy, base = _f32(xy).reshape(-1, 2), np.ascontiguousarray(sampler_base, np.int32)
    rng = np.ascontiguousarray(rng_state, np.uint32).reshape(-1, 2)
    n = xy.shape[0]
    out = np.empty((n, 4), np.float32)
    larr = (Light * len(lights))(*lights)
    lib().pto_trace_samples(C.byref(desc), C.byref(params), C.byref(cam), C.cast(larr, _P), len(lights), n, xy.ctypes.data, base.ctypes.data,
                            rng.ctypes.data, int(libm), out.ctypes.data, nthreads)
    return out[:, :3].copy(), out[:, 3].copy().view(np.uint32)


def texture_parameters(sd, params, tri, u, v):
    desc, keep = make_scene_desc(sd)
    tri, u, v = np.ascontiguousarray(tri, np.int32), _f32(u), _f32(v)
    out = np.empty((tri.shape[0], 9), np.float32)
    lib().pto_texture_parameters(C.byref(desc), C.byref(params), tri.shape[0], tri.ctypes.data, u.ctypes.data, v.ctypes.data, out.ctypes.data)
    return out[:, 0:3].copy(), out[:, 3:6].copy(), out[:, 6:9].copy()


def light_to_sample(lights, pos, rng_state):
    pos, rng = _f32(pos).reshape(-1, 3), np.ascontiguousarray(rng_state, np.uint32).reshape(-1, 2)
    out = np.empty((pos.shape[0], 2), np.int32)
    larr = (Light * len(lights))(*lights)
    lib().pto_light_to_sample(C.cast(larr, _P), len(lights), pos.shape[0], pos.ctypes.data, rng.ctypes.data, out.ctypes.data)
    return out[:, 0].copy(), out[:, 1].copy()


def resolve(accum, libm=False):
    accum = _f32(accum)
    h, w = accum.shape[:2]
    out = np.empty_like(accum)
    lib().pto_resolve(accum.ctypes.data, w, h, out.ctypes.data, int(libm))
    return out


def math_probe(fn, x, y=None, libm=False):
    x = _f32(x)
    y = None if y is None else _f32(y)
    out = np.empty_like(x)
    lib().pto_math_probe({"sin": 0, "cos": 1, "exp": 2, "pow": 3}[fn], x.ctypes.data, None if y is None else y.ctypes.data,
                         out.ctypes.data, x.size, int(libm))
    return out
