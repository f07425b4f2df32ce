"""ctypes binding of include/ptgpu.h (libptgpu.so, the gfx950 kernels behind a C ABI).

This is plumbing for tests and bench.py; the product's host side is the C++ mirror of the
reference's classes in pathtracer_amd/host/.  There is NO CPU fallback: if the library is
missing or a call fails, PtgpuError is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTGPU_LIB", os.path.join(_HERE, "libptgpu.so"))  # PTGPU_LIB: A/B builds of the same ABI


class PtgpuError(RuntimeError):
    pass


class FlatNode(C.Structure):  # reference src/BvhNode.hpp:67-113
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("startIdx", C.c_int32), ("endIdx", C.c_int32),
                ("leaf", C.c_uint8), ("_padleaf", C.c_uint8 * 3), ("rightChild", C.c_int32), ("axis", C.c_int32),
                ("leftChild", C.c_int32), ("triangles", C.c_int32), ("pad0", C.c_int32), ("pad1", C.c_int32),
                ("pad2", C.c_int32)]


FLATNODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("startIdx", "<i4"), ("endIdx", "<i4"),
                           ("leaf", "u1"), ("_padleaf", "u1", 3), ("rightChild", "<i4"), ("axis", "<i4"),
                           ("leftChild", "<i4"), ("triangles", "<i4"), ("pad0", "<i4"), ("pad1", "<i4"), ("pad2", "<i4")])
assert C.sizeof(FlatNode) == 64 and FLATNODE_DTYPE.itemsize == 64


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 4), ("specular", C.c_float * 3), ("glossiness", C.c_float),
                ("tex_diffuse", C.c_int32), ("tex_alpha", C.c_int32), ("tex_normal", C.c_int32), ("tex_specular", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.c_void_p)]


class Light(C.Structure):  # reference src/AreaLight.hpp:74-79
    _fields_ = [("xform", C.c_float * 16), ("size", C.c_float * 2), ("E", C.c_float * 3), ("intensity", C.c_float),
                ("power", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("invP", C.c_float * 16), ("forward", C.c_float * 3), ("orient_col0", C.c_float * 3),
                ("orient_col1", C.c_float * 3), ("far_", C.c_float)]


class Params(C.Structure):  # reference src/PathTraceRenderer.cpp:15-28
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("num_aa_rays", C.c_int32),
                ("sobol_block", C.c_int32), ("bilinear", C.c_int32), ("normal_mapped", C.c_int32),
                ("depth_of_field", C.c_int32), ("whitted", C.c_int32), ("num_light_rays", C.c_int32),
                ("use_alpha", C.c_int32), ("pass_seed", C.c_uint32), ("aperture", C.c_float), ("attenuation", C.c_float),
                ("filter_width", C.c_float), ("focal_distance", C.c_float), ("intensity", C.c_float),
                ("termination_prob", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("num_tris", C.c_int32), ("pos", C.c_void_p), ("nrm", C.c_void_p), ("uv", C.c_void_p),
                ("mat_id", C.c_void_p), ("woop", C.c_void_p), ("num_materials", C.c_int32), ("materials", C.c_void_p),
                ("num_textures", C.c_int32), ("textures", C.c_void_p), ("num_nodes", C.c_int32), ("nodes", C.c_void_p),
                ("list_size", C.c_int32), ("list", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64)]


# every symbol include/ptgpu.h declares: (name, restype, argtypes)
_P = C.c_void_p
SIGNATURES = [
    ("ptgpu_init", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("ptgpu_shutdown", C.c_int, [_P]),
    ("ptgpu_last_error", C.c_char_p, []),
    ("ptgpu_set_stream", C.c_int, [_P, _P]),
    ("ptgpu_sync", C.c_int, [_P]),
    ("ptgpu_query", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("ptgpu_malloc", C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    ("ptgpu_free", C.c_int, [_P, _P]),
    ("ptgpu_mem_info", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("ptgpu_memcpy_h2d", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("ptgpu_memcpy_d2h", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("ptgpu_memset", C.c_int, [_P, _P, C.c_int, C.c_uint64]),
    ("ptgpu_enable_counters", C.c_int, [_P, C.c_int]),
    ("ptgpu_get_counters", C.c_int, [_P, C.POINTER(Counters)]),
    ("ptgpu_reset_counters", C.c_int, [_P]),
    ("ptgpu_scene_create", C.c_int, [_P, C.POINTER(SceneDesc), C.POINTER(_P)]),
    ("ptgpu_scene_create_lbvh", C.c_int, [_P, C.POINTER(SceneDesc), C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32)]),
    ("ptgpu_scene_destroy", C.c_int, [_P]),
    ("ptgpu_scene_stats", C.c_int, [_P, C.POINTER(C.c_uint64 * 4), C.POINTER(C.c_int32)]),
    ("ptgpu_scene_ray_length", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    ("ptgpu_raycast_batch", C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    ("ptgpu_raycast_batch_dev", C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    ("ptgpu_raycast_batch_dev_noalloc", C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    ("ptgpu_trace_frame_dev", C.c_int, [_P, C.POINTER(Camera), C.POINTER(Light), C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    ("ptgpu_gen_primary_dev", C.c_int, [_P, C.POINTER(Camera), C.c_int32, C.c_int32, _P, _P]),
    ("ptgpu_gen_shadow_dev", C.c_int, [_P, C.POINTER(Light), C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("ptgpu_render_pass_dev", C.c_int, [_P, C.POINTER(Params), C.POINTER(Camera), _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    ("ptgpu_render_pass", C.c_int, [_P, C.POINTER(Params), C.POINTER(Camera), _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    ("ptgpu_cancel", C.c_int, [_P]),
    ("ptgpu_trace_samples_dev", C.c_int, [_P, C.POINTER(Params), C.POINTER(Camera), _P, C.c_int32, C.c_int64, _P, _P, _P, _P]),
    ("ptgpu_trace_samples", C.c_int, [_P, C.POINTER(Params), C.POINTER(Camera), _P, C.c_int32, C.c_int64, _P, _P, _P, _P]),
    ("ptgpu_texture_parameters_batch", C.c_int, [_P, C.POINTER(Params), C.c_int64, _P, _P, _P, _P]),
    ("ptgpu_light_to_sample_batch", C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    ("ptgpu_sobol_sample_batch", C.c_int, [_P, C.c_int64, _P, _P, _P]),
    ("ptgpu_resolve_dev", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    ("ptgpu_spill_status", C.c_int, [_P, C.POINTER(C.c_uint64 * 3)]),
    ("ptgpu_comm_create_id", C.c_int, [_P]),
    ("ptgpu_comm_init", C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    ("ptgpu_comm_destroy", C.c_int, [_P]),
    ("ptgpu_gather_tiles_dev", C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    ("ptgpu_tiles_slice_floats", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    ("ptgpu_tiles_pack_dev", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("ptgpu_tiles_unpack_dev", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("ptgpu_intersect_triangle1_batch", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, _P, _P, _P]),
    ("ptgpu_bvh_build_lbvh", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
]

NOALLOC_MAX_RAYS = 4194304  # PTGPU_NOALLOC_MAX_RAYS
COMM_ID_BYTES = 128  # PTGPU_COMM_ID_BYTES

_lib = None


def load_library(path=LIB_PATH):
    """Load libptgpu.so and bind every declared symbol.  Raises PtgpuError if the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise PtgpuError("HIP extension %s is not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    lib = C.CDLL(path)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def create_comm_id():
    """Rank 0: a fresh communicator id (bytes) to hand to every rank's Context.comm_init."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(load_library().ptgpu_comm_create_id(C.cast(buf, _P)))
    return bytes(buf)


def tiles_slice_floats(width, height, world):
    n = C.c_int64()
    _check(load_library().ptgpu_tiles_slice_floats(width, height, world, C.byref(n)))
    return n.value


def _check(rc):
    if rc != 0:
        raise PtgpuError("ptgpu error %d: %s" % (rc, load_library().ptgpu_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


class Context:
    def __init__(self, device=0):
        self.lib = load_library()
        h = _P()
        _check(self.lib.ptgpu_init(device, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.ptgpu_shutdown(self.handle)
            self.handle = None

    def set_stream(self, stream_ptr):
        _check(self.lib.ptgpu_set_stream(self.handle, stream_ptr))

    def sync(self):
        _check(self.lib.ptgpu_sync(self.handle))

    def cancel(self):
        """Cancel every render pass enqueued so far (PathTraceRenderer::stop() mid-pass)."""
        _check(self.lib.ptgpu_cancel(self.handle))

    def malloc(self, nbytes):
        p = _P()
        _check(self.lib.ptgpu_malloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def mem_info(self):
        """(free, total) bytes of device memory on the context's GPU."""
        f, t = C.c_uint64(), C.c_uint64()
        _check(self.lib.ptgpu_mem_info(self.handle, C.byref(f), C.byref(t)))
        return f.value, t.value

    def free(self, p):
        _check(self.lib.ptgpu_free(self.handle, p))

    def h2d(self, dev, arr):
        arr = np.ascontiguousarray(arr)
        _check(self.lib.ptgpu_memcpy_h2d(self.handle, dev, arr.ctypes.data, arr.nbytes))

    def d2h(self, dev, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        _check(self.lib.ptgpu_memcpy_d2h(self.handle, out.ctypes.data, dev, out.nbytes))
        return out

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self.h2d(p, arr)
        return p

    def enable_counters(self, on):
        _check(self.lib.ptgpu_enable_counters(self.handle, 1 if on else 0))

    def reset_counters(self):
        _check(self.lib.ptgpu_reset_counters(self.handle))

    def counters(self):
        c = Counters()
        _check(self.lib.ptgpu_get_counters(self.handle, C.byref(c)))
        return dict(rays=c.rays, node_visits=c.node_visits, tri_tests=c.tri_tests)

    def spill_status(self):
        """Out-of-block samples: dict(spilled = in the last Sobol-block pass, dropped = lost to a full list since the last call of this
        method (cleared by it), foreign_pending = waiting for the next tile exchange)."""
        o = (C.c_uint64 * 3)()
        _check(self.lib.ptgpu_spill_status(self.handle, C.byref(o)))
        return dict(spilled=o[0], dropped=o[1], foreign_pending=o[2])

    # ---- image tiles across GPUs (include/ptgpu.h "image tiles") ----
    def comm_init(self, world, rank, comm_id):
        """comm_id: the COMM_ID_BYTES bytes rank 0 got from create_comm_id(), the same on every rank (collective)."""
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        _check(self.lib.ptgpu_comm_init(self.handle, world, rank, C.cast(buf, _P)))

    def comm_destroy(self):
        _check(self.lib.ptgpu_comm_destroy(self.handle))

    def gather_tiles_dev(self, width, height, d_accum):
        _check(self.lib.ptgpu_gather_tiles_dev(self.handle, width, height, d_accum))

    def tiles_pack_dev(self, width, height, world, rank, d_accum, d_slice):
        _check(self.lib.ptgpu_tiles_pack_dev(self.handle, width, height, world, rank, d_accum, d_slice))

    def tiles_unpack_dev(self, width, height, world, rank, d_slices, d_accum):
        _check(self.lib.ptgpu_tiles_unpack_dev(self.handle, width, height, world, rank, d_slices, d_accum))

    def light_to_sample(self, lights, pos, rng_state):
        """getLightToSample for n positions -> (light index [n], stream counter after the call [n])."""
        pos, rng = _f32(pos).reshape(-1, 3), np.ascontiguousarray(rng_state, np.uint32).reshape(-1, 2)
        out = np.empty((pos.shape[0], 2), np.int32)
        larr = (Light * len(lights))(*lights)
        _check(self.lib.ptgpu_light_to_sample_batch(self.handle, C.cast(larr, _P), len(lights), pos.shape[0], pos.ctypes.data, rng.ctypes.data,
                                                    out.ctypes.data))
        return out[:, 0].copy(), out[:, 1].copy()

    def sobol_sample(self, index, dimension):
        """sobol::sample(index, dimension) on the device for arrays of (index uint64, dimension uint32) -> float32."""
        idx, dim = np.ascontiguousarray(index, np.uint64), np.ascontiguousarray(dimension, np.uint32)
        out = np.empty(idx.shape[0], np.float32)
        _check(self.lib.ptgpu_sobol_sample_batch(self.handle, idx.shape[0], idx.ctypes.data, dim.ctypes.data, out.ctypes.data))
        return out

    def resolve_dev(self, d_accum, width, height, d_out):
        _check(self.lib.ptgpu_resolve_dev(self.handle, d_accum, width, height, d_out))

    def resolve(self, accum):
        """updatePicture on a host accumulation image [h,w,4] -> resolved [h,w,4]."""
        a = np.ascontiguousarray(accum, dtype=np.float32)
        h, w = a.shape[:2]
        d_in, d_out = self.to_device(a), self.malloc(a.nbytes)
        try:
            self.resolve_dev(d_in, w, h, d_out)
            return self.d2h(d_out, a.shape, np.float32)
        finally:
            self.free(d_in)
            self.free(d_out)

    def intersect_triangle1(self, orig, direction, v0, v1, v2):
        """rtIntersect.inl intersect_triangle1 for n (ray, triangle) pairs -> (hit int32[n], t, u, v float32[n])."""
        a = [_f32(x).reshape(-1, 3) for x in (orig, direction, v0, v1, v2)]
        n = a[0].shape[0]
        hit, t, u, v = np.empty(n, np.int32), np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
        _check(self.lib.ptgpu_intersect_triangle1_batch(self.handle, *(x.ctypes.data for x in a), n, hit.ctypes.data, t.ctypes.data,
                                                        u.ctypes.data, v.ctypes.data))
        return hit, t, u, v

    def bvh_build_lbvh(self, pos, max_tris_in_leaf=3):
        """Device-side linear BVH over triangle positions [n,3,3] -> (nodes FLATNODE_DTYPE[num_nodes], list int32[n], depth), in the
        reference's layout (feed them to Scene like a tree from the CPU builders)."""
        pos = _f32(pos).reshape(-1, 3, 3)
        n = pos.shape[0]
        nodes = np.zeros(max(2 * n, 1), dtype=FLATNODE_DTYPE)
        lst = np.zeros(max(n, 1), dtype=np.int32)
        num, depth = C.c_int32(0), C.c_int32(0)
        _check(self.lib.ptgpu_bvh_build_lbvh(self.handle, pos.ctypes.data, n, max_tris_in_leaf, nodes.ctypes.data, C.byref(num), lst.ctypes.data,
                                             C.byref(depth)))
        return nodes[: num.value].copy(), lst[:n].copy(), depth.value

    def gen_primary_dev(self, cam, width, height, d_orig, d_dir):
        _check(self.lib.ptgpu_gen_primary_dev(self.handle, C.byref(cam), width, height, d_orig, d_dir))


def make_scene_desc(sd):
    """sd: dict with numpy arrays pos[n,3,3] nrm[n,3,3] uv[n,3,2] mat_id[n] woop[n,12] materials (list of Material)
    textures (list of uint8 arrays [h,w,4]) nodes (FLATNODE_DTYPE array) list (int32).  Returns (SceneDesc, keepalive)."""
    keep = {}
    n = int(sd["mat_id"].shape[0])
    keep["pos"] = _f32(sd["pos"]); keep["nrm"] = _f32(sd["nrm"]); keep["uv"] = _f32(sd["uv"]); keep["woop"] = _f32(sd["woop"])
    keep["mat_id"] = np.ascontiguousarray(sd["mat_id"], dtype=np.int32)
    mats = sd["materials"]
    keep["materials"] = (Material * max(len(mats), 1))(*mats)
    texs = [np.ascontiguousarray(t, dtype=np.uint8) for t in sd.get("textures", [])]
    keep["texels"] = texs
    tarr = (Texture * max(len(texs), 1))()
    for i, t in enumerate(texs):
        tarr[i].width, tarr[i].height, tarr[i].rgba = t.shape[1], t.shape[0], t.ctypes.data
    keep["textures"] = tarr
    keep["nodes"] = np.ascontiguousarray(sd["nodes"], dtype=FLATNODE_DTYPE)
    keep["list"] = np.ascontiguousarray(sd["list"], dtype=np.int32)
    d = SceneDesc()
    d.num_tris = n
    d.pos, d.nrm, d.uv, d.mat_id, d.woop = (_ptr(keep[k]) for k in ("pos", "nrm", "uv", "mat_id", "woop"))
    d.num_materials, d.materials = len(mats), C.cast(keep["materials"], _P)
    d.num_textures, d.textures = len(texs), C.cast(tarr, _P)
    d.num_nodes, d.nodes = int(keep["nodes"].shape[0]), _ptr(keep["nodes"])
    d.list_size, d.list = int(keep["list"].shape[0]), _ptr(keep["list"])
    return d, keep


class Scene:
    def __init__(self, ctx, sd, lbvh_max_tris_in_leaf=None):
        """sd: scene dict (triangle arrays, materials, textures, nodes, list).  lbvh_max_tris_in_leaf: build the hierarchy on the device
        instead (ptgpu_scene_create_lbvh; sd's nodes / list are ignored) -- num_nodes_built then holds the built tree's node count."""
        self.ctx, self.lib = ctx, ctx.lib
        h = _P()
        self.num_nodes_built = None
        if lbvh_max_tris_in_leaf is None:
            desc, keep = make_scene_desc(sd)
            _check(self.lib.ptgpu_scene_create(ctx.handle, C.byref(desc), C.byref(h)))
        else:
            sd2 = dict(sd)
            sd2["nodes"], sd2["list"] = np.zeros(0, FLATNODE_DTYPE), np.zeros(0, np.int32)
            desc, keep = make_scene_desc(sd2)
            desc.nodes, desc.list = None, None
            nn = C.c_int32()
            _check(self.lib.ptgpu_scene_create_lbvh(ctx.handle, C.byref(desc), int(lbvh_max_tris_in_leaf), C.byref(h), C.byref(nn)))
            self.num_nodes_built = nn.value
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.ptgpu_scene_destroy(self.handle)
            self.handle = None

    def stats(self):
        b, d = (C.c_uint64 * 4)(), C.c_int32()
        _check(self.lib.ptgpu_scene_stats(self.handle, C.byref(b), C.byref(d)))
        return dict(inner_bytes=b[0], tri_bytes=b[1], shading_bytes=b[2], texture_bytes=b[3], depth=d.value)

    def ray_length(self):
        """dict(steps_per_ray, service_threshold): the scene probe behind the path-trace passes' scheduling (include/ptgpu.h)."""
        st, thr = C.c_float(), C.c_int32()
        _check(self.lib.ptgpu_scene_ray_length(self.handle, C.byref(st), C.byref(thr)))
        return dict(steps_per_ray=st.value, service_threshold=thr.value)

    def raycast(self, orig, direction, occlusion=False, use_alpha=True, want=("id", "t", "u", "v")):
        """Host-pointer RayTracer::raycast batch. Returns dict of numpy arrays."""
        o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        n = o.shape[0]
        out = {k: np.empty(n, dtype=np.int32 if k == "id" else np.float32) for k in want}
        _check(self.lib.ptgpu_raycast_batch(self.handle, int(use_alpha), o.ctypes.data, d.ctypes.data, n, int(occlusion),
                                            _ptr(out.get("id")), _ptr(out.get("t")), _ptr(out.get("u")), _ptr(out.get("v"))))
        return out

    def raycast_dev(self, d_orig, d_dir, n, occlusion, d_id, d_t, d_u, d_v, use_alpha=True):
        _check(self.lib.ptgpu_raycast_batch_dev(self.handle, int(use_alpha), d_orig, d_dir, n, int(occlusion), d_id, d_t, d_u, d_v))

    def raycast_dev_noalloc(self, d_orig, d_dir, n, occlusion, d_id, d_t, d_u, d_v, use_alpha=True):
        """The allocation-free variant (fixed scratch inside the library, n <= NOALLOC_MAX_RAYS)."""
        _check(self.lib.ptgpu_raycast_batch_dev_noalloc(self.handle, int(use_alpha), d_orig, d_dir, n, int(occlusion), d_id, d_t, d_u, d_v))

    def trace_frame_dev(self, cam, light, width, height, d_id, d_t, d_u, d_v, d_shadow_id):
        """Primary closest-hit + shadow any-hit rays of a whole frame in one launch (BASELINE configs[1])."""
        _check(self.lib.ptgpu_trace_frame_dev(self.handle, C.byref(cam), C.byref(light), width, height, d_id, d_t, d_u, d_v, d_shadow_id))

    def gen_shadow_dev(self, light, n, d_orig, d_dir, d_id, d_t, d_u, d_v, d_so, d_sd, d_valid):
        _check(self.lib.ptgpu_gen_shadow_dev(self.handle, C.byref(light), n, d_orig, d_dir, d_id, d_t, d_u, d_v, d_so, d_sd, d_valid))

    def trace_samples(self, params, cam, lights, xy, sampler_base, rng_state):
        """tracePath / traceWhitted for n arbitrary samples -> (radiance [n,3] float32, stream counters after the call [n] uint32)."""
        xy, base = _f32(xy).reshape(-1, 2), np.ascontiguousarray(sampler_base, np.int32)
        rng = np.ascontiguousarray(rng_state, np.uint32).reshape(-1, 2)
        n = xy.shape[0]
        out = np.empty((n, 4), np.float32)
        larr = (Light * len(lights))(*lights)
        _check(self.lib.ptgpu_trace_samples(self.handle, C.byref(params), C.byref(cam), C.cast(larr, _P), len(lights), n, xy.ctypes.data,
                                            base.ctypes.data, rng.ctypes.data, out.ctypes.data))
        return out[:, :3].copy(), out[:, 3].copy().view(np.uint32)

    def texture_parameters(self, params, tri, u, v):
        """getTextureParameters for n hits -> (diffuse [n,3], normal [n,3], specular [n,3])."""
        tri, u, v = np.ascontiguousarray(tri, np.int32), _f32(u), _f32(v)
        out = np.empty((tri.shape[0], 9), np.float32)
        _check(self.lib.ptgpu_texture_parameters_batch(self.handle, C.byref(params), tri.shape[0], tri.ctypes.data, u.ctypes.data, v.ctypes.data,
                                                       out.ctypes.data))
        return out[:, 0:3].copy(), out[:, 3:6].copy(), out[:, 6:9].copy()

    def render_pass(self, params, cam, lights, accum=None, block_begin=0, block_end=-1, block_stride=1):
        """One pathTraceBlock pass over the selected blocks, accumulated into (a copy of) accum [h,w,4] on the host."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 4), np.float32)
        accum = np.ascontiguousarray(accum, dtype=np.float32)
        larr = (Light * len(lights))(*lights)
        _check(self.lib.ptgpu_render_pass(self.handle, C.byref(params), C.byref(cam), C.cast(larr, _P), len(lights), block_begin,
                                          block_end, block_stride, accum.ctypes.data))
        return accum

    def render_pass_dev(self, params, cam, lights, d_accum, block_begin=0, block_end=-1, block_stride=1):
        larr = (Light * len(lights))(*lights)
        _check(self.lib.ptgpu_render_pass_dev(self.handle, C.byref(params), C.byref(cam), C.cast(larr, _P), len(lights), block_begin,
                                              block_end, block_stride, d_accum))
