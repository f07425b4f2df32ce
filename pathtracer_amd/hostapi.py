"""ctypes binding of libpthost.so: the C++ host mirror of the reference's classes (FW::BVH, RayTracer, AreaLight,
CameraControls, PathTraceRenderer) as exposed by host/hostapi.cpp.  libpthost.so links libptgpu.so; functions that trace
rays or render need a GPU, the builders and samplers do not."""
import ctypes as C
import os

import numpy as np

from .capi import FLATNODE_DTYPE, Camera, Light, Params, load_library

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpthost.so")
_P = C.c_void_p
_lib = None


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        load_library()  # libptgpu.so first: fail loudly if the HIP extension is missing
        if not os.path.exists(LIB_PATH):
            raise HostError("%s is not built: run __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.pth_last_error.restype = C.c_char_p
        L.pth_woop_from_positions.argtypes = [C.c_int, _P, _P]
        L.pth_bvh_build.restype = C.c_int
        L.pth_bvh_build.argtypes = [C.c_int, _P, C.c_int, C.c_int, _P, _P, _P]
        L.pth_camera_make.argtypes = [_P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]
        L.pth_light_make.argtypes = [_P, _P, _P, _P, C.c_float, C.POINTER(Light)]
        L.pth_light_sample.argtypes = [C.POINTER(Light), C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_float), _P]
        L.pth_map_to_disk.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.pth_form_basis.argtypes = [_P, _P]
        L.pth_sobol_sample.restype = C.c_float
        L.pth_sobol_sample.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.pth_random_u32.restype = C.c_uint32
        L.pth_random_u32.argtypes = [C.c_uint32] * 5
        L.pth_md5.argtypes = [_P, C.c_uint64, _P]
        L.pth_image_load.argtypes = [C.c_char_p, C.POINTER(_P), _P]
        L.pth_image_copy.argtypes = [_P, _P]
        L.pth_image_free.argtypes = [_P]
        L.pth_image_save.argtypes = [C.c_char_p, _P, C.c_int32, C.c_int32]
        L.pth_rt_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(_P)]
        L.pth_rt_close.argtypes = [_P]
        L.pth_rt_counts.argtypes = [_P, _P]
        L.pth_rt_get_bvh.argtypes = [_P, _P, _P]
        L.pth_rt_get_triangles.argtypes = [_P, _P, _P, _P, _P]
        L.pth_rt_num_materials.argtypes = [_P]
        L.pth_rt_material.argtypes = [_P, C.c_int, _P, _P, _P]
        L.pth_rt_save_hierarchy.argtypes = [_P, C.c_char_p]
        L.pth_rt_load_hierarchy.argtypes = [_P, C.c_char_p]
        L.pth_rt_construct_on_device.argtypes = [_P, C.c_int, C.c_int]
        L.pth_rt_mesh_md5.argtypes = [_P, _P]
        L.pth_rt_raycast.argtypes = [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P]
        L.pth_pt_render.argtypes = [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P, C.c_int, C.POINTER(Params), C.c_int,
                                    C.c_int, C.c_int, _P, _P, _P]
        L.pth_pt_render_stop.argtypes = [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P, C.c_int, C.POINTER(Params), C.c_int,
                                         _P, _P]
        L.pth_pt_statics.argtypes = [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P, C.c_int, C.POINTER(Params), C.c_int,
                                     _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]
        L.pth_pt_texture_parameters.argtypes = [_P, C.c_int, C.c_int, _P, _P, C.c_int64, _P]
        L.pth_rt_ray_count.argtypes = [_P]
        L.pth_rt_reset_ray_counter.argtypes = [_P]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(rc):
    if rc != 0:
        raise HostError(lib().pth_last_error().decode())


def woop_from_positions(pos):
    """tri_data for every triangle (reference src/RTTriangle.hpp:14-20) -> [n,12] rows (M | N)."""
    pos = _f32(pos).reshape(-1, 9)
    out = np.empty((pos.shape[0], 12), np.float32)
    lib().pth_woop_from_positions(pos.shape[0], pos.ctypes.data, out.ctypes.data)
    return out


def bvh_build(pos, sah=True, threads=0):
    """FW::BVH(triangles, sah) on the host (multi-threaded over subtrees). Returns (nodes, list, stats)."""
    pos = _f32(pos).reshape(-1, 9)
    n = pos.shape[0]
    nodes = np.zeros(max(2 * n, 1), dtype=FLATNODE_DTYPE)
    lst = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(6, dtype=np.int64)
    cnt = lib().pth_bvh_build(n, pos.ctypes.data, int(bool(sah)), threads, lst.ctypes.data, nodes.ctypes.data, st.ctypes.data)
    stats = dict(total_splits=int(st[0]), bad_splits=int(st[1]), depth_field=int(st[2]), true_depth=int(st[3]),
                 build_ms=st[4] / 1000.0, flatten_ms=st[5] / 1000.0, nodes=int(cnt))
    return nodes[:cnt].copy(), lst[:n].copy(), stats


def make_camera(eye=(0.0, 0.0, 0.48), forward=(0.0, -0.1, -1.0), up=(0.0, 1.0, 0.0), fov_deg=70.0, near=0.001, far=3.0,
                width=512, height=512):
    """Camera POD from FW::CameraControls; defaults = the probe camera of SURVEY.md 8d config C1."""
    cam = Camera()
    e, f, u = _f32(eye), _f32(forward), _f32(up)
    lib().pth_camera_make(e.ctypes.data, f.ctypes.data, u.ctypes.data, fov_deg, near, far, width, height, C.byref(cam))
    return cam


def make_light(position=(0.0, 0.9, 0.2), col0=(1.0, 0.0, 0.0), col1=(0.0, 0.0, -1.0), col2=(0.0, 1.0, 0.0), size=(0.25, 0.25),
               E=(100.0, 100.0, 100.0), intensity=1.0):
    """AreaLight POD from FW::AreaLight. Default: the SURVEY.md C1 light, normal = -col2 = straight down."""
    L = Light()
    p, o, s, e = _f32(position), _f32([col0, col1, col2]).reshape(-1), _f32(size), _f32(E)
    lib().pth_light_make(p.ctypes.data, o.ctypes.data, s.ctypes.data, e.ctypes.data, intensity, C.byref(L))
    return L


def light_sample(light, base, bounce, key, disk=False):
    k = np.asarray(key, np.uint32)
    pdf, p = C.c_float(), np.empty(3, np.float32)
    lib().pth_light_sample(C.byref(light), base, bounce, k.ctypes.data, int(disk), C.byref(pdf), p.ctypes.data)
    return pdf.value, p


def md5_hex(data):
    buf = np.ascontiguousarray(data)
    out = C.create_string_buffer(33)
    lib().pth_md5(buf.ctypes.data, buf.nbytes, out)
    return out.value.decode()


def load_image(path):
    """FW::importImage (host/ImageIO.cpp: the build's own PNG decoder, or binary PPM) -> uint8 array [h,w,4]."""
    h, wh = _P(), np.zeros(2, np.int32)
    _check(lib().pth_image_load(os.fspath(path).encode(), C.byref(h), wh.ctypes.data))
    try:
        out = np.empty((int(wh[1]), int(wh[0]), 4), np.uint8)
        lib().pth_image_copy(h, out.ctypes.data)
    finally:
        lib().pth_image_free(h)
    return out


def save_image(path, rgba):
    """FW::exportPNG / exportPPM of a resolved float picture [h,w,4] in [0,1] (8-bit RGB out)."""
    a = _f32(rgba)
    _check(lib().pth_image_save(os.fspath(path).encode(), a.ctypes.data, a.shape[1], a.shape[0]))


class HostRayTracer:
    """FW::RayTracer over a mesh file: App::constructTracer's triangle array + constructHierarchy."""

    def __init__(self, mesh_path, sah=True):
        h = _P()
        _check(lib().pth_rt_open(mesh_path.encode(), int(sah), C.byref(h)))
        self.h = h
        c = np.zeros(4, np.int32)
        lib().pth_rt_counts(h, c.ctypes.data)
        self.num_tris, self.num_nodes, self.list_size, self.depth_field = (int(x) for x in c)

    def close(self):
        if self.h:
            lib().pth_rt_close(self.h)
            self.h = None

    def bvh(self):
        c = np.zeros(4, np.int32)
        lib().pth_rt_counts(self.h, c.ctypes.data)
        nodes, lst = np.zeros(int(c[1]), FLATNODE_DTYPE), np.zeros(int(c[2]), np.int32)
        lib().pth_rt_get_bvh(self.h, lst.ctypes.data, nodes.ctypes.data)
        return nodes, lst

    def triangles(self):
        """App::constructTracer's triangle array: dict(pos [n,3,3], nrm [n,3,3], uv [n,3,2], mat_id [n])."""
        n = self.num_tris
        out = dict(pos=np.empty((n, 3, 3), np.float32), nrm=np.empty((n, 3, 3), np.float32), uv=np.empty((n, 3, 2), np.float32),
                   mat_id=np.empty(n, np.int32))
        lib().pth_rt_get_triangles(self.h, out["pos"].ctypes.data, out["nrm"].ctypes.data, out["uv"].ctypes.data, out["mat_id"].ctypes.data)
        return out

    def materials(self):
        out = []
        for i in range(lib().pth_rt_num_materials(self.h)):
            name, f, t = C.create_string_buffer(64), np.zeros(8, np.float32), np.zeros(4, np.int32)
            lib().pth_rt_material(self.h, i, name, f.ctypes.data, t.ctypes.data)
            out.append(dict(name=name.value.decode(), diffuse=tuple(f[:4]), specular=tuple(f[4:7]), glossiness=float(f[7]),
                            diffuse_map=tuple(int(x) for x in t[:2]), normal_map=tuple(int(x) for x in t[2:])))
        return out

    def save_hierarchy(self, path):
        _check(lib().pth_rt_save_hierarchy(self.h, path.encode()))

    def load_hierarchy(self, path):
        _check(lib().pth_rt_load_hierarchy(self.h, path.encode()))

    def construct_hierarchy_on_device(self, max_tris_in_leaf=3, keep_tree=True):
        """RayTracer::constructHierarchyOnDevice: the device-built linear BVH replaces the current hierarchy.  keep_tree=False: the tree is
        built together with the device scene and never copied to the host (no nodes / list to read back, save_hierarchy raises)."""
        _check(lib().pth_rt_construct_on_device(self.h, int(max_tris_in_leaf), int(bool(keep_tree))))
        c = np.zeros(4, np.int32)
        lib().pth_rt_counts(self.h, c.ctypes.data)
        self.num_tris, self.num_nodes, self.list_size, self.depth_field = (int(x) for x in c)

    def mesh_md5(self):
        out = C.create_string_buffer(33)
        lib().pth_rt_mesh_md5(self.h, out)
        return out.value.decode()

    def raycast(self, orig, direction, occlusion=False, single=False):
        o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        n = o.shape[0]
        out = dict(id=np.empty(n, np.int32), t=np.empty(n, np.float32), u=np.empty(n, np.float32), v=np.empty(n, np.float32),
                   point=np.empty((n, 3), np.float32))
        _check(lib().pth_rt_raycast(self.h, o.ctypes.data, d.ctypes.data, n, int(occlusion), int(single), out["id"].ctypes.data,
                                    out["t"].ctypes.data, out["u"].ctypes.data, out["v"].ctypes.data, out["point"].ctypes.data))
        return out

    def statics(self, params, lights, xy, base, seeds, advance, block_idx=None, eye=(0.0, 0.0, 0.48), forward=(0.0, -0.1, -1.0), up=(0.0, 1.0, 0.0),
                fov_deg=70.0, near=0.001, far=3.0):
        """PathTraceRenderer::tracePath / traceWhitted (by params.whitted), getLightToSample and pathTraceBlock called as the reference calls
        them, one sample per call.  Returns dict(rgb [n,3], ctr [n], light [n,2], block image or None)."""
        xy, base = _f32(xy).reshape(-1, 2), np.ascontiguousarray(base, np.int32)
        seeds, advance = np.ascontiguousarray(seeds, np.uint32).reshape(-1, 4), np.ascontiguousarray(advance, np.uint32)
        n, w, h = xy.shape[0], params.width, params.height
        rgb, ctr, light = np.empty((n, 3), np.float32), np.empty(n, np.uint32), np.empty((n, 2), np.int32)
        blk = np.empty((h, w, 4), np.float32) if block_idx is not None else None
        e, f, u = _f32(eye), _f32(forward), _f32(up)
        larr = (Light * len(lights))(*lights)
        _check(lib().pth_pt_statics(self.h, e.ctypes.data, f.ctypes.data, u.ctypes.data, fov_deg, near, far, w, h, C.cast(larr, _P), len(lights),
                                    C.byref(params), n, xy.ctypes.data, base.ctypes.data, seeds.ctypes.data, advance.ctypes.data, rgb.ctypes.data,
                                    ctr.ctypes.data, light.ctypes.data, None, -1 if block_idx is None else block_idx,
                                    None if blk is None else blk.ctypes.data))
        return dict(rgb=rgb, ctr=ctr, light=light, block=blk)

    def texture_parameters_of_hits(self, orig, direction, bilinear=True, normal_mapped=True):
        """PathTraceRenderer::getTextureParameters on the closest hits of n rays -> [n,9] (diffuse, normal, specular; zero rows = miss)."""
        o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        out = np.empty((o.shape[0], 9), np.float32)
        _check(lib().pth_pt_texture_parameters(self.h, int(bilinear), int(normal_mapped), o.ctypes.data, d.ctypes.data, o.shape[0], out.ctypes.data))
        return out

    def ray_count(self):
        return lib().pth_rt_ray_count(self.h)

    def reset_ray_counter(self):
        lib().pth_rt_reset_ray_counter(self.h)

    def path_trace(self, params, lights, passes=1, eye=(0.0, 0.0, 0.48), forward=(0.0, -0.1, -1.0), up=(0.0, 1.0, 0.0), fov_deg=70.0, near=0.001,
                   far=3.0, block_begin=0, block_stride=1):
        """FW::PathTraceRenderer::startPathTracingProcess ... checkFinish x passes ... updatePicture on this tracer.
        Returns (accumulation image, resolved display image, total rays, passes done)."""
        w, h = params.width, params.height
        accum, resolved = np.empty((h, w, 4), np.float32), np.empty((h, w, 4), np.float32)
        st = np.zeros(2, np.uint64)
        e, f, u = _f32(eye), _f32(forward), _f32(up)
        larr = (Light * len(lights))(*lights)
        _check(lib().pth_pt_render(self.h, e.ctypes.data, f.ctypes.data, u.ctypes.data, fov_deg, near, far, w, h, C.cast(larr, _P), len(lights),
                                   C.byref(params), passes, block_begin, block_stride, accum.ctypes.data, resolved.ctypes.data, st.ctypes.data))
        return accum, resolved, int(st[0]), int(st[1])

    def path_trace_then_stop(self, params, lights, stop_after_ms, eye=(0.0, 0.0, 0.48), forward=(0.0, -0.1, -1.0), up=(0.0, 1.0, 0.0), fov_deg=70.0,
                             near=0.001, far=3.0):
        """startPathTracingProcess, then FW::PathTraceRenderer::stop() after stop_after_ms (reference src/PathTraceRenderer.cpp:918-930).
        Returns (accumulation image as stop() left it, milliseconds stop() took, whether the first pass was still running)."""
        w, h = params.width, params.height
        accum = np.empty((h, w, 4), np.float32)
        ms = np.zeros(2, np.float64)
        e, f, u = _f32(eye), _f32(forward), _f32(up)
        larr = (Light * len(lights))(*lights)
        _check(lib().pth_pt_render_stop(self.h, e.ctypes.data, f.ctypes.data, u.ctypes.data, fov_deg, near, far, w, h, C.cast(larr, _P), len(lights),
                                        C.byref(params), int(stop_after_ms), accum.ctypes.data, ms.ctypes.data))
        return accum, float(ms[0]), bool(ms[1])
