"""CPU tests of the C++ host mirror (libpthost.so) and of the C-ABI library's surface (no compute calls without a GPU)."""
import ctypes as C
import hashlib
import os
import re

import numpy as np

from pathtracer_amd import capi, hostapi, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    """libptgpu.so loads and exports every function include/ptgpu.h declares; the ctypes table covers them all."""
    hdr = open(os.path.join(ROOT, "include", "ptgpu.h")).read()
    declared = set(re.findall(r"\b(ptgpu_[a-z0-9_]+)\s*\(", hdr))
    lib = capi.load_library()
    bound = {name for name, _r, _a in capi.SIGNATURES}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    # and nothing beyond the header is exported under the library's prefix
    import subprocess
    path = os.environ.get("PTGPU_LIB") or os.path.join(ROOT, "pathtracer_amd", "libptgpu.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in syms.splitlines() if " T ptgpu_" in ln}
    assert exported == declared, exported ^ declared


def test_cabi_struct_layouts_match_header():
    assert C.sizeof(capi.FlatNode) == 64 and C.sizeof(capi.Material) == 48 and C.sizeof(capi.Light) == 92
    assert C.sizeof(capi.Camera) == 104 and C.sizeof(capi.Params) == 72 and C.sizeof(capi.Counters) == 24


def test_no_gpu_init_fails_loudly_not_silently():
    """Without a device the library reports an error code and message; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    rc = capi.load_library().ptgpu_init(0, C.byref(h))
    assert rc < 0 and capi.load_library().ptgpu_last_error()


def test_parallel_bvh_builder_equals_serial_restatement(head, soup, oracle):
    """FW::BVH built on 1..8 host threads is byte-identical (nodes and list) to the oracle's serial restatement."""
    for fx, sah in ((head, True), (head, False), (soup, True), (soup, False)):
        on, ol, _ = oracle.bvh_build(fx["mesh"]["pos"], sah)
        for threads in (1, 3, 8):
            hn, hl, st = hostapi.bvh_build(fx["mesh"]["pos"], sah, threads)
            assert hn.tobytes() == on.tobytes() and np.array_equal(hl, ol)
    hn, hl, st = hostapi.bvh_build(head["mesh"]["pos"], True, 8)
    assert (st["nodes"], st["bad_splits"], st["total_splits"], st["true_depth"]) == (15619, 685, 7809, 22)


def test_bvh_degenerate_inputs(oracle):
    """1..4 triangles, all-identical triangles (SAH refuses to split => one big leaf), collinear centroids."""
    base = scene.grid_scene(1, 1, seed=9)
    for n in (1, 2, 3, 4):
        hn, hl, _ = hostapi.bvh_build(base[:n], True, 2)
        on, ol, _ = oracle.bvh_build(base[:n], True)
        assert hn.tobytes() == on.tobytes() and np.array_equal(hl, ol)
    same = np.repeat(base[:1], 40, axis=0)
    hn, hl, st = hostapi.bvh_build(same, True, 2)
    on, ol, ost = oracle.bvh_build(same, True)
    assert hn.tobytes() == on.tobytes() and len(hn) == 1 and hn[0]["leaf"] == 1 and hn[0]["triangles"] == 40
    line = np.stack([base[0] + np.float32([i * 0.1, 0, 0]) for i in range(33)])
    for sah in (True, False):
        hn, hl, _ = hostapi.bvh_build(line, sah, 4)
        on, ol, _ = oracle.bvh_build(line, sah)
        assert hn.tobytes() == on.tobytes() and np.array_equal(hl, ol)


def test_host_woop_sobol_random_light_match_oracle(head, oracle):
    assert np.array_equal(hostapi.woop_from_positions(head["mesh"]["pos"]).view(np.uint32), head["woop"].view(np.uint32))
    L, H = oracle.lib(), hostapi.lib()
    rng = np.random.RandomState(5)
    for idx, dim in zip(rng.randint(0, 2 ** 31, 500), rng.randint(0, 1024, 500)):
        assert H.pth_sobol_sample(int(idx), int(dim), 0) == oracle.sobol(int(idx), int(dim))
    for k in rng.randint(0, 2 ** 31, (50, 4)):
        for draw in (0, 1, 7):
            assert H.pth_random_u32(*[int(x) for x in k], draw) == L.pto_rng_u32(*[int(x) for x in k], draw)
    light = scene.make_light(position=(0.1, 0.9, 0.2), size=(0.25, 0.4))
    for base, bounce in ((1234, 0), (99999, 3), (7, 300)):
        key = np.array([1, 2, base, bounce], np.uint32)
        pdf, p = hostapi.light_sample(light, base, bounce, key)
        pdf2, p2 = C.c_float(), np.empty(3, np.float32)
        L.pto_light_sample(C.byref(light), base, bounce, key.ctypes.data, C.byref(pdf2), p2.ctypes.data)
        assert pdf == pdf2.value == np.float32(1.0) / (np.float32(4.0) * np.float32(0.25) * np.float32(0.4))
        assert np.array_equal(p.view(np.uint32), p2.view(np.uint32))
        assert abs(p[1] - 0.9) < 1e-6 and abs(p[0] - 0.1) <= 0.25 + 1e-6  # on the light's rectangle


def test_host_small_functions_match_oracle(oracle):
    L, H = oracle.lib(), hostapi.lib()
    rng = np.random.RandomState(6)
    x1, y1, x2, y2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for a, b in rng.uniform(-1, 1, (300, 2)).astype(np.float32):
        H.pth_map_to_disk(a, b, C.byref(x1), C.byref(y1))
        L.pto_map_to_disk(a, b, C.byref(x2), C.byref(y2), 1)
        assert (x1.value, y1.value) == (x2.value, y2.value)
    for n in rng.normal(size=(100, 3)).astype(np.float32):
        n = (n / np.linalg.norm(n)).astype(np.float32)
        a, b = np.empty(9, np.float32), np.empty(9, np.float32)
        H.pth_form_basis(n.ctypes.data, a.ctypes.data)
        L.pto_form_basis(n.ctypes.data, b.ctypes.data)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        R = a.reshape(3, 3)  # columns stored consecutively
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5) and np.allclose(R[2], n)


def test_md5_is_standard():
    for blob in (b"", b"abc", bytes(range(256)) * 5, b"x" * 55, b"y" * 56, b"z" * 64):
        assert hostapi.md5_hex(np.frombuffer(blob, np.uint8)) == hashlib.md5(blob).hexdigest()


def test_camera_pod_unprojects_pixel_centres(oracle):
    """invP from FW::CameraControls: the centre pixel's ray points along the camera's forward axis and starts in front
    of the eye; rays end on the far plane (segment semantics of reference src/PathTraceRenderer.cpp:251-266)."""
    cam = scene.make_camera(eye=(0.0, 0.0, 0.48), forward=(0.0, 0.0, -1.0), width=65, height=65, near=0.001, far=3.0)
    o, d = oracle.gen_primary(cam, 65, 65)
    c = 32 * 65 + 32
    assert np.allclose(d[c] / np.linalg.norm(d[c]), [0, 0, -1], atol=1e-5)
    assert abs(o[c][2] - (0.48 - 2 * 0.001 * 3.0 / 3.001)) < 1e-5
    assert np.allclose((o + d)[:, 2], 0.48 - 3.0, atol=2e-3)
    wide = scene.make_camera(eye=(0.0, 0.0, 0.48), forward=(0.0, 0.0, -1.0), width=130, height=65)
    ow, dw = oracle.gen_primary(wide, 130, 65)
    # fitToView keeps the vertical field of view and widens the horizontal one
    assert np.allclose(dw[32 * 130 + 65 + 32] / np.linalg.norm(dw[32 * 130 + 65 + 32]), d[32 * 65 + 64] / np.linalg.norm(d[32 * 65 + 64]), atol=2e-2)
