"""CPU tests of the C++ host mirror (libpthost.so) and of the C-ABI library's surface (no compute calls without a GPU)."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

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


def test_md5_matches_the_reference_file(oracle):
    """The host mirror's MD5 (pathtracer_amd/host/Md5.hpp; names the hierarchy cache like RayTracer::computeMD5, reference
    src/RayTracer.cpp:57-66) against the reference's own src/Md5.c compiled where it lies (oracle/_ref, with the reference platform's
    32-bit `long`: oracle/ref_md5_shim.c says why), and the mesh digest the host RayTracer computes against MD5Buffer over the same bytes."""
    ref = oracle.ref()
    if ref is None or not hasattr(ref, "MD5Buffer"):
        pytest.skip("oracle/_ref not built (no /root/reference here); test_md5_is_standard covers the RFC vectors")
    import ctypes as C
    import struct
    rng = np.random.RandomState(3)
    blobs = [b"", b"abc", b"a" * 55, b"b" * 56, b"c" * 63, b"d" * 64, b"e" * 65, rng.bytes(100_003)]
    blobs.append(np.ascontiguousarray(scene.load_ptmesh()["pos"], np.float32).tobytes())  # what computeMD5 hashes: the vertex positions
    for blob in blobs:
        buf = C.create_string_buffer(blob, max(len(blob), 1))
        out = (C.c_uint * 4)()
        ref.MD5Buffer(buf, C.c_size_t(len(blob)), out)
        theirs = struct.pack("<4I", *out).hex()
        assert theirs == hashlib.md5(blob).hexdigest() == hostapi.md5_hex(np.frombuffer(blob, np.uint8)), len(blob)


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


def test_png_decoder_matches_pil_on_reference_maps(tmp_path):
    """host/ImageIO.cpp (own inflate + PNG unfilter) against PIL on the three maps the reference checkout holds for head.obj
    (assets/, byte copies made by tools/import_head_textures.py), and the stored-deflate writer round trip."""
    PIL = pytest.importorskip("PIL.Image")
    from pathtracer_amd import hostapi, scene
    for name in ("concrete1.png", "concrete1_normal.png", "concrete2_normal.png"):
        path = os.path.join(scene.HEAD_TEXTURE_DIR, name)
        got = hostapi.load_image(path)
        want = np.asarray(PIL.open(path).convert("RGBA"))
        assert got.shape == want.shape == (1600, 1600, 4)
        assert np.array_equal(got, want), name
    # writer: 8-bit RGB, readable by PIL and by the own reader; palette / grey+alpha / RGBA inputs through PIL-made files
    rng = np.random.RandomState(3)
    pic = rng.rand(37, 53, 4).astype(np.float32)
    out = tmp_path / "pic.png"
    hostapi.save_image(out, pic)
    q = (np.clip(pic[..., :3], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(np.asarray(PIL.open(out).convert("RGB")), q)
    assert np.array_equal(hostapi.load_image(out)[..., :3], q)
    for mode in ("P", "LA", "RGBA", "L"):
        f = tmp_path / ("m_%s.png" % mode)
        src = PIL.fromarray((rng.rand(19, 23, 4) * 255).astype(np.uint8), "RGBA").convert(mode)
        src.save(f)
        assert np.array_equal(hostapi.load_image(f), np.asarray(src.convert("RGBA"))), mode
    with pytest.raises(hostapi.HostError):
        bad = bytearray(open(out, "rb").read())
        bad[60] ^= 0xFF
        (tmp_path / "bad.png").write_bytes(bytes(bad))
        hostapi.load_image(tmp_path / "bad.png")


def test_head_textures_mix_reference_files_and_standins():
    from pathtracer_amd import scene
    mesh = scene.load_ptmesh()
    tex, index, labels = scene.head_textures(mesh, size=64)
    assert len(tex) == 6 and sorted(index) == sorted(["lambertian.png", "normal.png", "concrete2.png", "concrete2_normal.png",
                                                      "concrete1.png", "concrete1_normal.png"])
    assert sorted(n for n, l in labels.items() if l == "reference file") == ["concrete1.png", "concrete1_normal.png", "concrete2_normal.png"]
    assert tex[index["concrete1.png"]].shape == (1600, 1600, 4) and tex[index["lambertian.png"]].shape == (64, 64, 4)


def test_import_wavefront_mesh_small():
    """importWavefrontMesh (host/MeshIO.cpp; stands in for FW's importer behind App::loadMesh, reference src/App.cpp:754-794) and
    App::constructTracer's triangle array (:826-838) on tests/data/mini.obj: polygon fans, negative indices, v//vn and v/vt corners,
    faces without normals, one submesh per material in order of first use, map_Kd attached as PNG, a missing map_bump left empty."""
    from pathtracer_amd import hostapi
    rt = hostapi.HostRayTracer(os.path.join(os.path.dirname(__file__), "data", "mini.obj"), sah=True)
    try:
        t, mats = rt.triangles(), rt.materials()
        assert [m["name"] for m in mats] == ["checker", "plain"]
        assert np.allclose(mats[0]["diffuse"][:3], (0.5, 0.6, 0.7)) and np.allclose(mats[0]["specular"], 0.15) and mats[0]["glossiness"] == 50.0
        assert mats[0]["diffuse_map"] == (12, 8) and mats[0]["normal_map"] == (0, 0) and mats[1]["diffuse_map"] == (0, 0)
        # submesh 0 (checker): the quad as a fan (2 triangles) + the last face; submesh 1 (plain): two triangles
        assert t["mat_id"].tolist() == [0, 0, 0, 1, 1]
        V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1]], np.float32)
        assert np.array_equal(t["pos"][0], V[[0, 1, 2]]) and np.array_equal(t["pos"][1], V[[0, 2, 3]])
        assert np.array_equal(t["pos"][2], V[[4, 5, 2]])
        assert np.array_equal(t["pos"][3], V[[0, 1, 5]]) and np.array_equal(t["pos"][4], V[[4, 5, 0]])  # -2 -1 1 = vertices 5 6 1
        assert np.array_equal(t["uv"][0], np.array([[0, 0], [2, 0], [2, 3]], np.float32))
        assert np.array_equal(t["nrm"][0], np.tile(np.array([0, 0, 1], np.float32), (3, 1)))
        # faces without vn get the geometric normal: triangle 4 lies in the y = 0 plane
        assert np.allclose(np.abs(t["nrm"][4]), np.tile(np.array([0, 1, 0], np.float32), (3, 1)))
        assert rt.num_nodes >= 1 and rt.list_size == 5
    finally:
        rt.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/head/head.obj"), reason="build container only: reads the reference's head.obj")
def test_import_wavefront_head_obj_equals_ptmesh():
    """The reference's head/head.obj through importWavefrontMesh gives the per-triangle arrays assets/head.ptmesh holds
    (tools/convert_head_obj.py wrote that file with an independent Python reader), so GPU-box tests that load the .ptmesh render the
    mesh a maintainer gets from the .obj."""
    from pathtracer_amd import hostapi, scene
    rt = hostapi.HostRayTracer("/root/reference/head/head.obj", sah=True)
    try:
        t, mesh = rt.triangles(), scene.load_ptmesh()
        assert rt.num_tris == 17692 and [m["name"] for m in rt.materials()] == [m["name"] for m in mesh["materials"]]
        for k in ("pos", "nrm", "uv"):
            assert np.array_equal(t[k].view(np.uint32), mesh[k].view(np.uint32)), k
        assert np.array_equal(t["mat_id"], mesh["mat_id"])
        maps = {m["name"]: (m["diffuse_map"], m["normal_map"]) for m in rt.materials()}
        assert maps["floor"] == ((1600, 1600), (1600, 1600)) and maps["wall"] == ((0, 0), (1600, 1600)) and maps["head"] == ((0, 0), (0, 0))
    finally:
        rt.close()
