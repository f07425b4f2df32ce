"""GPU tests of the C++ host classes: FW::RayTracer::raycast / raycastBatch / hierarchy cache run on the device through the
C ABI and agree bit-for-bit with the oracle."""
import os

import numpy as np
import pytest

from pathtracer_amd import hostapi, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_rt():
    rt = hostapi.HostRayTracer(scene.HEAD_MESH, sah=True)
    yield rt
    rt.close()


def test_host_raytracer_matches_oracle(host_rt, head, oracle):
    assert (host_rt.num_tris, host_rt.num_nodes) == (17692, 15619)
    nodes, lst = host_rt.bvh()
    assert nodes.tobytes() == head["nodes"].tobytes() and np.array_equal(lst, head["list"])
    cam = scene.make_camera(width=320, height=180)
    o, d = oracle.gen_primary(cam, 320, 180)
    host_rt.reset_ray_counter()
    for occ in (False, True):
        g, c = host_rt.raycast(o, d, occ), oracle.raycast(head["sd"], o, d, occ)
        assert np.array_equal(g["id"], c["id"])
        for k in "tuv":
            assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32))
        hit = c["id"] >= 0
        want = (o + c["t"][:, None] * d)[hit]
        assert np.array_equal(g["point"][hit].view(np.uint32), want.astype(np.float32).view(np.uint32))
    assert host_rt.ray_count() == 2 * 320 * 180


def test_single_ray_signature(host_rt, head, oracle):
    """RayTracer::raycast(orig, dir, occlusion) -- the reference's one-ray signature -- also runs on the device."""
    cam = scene.make_camera(width=16, height=9)
    o, d = oracle.gen_primary(cam, 16, 9)
    g, c = host_rt.raycast(o, d, False, single=True), oracle.raycast(head["sd"], o, d, False)
    assert np.array_equal(g["id"], c["id"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    miss = host_rt.raycast(np.float32([[5, 5, 5]]), np.float32([[1, 0, 0]]), False, single=True)
    assert miss["id"][0] == -1 and miss["t"][0] == np.float32(3.402823466e38) and miss["u"][0] == 0 and miss["v"][0] == 0


def test_hierarchy_cache_roundtrip(host_rt, head, oracle, tmp_path):
    """saveHierarchy / loadHierarchy (reference src/RayTracer.cpp:74-165): 41-byte unpadded node records (2 x Vec3f, 3 x int32, 1-byte bool, int32); a reloaded
    hierarchy traces identically."""
    path = str(tmp_path / ("Hierarchy-%s-sah-x64.bin" % host_rt.mesh_md5()))
    host_rt.save_hierarchy(path)
    assert os.path.getsize(path) == 4 + 4 * 17692 + 4 + 41 * 15619 + 4
    raw = open(path, "rb").read()
    assert np.frombuffer(raw, "<i4", 1)[0] == 17692
    assert np.array_equal(np.frombuffer(raw, "<i4", 17692, 4), head["list"])
    host_rt.load_hierarchy(path)
    nodes, lst = host_rt.bvh()
    for f in ("bmin", "bmax", "startIdx", "endIdx", "rightChild", "leaf", "axis", "triangles"):
        assert np.array_equal(nodes[f], head["nodes"][f]), f
    cam = scene.make_camera(width=200, height=100)
    o, d = oracle.gen_primary(cam, 200, 100)
    g, c = host_rt.raycast(o, d, False), oracle.raycast(head["sd"], o, d, False)
    assert np.array_equal(g["id"], c["id"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))


def test_host_path_trace_renderer_matches_oracle(host_rt, head, oracle):
    """FW::PathTraceRenderer (start -> checkFinish per pass -> updatePicture) over the device pass: two passes accumulate exactly
    what two oracle passes with pass seeds 0 and 1 accumulate; the display image is the oracle's resolve; rays are counted."""
    w, h = 80, 48
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True, normal_mapped=True)
    lights = [scene.make_light()]
    # the host importer attached the maps that exist next to head.ptmesh (floor Kd + normal, wall normal): same scene for the oracle
    tex, index = scene.head_reference_textures(head["mesh"])
    assert len(tex) == 3
    sd = dict(head["sd"])
    sd["materials"], sd["textures"] = scene.mesh_materials(head["mesh"], textured=True, texture_index=index), tex
    host_rt.reset_ray_counter()  # like the reference, a pass adds whatever the tracer counted since its last reset
    accum, resolved, rays, passes = host_rt.path_trace(prm, lights, passes=2)
    assert passes == 2
    cam = scene.make_camera(width=w, height=h)
    c, total = None, 0
    for seed in (0, 1):
        prm.pass_seed = seed
        c, ctr = oracle.render_pass(sd, prm, cam, lights, accum=c, nthreads=8)
        total += ctr["rays"]
    assert np.array_equal(accum.view(np.uint32), c.view(np.uint32))
    assert np.array_equal(resolved.view(np.uint32), oracle.resolve(c, libm=False).view(np.uint32))
    assert rays == total


def test_host_device_built_hierarchy(head, oracle, tmp_path):
    """RayTracer::constructHierarchyOnDevice (device-side linear BVH, not in the reference): the installed nodes / list trace exactly
    like the oracle on the same tree, survive the hierarchy cache file, and give the SAH tree's closest distances."""
    rt = hostapi.HostRayTracer(scene.HEAD_MESH, sah=True)
    try:
        rt.construct_hierarchy_on_device(3)
        nodes, lst = rt.bvh()
        assert rt.num_nodes == nodes.shape[0] and sorted(lst.tolist()) == list(range(17692))
        assert rt.depth_field == int((nodes["leaf"] == 0).sum())  # BVH::depth counts splits
        sd = dict(head["sd"])
        sd["nodes"], sd["list"] = nodes, lst
        cam = scene.make_camera(width=200, height=120)
        o, d = oracle.gen_primary(cam, 200, 120)
        for occ in (False, True):
            g, c = rt.raycast(o, d, occ), oracle.raycast(sd, o, d, occ)
            assert np.array_equal(g["id"], c["id"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
        ref = oracle.raycast(head["sd"], o, d, False)
        assert np.array_equal(rt.raycast(o, d, False)["t"].view(np.uint32), ref["t"].view(np.uint32))
        path = str(tmp_path / "Hierarchy-lbvh.bin")
        rt.save_hierarchy(path)
        rt.load_hierarchy(path)
        n2, l2 = rt.bvh()
        for f in ("bmin", "bmax", "startIdx", "endIdx", "rightChild", "leaf", "axis"):
            assert np.array_equal(n2[f], nodes[f]), f
        assert np.array_equal(l2, lst)
        # keepTree = false: the same tree built together with the device scene (ptgpu_scene_create_lbvh), nothing copied to the host
        want = {occ: rt.raycast(o, d, occ) for occ in (False, True)}
        rt.construct_hierarchy_on_device(3, keep_tree=False)
        assert rt.num_nodes == 0 and rt.list_size == 0
        for occ in (False, True):
            g = rt.raycast(o, d, occ)
            assert np.array_equal(g["id"], want[occ]["id"]) and np.array_equal(g["t"].view(np.uint32), want[occ]["t"].view(np.uint32))
        with pytest.raises(hostapi.HostError, match="without a host copy"):
            rt.save_hierarchy(path)
    finally:
        rt.close()


def _run_ptrender(tmp_path, *args):
    import subprocess
    exe = os.path.join(os.path.dirname(hostapi.LIB_PATH), "ptrender")
    assert os.path.exists(exe), "pathtracer_amd/ptrender is not built (run __graft_entry__.build())"
    r = subprocess.run([exe, scene.HEAD_MESH] + [str(a) for a in args], cwd=str(tmp_path), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_ptrender_cli_picture_matches_oracle(tmp_path, head, oracle):
    """ptrender = the headless App (reference src/App.cpp:146-239 process_args, :821-908 constructTracer, :457-489 path-trace mode):
    -builder / -spp / -bat_render as the reference spells them; its PNG must be the oracle's resolved picture, quantised to 8 bits,
    and -output_images must leave the reference's per-pass "pt-%03dppp.png" (src/PathTraceRenderer.cpp:900-902)."""
    w, h = 96, 64
    out = _run_ptrender(tmp_path, "-builder", "sah", "-spp", 2, "-size", "%dx%d" % (w, h), "-rr", "-bat_render", "-passes", 2, "-output_images",
                        "-o", "out.png")
    assert "Triangles: 17692  nodes: 15619" in out and "2 pass(es)" in out
    cam, lights = scene.make_camera(width=w, height=h), [scene.make_light()]
    acc = None
    pictures = []
    for seed in (0, 1):
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True, attenuation=1.0, filter_width=1.0,
                                focal_distance=0.5, aperture=0.0, pass_seed=seed)
        acc, _ = oracle.render_pass(head["sd"], prm, cam, lights, accum=acc, nthreads=8)
        pic = oracle.resolve(acc)
        pictures.append((np.clip(pic[..., :3], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))
    assert np.array_equal(hostapi.load_image(tmp_path / "out.png")[..., :3], pictures[1])
    assert np.array_equal(hostapi.load_image(tmp_path / "pt-001ppp.png")[..., :3], pictures[0])
    assert np.array_equal(hostapi.load_image(tmp_path / "pt-002ppp.png")[..., :3], pictures[1])


def test_ptrender_lbvh_builder_and_cache(tmp_path):
    """-builder lbvh (device-built tree) renders the same picture as the SAH tree up to grazing-ray ties, and without -bat_render the
    MD5-named hierarchy cache file is written and then loaded (reference src/App.cpp:841-892)."""
    w, h = 64, 48
    args = ["-spp", 4, "-size", "%dx%d" % (w, h), "-bounces", 1, "-cache", "."]
    out1 = _run_ptrender(tmp_path, "-builder", "sah", *args, "-o", "sah.png")
    assert "Saved hierarchy to" in out1
    out2 = _run_ptrender(tmp_path, "-builder", "sah", *args, "-o", "sah2.png")
    assert "Loaded hierarchy from" in out2
    a, b = hostapi.load_image(tmp_path / "sah.png"), hostapi.load_image(tmp_path / "sah2.png")
    assert np.array_equal(a, b) and a[..., :3].max() > 30
    _run_ptrender(tmp_path, "-builder", "lbvh", *args, "-o", "lbvh.png")
    c = hostapi.load_image(tmp_path / "lbvh.png")
    assert np.abs(a[..., :3].astype(np.int32) - c[..., :3].astype(np.int32)).mean() < 1.0


def test_static_per_sample_functions_match_oracle(host_rt, head, oracle):
    """PathTraceRenderer::tracePath / traceWhitted / getLightToSample / getTextureParameters / pathTraceBlock with the reference's static
    signatures (src/PathTraceRenderer.hpp:97-102), one device call each: values and random draws consumed equal the oracle's."""
    tex, index = scene.head_reference_textures(head["mesh"])
    sd = dict(head["sd"])
    sd["materials"], sd["textures"] = scene.mesh_materials(head["mesh"], textured=True, texture_index=index), tex
    w, h, n = 96, 64, 150
    rng = np.random.RandomState(31)
    xy = rng.uniform(-0.9, 0.9, (n, 2)).astype(np.float32)
    base = rng.randint(0, 1 << 20, n).astype(np.int32)
    seeds = rng.randint(0, 1 << 31, (n, 4)).astype(np.uint32)
    advance = rng.randint(0, 4, n).astype(np.uint32)
    state = np.stack([np.array([oracle.rng_key(*k) for k in seeds], np.uint32), advance], 1)
    cam = scene.make_camera(width=w, height=h)
    lights = [scene.make_light(), scene.make_light(position=(0.4, 0.8, 0.5), size=(0.1, 0.2), E=(30, 40, 50))]
    for whitted in (False, True):
        prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True, normal_mapped=True, whitted=whitted,
                                num_light_rays=3, pass_seed=5)
        got = host_rt.statics(prm, lights, xy, base, seeds, advance, block_idx=None if whitted else 4)
        c_rgb, c_ctr = oracle.trace_samples(sd, prm, cam, lights, xy, base, state)
        same = (got["rgb"].view(np.uint32) == c_rgb.view(np.uint32)) | (np.isnan(got["rgb"]) & np.isnan(c_rgb))
        assert same.all() and np.array_equal(got["ctr"], c_ctr) and c_rgb.max() > 0.01
        pos = np.concatenate([xy, np.full((n, 1), 0.3, np.float32)], 1)
        ci, cc = oracle.light_to_sample(lights, pos, state)
        assert np.array_equal(got["light"][:, 0], ci) and np.array_equal(got["light"][:, 1], cc) and len(np.unique(ci)) == 2
        if not whitted:
            c_img, _ = oracle.render_pass(sd, prm, cam, lights, nthreads=1, block_begin=4, block_end=5)
            assert np.array_equal(got["block"].view(np.uint32), c_img.view(np.uint32)) and c_img[..., 3].sum() > 0
    o, d = oracle.gen_primary(scene.make_camera(width=64, height=40), 64, 40)
    hits = oracle.raycast(sd, o, d, False)
    g9 = host_rt.texture_parameters_of_hits(o, d, bilinear=True, normal_mapped=True)
    m = hits["id"] >= 0
    cd, cn, cs = oracle.texture_parameters(sd, scene.make_params(bilinear=True, normal_mapped=True), hits["id"][m], hits["u"][m], hits["v"][m])
    want = np.concatenate([cd, cn, cs], 1)
    same = (g9[m].view(np.uint32) == want.view(np.uint32)) | (np.isnan(g9[m]) & np.isnan(want))
    assert same.all() and m.sum() > 500 and not g9[~m].any()
