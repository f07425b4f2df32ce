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
    prm = scene.make_params(width=w, height=h, bounces=-1, num_aa_rays=2, sobol_block=True)
    lights = [scene.make_light()]
    host_rt.reset_ray_counter()  # like the reference, a pass adds whatever the tracer counted since its last reset
    accum, resolved, rays, passes = host_rt.path_trace(prm, lights, passes=2)
    assert passes == 2
    cam = scene.make_camera(width=w, height=h)
    c, total = None, 0
    for seed in (0, 1):
        prm.pass_seed = seed
        c, ctr = oracle.render_pass(head["sd"], prm, cam, lights, accum=c, nthreads=8)
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
    finally:
        rt.close()
