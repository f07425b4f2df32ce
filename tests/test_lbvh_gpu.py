"""Device-side linear BVH builder (ptgpu_bvh_build_lbvh): the tree must be a valid reference-layout tree
(ptgpu_scene_create's checks, every triangle in exactly one leaf, boxes that contain what is below them), the GPU kernels and the CPU
oracle must agree bit for bit when both traverse it, and on the closed head mesh closest-hit distances must equal those found through
the reference's SAH tree (the closest hit does not depend on the tree; only ties between triangles at exactly equal t may name another
triangle -- rays grazing zero-thickness boxes, which the unpadded slab test decides per tree, do not occur there)."""
import numpy as np
import pytest

from pathtracer_amd import capi, scene

pytestmark = pytest.mark.gpu


def _check_tree(nodes, lst, pos, max_leaf):
    n = pos.shape[0]
    assert sorted(lst.tolist()) == list(range(n)), "list is not a permutation of the triangles"
    lo, hi = pos.min(axis=1), pos.max(axis=1)
    # walk the pre-order layout: left child = index + 1
    stack, seen, next_list, depth = [(0, 1)], 0, 0, 0
    while stack:
        i, d = stack.pop()
        seen += 1
        depth = max(depth, d)
        nd = nodes[i]
        s, e = int(nd["startIdx"]), int(nd["endIdx"])
        tri = lst[s:e + 1]
        assert np.all(nd["bmin"] <= lo[tri].min(axis=0)) and np.all(nd["bmax"] >= hi[tri].max(axis=0)), "node %d: box does not contain its triangles" % i
        assert int(nd["triangles"]) == e - s + 1
        if nd["leaf"]:
            assert s == next_list and e - s + 1 <= max_leaf
            next_list = e + 1
        else:
            assert e - s + 1 > max_leaf and int(nd["leftChild"]) == i + 1
            l, r = nodes[i + 1], nodes[int(nd["rightChild"])]
            assert int(l["startIdx"]) == s and int(l["endIdx"]) + 1 == int(r["startIdx"]) and int(r["endIdx"]) == e
            stack.append((int(nd["rightChild"]), d + 1))
            stack.append((i + 1, d + 1))
    assert seen == nodes.shape[0] and next_list == n
    return depth


def _scene_with(sd, nodes, lst):
    out = dict(sd)
    out["nodes"], out["list"] = nodes, lst
    return out


def _rays(n, seed, lo, hi, length):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= (length / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("max_leaf", [1, 3, 8])
def test_head_lbvh_valid_and_traversal_exact(gpu_ctx, head, oracle, max_leaf):
    pos = head["mesh"]["pos"]
    nodes, lst, depth = gpu_ctx.bvh_build_lbvh(pos, max_leaf)
    assert _check_tree(nodes, lst, pos, max_leaf) == depth <= 64
    sd = _scene_with(head["sd"], nodes, lst)
    s = capi.Scene(gpu_ctx, sd)
    try:
        cam = scene.make_camera(width=160, height=120)
        o, d = oracle.gen_primary(cam, 160, 120)
        ro, rd = _rays(20000, 5, -0.4, 0.4, 3.0)
        o, d = np.concatenate([o, ro]), np.concatenate([d, rd])
        for occlusion in (False, True):
            g, c = s.raycast(o, d, occlusion), oracle.raycast(sd, o, d, occlusion)
            assert np.array_equal(g["id"], c["id"])
            for k in ("t", "u", "v"):
                assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32)), k
        # same closest distances as through the reference's SAH tree
        g = s.raycast(o, d, False)
        ref = oracle.raycast(head["sd"], o, d, False)
        assert np.array_equal((g["id"] >= 0), (ref["id"] >= 0))
        assert np.array_equal(g["t"].view(np.uint32), ref["t"].view(np.uint32))
        assert (g["id"] != ref["id"]).mean() < 1e-3  # only exact ties may differ
    finally:
        s.close()


def test_soup_render_on_lbvh_exact(gpu_ctx, soup, oracle):
    """The whole path tracer on the device-built tree: accumulation image bit-exact against the oracle on the same tree."""
    pos = soup["mesh"]["pos"]
    nodes, lst, depth = gpu_ctx.bvh_build_lbvh(pos, 3)
    _check_tree(nodes, lst, pos, 3)
    sd = _scene_with(soup["sd"], nodes, lst)
    s = capi.Scene(gpu_ctx, sd)
    try:
        w, h = 64, 48
        cam = scene.make_camera(eye=(0.2, -0.3, 2.2), forward=(-0.05, 0.1, -1.0), up=(0, 1, 0), fov_deg=60.0, near=0.01, far=8.0, width=w, height=h)
        lights = [scene.make_light(position=(0.3, 0.2, 1.5), col0=(1, 0, 0), col1=(0, 1, 0), col2=(0, 0, 1), size=(0.3, 0.2), E=(60, 50, 40))]
        prm = scene.make_params(width=w, height=h, bounces=-2, num_aa_rays=2, sobol_block=True, pass_seed=4)
        g = s.render_pass(prm, cam, lights)
        c, _ = oracle.render_pass(sd, prm, cam, lights, nthreads=8)
        assert c[..., :3].max() > 0.01
        assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    finally:
        s.close()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 64, 1000])
def test_small_and_degenerate_inputs(gpu_ctx, oracle, n):
    rng = np.random.RandomState(n)
    pos = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    if n >= 64:
        pos[: n // 2] = pos[0]  # many coincident triangles: equal Morton keys are split by position
    nodes, lst, depth = gpu_ctx.bvh_build_lbvh(pos, 3)
    assert _check_tree(nodes, lst, pos, 3) == depth
    sd = dict(pos=pos, nrm=scene.flat_normals(pos), uv=np.zeros((n, 3, 2), np.float32), mat_id=np.zeros(n, np.int32),
              woop=oracle.woop_from_positions(pos), materials=[scene.make_material()], textures=[], nodes=nodes, list=lst)
    s = capi.Scene(gpu_ctx, sd)
    try:
        o, d = _rays(4000, 9, -1.5, 1.5, 4.0)
        g, c = s.raycast(o, d, False), oracle.raycast(sd, o, d, False)
        assert np.array_equal(g["id"], c["id"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    finally:
        s.close()


def test_rejects_bad_input(gpu_ctx):
    nodes, lst, depth = gpu_ctx.bvh_build_lbvh(np.zeros((0, 3, 3), np.float32), 3)
    assert nodes.shape[0] == 0 and depth == 0
    pos = np.zeros((4, 3, 3), np.float32)
    pos[2, 1, 0] = np.nan
    with pytest.raises(capi.PtgpuError):
        gpu_ctx.bvh_build_lbvh(pos, 3)
    with pytest.raises(capi.PtgpuError):
        gpu_ctx.bvh_build_lbvh(np.zeros((4, 3, 3), np.float32), 0)


@pytest.mark.parametrize("max_leaf", [1, 3])
def test_scene_straight_from_the_device_built_tree(gpu_ctx, head, oracle, max_leaf):
    """ptgpu_scene_create_lbvh (tree and traversal records never leave the device) against ptgpu_bvh_build_lbvh + ptgpu_scene_create
    (the same tree through host arrays): same node count, depth and record bytes; hit records of closest-hit and any-hit rays and a
    path-trace pass bit-identical -- and, through the host-array tree, equal to the oracle's."""
    pos = head["mesh"]["pos"]
    nodes, lst, depth = gpu_ctx.bvh_build_lbvh(pos, max_leaf)
    sd = _scene_with(head["sd"], nodes, lst)
    a = capi.Scene(gpu_ctx, sd)
    b = capi.Scene(gpu_ctx, head["sd"], lbvh_max_tris_in_leaf=max_leaf)
    try:
        assert b.num_nodes_built == nodes.shape[0] and a.stats() == b.stats() and b.stats()["depth"] == depth
        cam = scene.make_camera(width=160, height=96)
        o, d = oracle.gen_primary(cam, 160, 96)
        ga, gb = a.raycast(o, d, False), b.raycast(o, d, False)
        c = oracle.raycast(sd, o, d, False, nthreads=8)
        for k in ("id", "t", "u", "v"):
            assert np.array_equal(ga[k].view(np.uint32), gb[k].view(np.uint32)) and np.array_equal(gb[k].view(np.uint32), c[k].view(np.uint32)), k
        ro, rd = _rays(30000, 4, -0.6, 0.6, 1.5)
        assert np.array_equal(a.raycast(ro, rd, True)["id"], b.raycast(ro, rd, True)["id"])
        prm = scene.make_params(width=96, height=64, bounces=-1, num_aa_rays=2, sobol_block=True, pass_seed=8)
        cam2, lights = scene.make_camera(width=96, height=64), [scene.make_light()]
        assert np.array_equal(a.render_pass(prm, cam2, lights).view(np.uint32), b.render_pass(prm, cam2, lights).view(np.uint32))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("n", [0, 1, 2, 5, 1000])
def test_scene_from_device_tree_small_inputs(gpu_ctx, oracle, n):
    rng = np.random.RandomState(n + 3)
    pos = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    nodes, lst, _ = gpu_ctx.bvh_build_lbvh(pos, 3)
    sd = dict(pos=pos, nrm=scene.flat_normals(pos) if n else np.zeros((0, 3, 3), np.float32), uv=np.zeros((n, 3, 2), np.float32),
              mat_id=np.zeros(n, np.int32), woop=oracle.woop_from_positions(pos) if n else np.zeros((0, 12), np.float32),
              materials=[scene.make_material()], textures=[], nodes=nodes, list=lst)
    a, b = capi.Scene(gpu_ctx, sd), capi.Scene(gpu_ctx, sd, lbvh_max_tris_in_leaf=3)
    try:
        o, d = _rays(3000, 2, -1.5, 1.5, 4.0)
        ga, gb = a.raycast(o, d, False), b.raycast(o, d, False)
        assert np.array_equal(ga["id"], gb["id"]) and np.array_equal(ga["t"].view(np.uint32), gb["t"].view(np.uint32))
        assert (ga["id"] >= 0).any() or n < 2
    finally:
        a.close()
        b.close()
    with pytest.raises(capi.PtgpuError, match="builds the hierarchy itself"):
        desc, keep = capi.make_scene_desc(sd)
        if n == 0:
            desc.num_nodes = 1  # an empty description has nothing to object to: force the check
        h = capi._P()
        capi._check(gpu_ctx.lib.ptgpu_scene_create_lbvh(gpu_ctx.handle, capi.C.byref(desc), 3, capi.C.byref(h), None))
