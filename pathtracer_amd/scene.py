"""Scene plumbing for tests and bench.py: .ptmesh loading, camera / light PODs, procedural stand-ins.

Nothing here computes on the hot path; it prepares the byte-identical inputs that both the HIP library and
(in tests) the oracle consume.  The camera follows the conventions of the reference's external FW framework as
far as its call sites show them; FW itself is not in the reference's tree, so the
matrices are this build's definition ("parity unpinned").
"""
import os
import struct

import numpy as np

from .capi import Camera, Light, Material, Params

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
HEAD_MESH = os.path.join(ASSETS, "head.ptmesh")


def load_ptmesh(path=HEAD_MESH):
    """Read a .ptmesh (tools/convert_head_obj.py documents the layout). Returns per-triangle arrays like
    App::constructTracer builds them (reference src/App.cpp:826-838): pos/nrm [n,3,3], uv [n,3,2], mat_id [n]."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:8] != b"PTMESH1\0":
        raise ValueError("%s: not a .ptmesh file" % path)
    nv, nt, nm, _ = struct.unpack_from("<4i", blob, 8)
    off = 24
    def take(count, dtype):
        nonlocal off
        a = np.frombuffer(blob, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a
    vpos = take(nv * 3, "<f4").reshape(nv, 3)
    vnrm = take(nv * 3, "<f4").reshape(nv, 3)
    vuv = take(nv * 2, "<f4").reshape(nv, 2)
    idx = take(nt * 3, "<i4").reshape(nt, 3)
    tmat = take(nt, "<i4")
    mats = []
    for _i in range(nm):
        name, d0, d1, d2, d3, s0, s1, s2, gl, kd, bump = struct.unpack_from("<32s4f3ff64s64s", blob, off)
        off += struct.calcsize("<32s4f3ff64s64s")
        mats.append(dict(name=name.split(b"\0")[0].decode(), diffuse=(d0, d1, d2, d3), specular=(s0, s1, s2), glossiness=gl,
                         map_kd=kd.split(b"\0")[0].decode(), map_bump=bump.split(b"\0")[0].decode()))
    return dict(pos=np.ascontiguousarray(vpos[idx]), nrm=np.ascontiguousarray(vnrm[idx]), uv=np.ascontiguousarray(vuv[idx]),
                mat_id=np.ascontiguousarray(tmat), indices=idx, materials=mats)


def make_material(diffuse=(0.75, 0.75, 0.75, 1.0), specular=(0.0, 0.0, 0.0), glossiness=0.0, tex_diffuse=-1, tex_alpha=-1,
                  tex_normal=-1, tex_specular=-1):
    m = Material()
    m.diffuse[:] = diffuse
    m.specular[:] = specular
    m.glossiness = glossiness
    m.tex_diffuse, m.tex_alpha, m.tex_normal, m.tex_specular = tex_diffuse, tex_alpha, tex_normal, tex_specular
    return m


def mesh_materials(mesh, textured=False, texture_index=None):
    """Material PODs for a loaded mesh. textured=False drops every map (the SURVEY.md 6 probe setup)."""
    out = []
    for m in mesh["materials"]:
        td = tn = -1
        if textured and texture_index is not None:
            td = texture_index.get(m["map_kd"], -1)
            tn = texture_index.get(m["map_bump"], -1)
        out.append(make_material(m["diffuse"], m["specular"], m["glossiness"], tex_diffuse=td, tex_normal=tn))
    return out


# ---- camera / light PODs come from the C++ host classes (FW::CameraControls, FW::AreaLight) ----
def make_camera(**kw):
    """Camera POD; defaults = the probe camera of SURVEY.md 8d config C1 (see hostapi.make_camera)."""
    from . import hostapi
    return hostapi.make_camera(**kw)


def make_light(**kw):
    """AreaLight POD; default = the SURVEY.md C1 light (see hostapi.make_light)."""
    from . import hostapi
    return hostapi.make_light(**kw)


def make_params(width=512, height=512, bounces=-1, num_aa_rays=16, sobol_block=True, bilinear=True, normal_mapped=False,
                depth_of_field=False, whitted=False, num_light_rays=32, use_alpha=True, pass_seed=0, aperture=0.0001,
                attenuation=1.0, filter_width=1.0, focal_distance=0.1, intensity=1.0, termination_prob=0.2):
    """RenderParams POD; defaults = App's (reference src/App.cpp:33-53) with the C1 settings of SURVEY.md 8d."""
    p = Params()
    p.width, p.height, p.bounces, p.num_aa_rays = width, height, bounces, num_aa_rays
    p.sobol_block, p.bilinear, p.normal_mapped = int(sobol_block), int(bilinear), int(normal_mapped)
    p.depth_of_field, p.whitted, p.num_light_rays, p.use_alpha = int(depth_of_field), int(whitted), num_light_rays, int(use_alpha)
    p.pass_seed = pass_seed
    p.aperture, p.attenuation, p.filter_width = aperture, attenuation, filter_width
    p.focal_distance, p.intensity, p.termination_prob = focal_distance, intensity, termination_prob
    return p


# ---- procedural stand-ins (deterministic) ----
def value_noise_texture(seed, size=256, octaves=4, normal_map=False):
    """Seeded value-noise RGBA8 texture; stands in for the head maps missing from the reference checkout."""
    rng = np.random.RandomState(seed)
    acc = np.zeros((size, size), np.float64)
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        n = 4 << o
        g = rng.rand(n + 1, n + 1)
        g[-1, :], g[:, -1] = g[0, :], g[:, 0]  # tileable
        xs = np.linspace(0, n, size, endpoint=False)
        i = xs.astype(int)
        fr = xs - i
        fr = fr * fr * (3 - 2 * fr)
        a = g[i][:, i] * (1 - fr)[None, :] + g[i][:, i + 1] * fr[None, :]
        b = g[i + 1][:, i] * (1 - fr)[None, :] + g[i + 1][:, i + 1] * fr[None, :]
        acc += amp * (a * (1 - fr)[:, None] + b * fr[:, None])
        tot += amp
        amp *= 0.5
    h = acc / tot
    out = np.empty((size, size, 4), np.uint8)
    if normal_map:
        dx = np.roll(h, -1, 1) - np.roll(h, 1, 1)
        dy = np.roll(h, -1, 0) - np.roll(h, 1, 0)
        n = np.stack([-dx * size * 0.05, -dy * size * 0.05, np.ones_like(h)], -1)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        out[..., :3] = np.clip((n * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    else:
        base = np.array([0.80, 0.62, 0.52]) if seed % 2 else np.array([0.62, 0.60, 0.58])
        out[..., :3] = np.clip((0.55 + 0.45 * h)[..., None] * base * 255.0 + 0.5, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


HEAD_TEXTURE_DIR = ASSETS  # next to head.ptmesh, where the host importer looks for the maps a material names


def head_textures(mesh, size=1600):
    """The six maps head.mtl names, as BASELINE configs[2] / SURVEY.md 8d C3 specify them: the three files the reference checkout
    holds (assets/: concrete1.png = floor map_Kd, concrete1_normal.png = floor map_bump, concrete2_normal.png = wall map_bump,
    copied by tools/import_head_textures.py, decoded by the host library's own PNG reader) and seeded value-noise stand-ins of the same
    size for the three missing ones (lambertian.png, normal.png, concrete2.png).  Returns (textures, name -> index, labels)."""
    from . import hostapi
    textures, index, labels = [], {}, {}
    for m in mesh["materials"]:
        for name, is_normal in ((m["map_kd"], False), (m["map_bump"], True)):
            if name and name not in index:
                index[name] = len(textures)
                path = os.path.join(HEAD_TEXTURE_DIR, name)
                if os.path.exists(path):
                    textures.append(hostapi.load_image(path))
                    labels[name] = "reference file"
                else:
                    textures.append(value_noise_texture(17 + 2 * len(textures) + (0 if is_normal else 1), size, normal_map=is_normal))
                    labels[name] = "procedural stand-in"
    return textures, index, labels


def head_reference_textures(mesh):
    """Only the maps that exist as files next to head.ptmesh -- what the host importer attaches when it loads the mesh
    (host/MeshIO.cpp attachTextures): floor map_Kd + map_bump, wall map_bump.  Returns (textures, name -> index)."""
    from . import hostapi
    textures, index = [], {}
    for m in mesh["materials"]:
        for name in (m["map_kd"], m["map_bump"]):
            if name and name not in index and os.path.exists(os.path.join(HEAD_TEXTURE_DIR, name)):
                index[name] = len(textures)
                textures.append(hostapi.load_image(os.path.join(HEAD_TEXTURE_DIR, name)))
    return textures, index


def head_standin_textures(mesh, size=1600):
    """Procedural stand-ins (seeded value noise, size x size RGBA8) for the six maps head.mtl names.  Three of them are
    missing from the reference checkout (lambertian.png, normal.png, concrete2.png) and none may be read on the
    GPU box, so every map is generated: same count, size and roles (map_Kd -> diffuse, map_bump -> tangent-space normal).
    Returns (textures, name -> index)."""
    textures, index = [], {}
    for m in mesh["materials"]:
        for name, is_normal in ((m["map_kd"], False), (m["map_bump"], True)):
            if name and name not in index:
                index[name] = len(textures)
                textures.append(value_noise_texture(17 + 2 * len(textures) + (0 if is_normal else 1), size, normal_map=is_normal))
    return textures, index


def grid_scene(nx=8, ny=8, seed=1, jitter=0.3):
    """Small random triangle soup (tetra-ish fans on a jittered grid) for fast parity tests. Returns pos [n,3,3]."""
    rng = np.random.RandomState(seed)
    tris = []
    for i in range(nx):
        for j in range(ny):
            c = np.array([i - nx / 2.0, j - ny / 2.0, 0.0]) * 0.25 + rng.uniform(-jitter, jitter, 3) * 0.25
            p = c + rng.uniform(-0.15, 0.15, (4, 3))
            for a, b, cc in ((0, 1, 2), (0, 2, 3), (0, 3, 1), (1, 3, 2)):
                tris.append([p[a], p[b], p[cc]])
    return np.asarray(tris, np.float32)


def flat_normals(pos):
    pos = np.asarray(pos, np.float64)  # float64: squared edge lengths of very large triangles overflow float32
    e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    n = np.cross(e1, e2)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-300)
    return np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)


# ---- procedural stand-in scenes for BASELINE configs[3] and [4] (Sponza is not in the reference checkout) ----
def _grid_patch(p00, du, dv, nu, nv, disp=None):
    """Tessellated parallelogram p00 + s*du + t*dv, (nu x nv) quads -> triangles [2*nu*nv,3,3], uv [.,3,2]."""
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    P = p00[None, None, :] + s[..., None] * du[None, None, :] + t[..., None] * dv[None, None, :]
    if disp is not None:
        P = P + disp(s, t)
    UV = np.stack([s * 4.0, t * 4.0], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    ua, ub, uc, ud = UV[:-1, :-1], UV[1:, :-1], UV[1:, 1:], UV[:-1, 1:]
    tri = np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)])
    uv = np.concatenate([np.stack([ua, ub, uc], -2).reshape(-1, 3, 2), np.stack([ua, uc, ud], -2).reshape(-1, 3, 2)])
    return tri, uv


def _smooth_normals_flat(pos):
    return flat_normals(pos.astype(np.float32))


def atrium_scene(target_tris=262144, seed=7):
    """Procedural "atrium" standing in for Crytek Sponza (BASELINE configs[3]): a long hall with a displaced
    stone floor, side walls, two rows of fluted columns, arches between them and hanging sinusoidal curtains.  Deterministic.
    Returns dict(pos, nrm, uv, mat_id, materials) with ~target_tris triangles and 4 materials."""
    rng = np.random.RandomState(seed)
    k = max(1, int(round(np.sqrt(target_tris / 262144.0) * 16)))  # tessellation unit
    parts, uvs, mats = [], [], []

    def add(tri, uv, m):
        parts.append(tri); uvs.append(uv); mats.append(np.full(tri.shape[0], m, np.int32))

    bump = lambda amp: (lambda s, t: np.stack([np.zeros_like(s), amp * np.sin(37 * s) * np.cos(29 * t), np.zeros_like(s)], -1))  # noqa: E731
    L, Wd, Hh = 12.0, 5.0, 6.0
    add(*_grid_patch(np.array([-L / 2, 0, -Wd / 2]), np.array([L, 0, 0.0]), np.array([0, 0, Wd]), 12 * k, 5 * k, bump(0.01)), 0)   # floor
    wall = lambda s, t: np.stack([np.zeros_like(s), np.zeros_like(s), 0.02 * np.sin(23 * s) * np.sin(31 * t)], -1)  # noqa: E731
    add(*_grid_patch(np.array([-L / 2, 0, -Wd / 2]), np.array([L, 0, 0.0]), np.array([0, Hh, 0.0]), 10 * k, 5 * k, wall), 1)
    add(*_grid_patch(np.array([L / 2, 0, Wd / 2]), np.array([-L, 0, 0.0]), np.array([0, Hh, 0.0]), 10 * k, 5 * k, wall), 1)
    add(*_grid_patch(np.array([-L / 2, 0, Wd / 2]), np.array([0, 0, -Wd]), np.array([0, Hh, 0.0]), 4 * k, 5 * k, wall), 1)         # back wall
    ncol = 6
    for side in (-1.0, 1.0):
        for ci in range(ncol):
            cx, cz, r = -L / 2 + (ci + 0.5) * L / ncol, side * Wd * 0.28, 0.22
            nu, nv = 3 * k, 4 * k
            th, hh = np.meshgrid(np.linspace(0, 2 * np.pi, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
            rr = r * (1.0 + 0.06 * np.cos(12 * th)) * (1.0 + 0.15 * (hh - 0.5) ** 2)
            P = np.stack([cx + rr * np.cos(th), hh * 3.6, cz + rr * np.sin(th)], -1)
            UV = np.stack([th / (2 * np.pi) * 2, hh * 4], -1)
            a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
            ua, ub, uc, ud = UV[:-1, :-1], UV[1:, :-1], UV[1:, 1:], UV[:-1, 1:]
            add(np.concatenate([np.stack([a, c, b], -2).reshape(-1, 3, 3), np.stack([a, d, c], -2).reshape(-1, 3, 3)]),
                np.concatenate([np.stack([ua, uc, ub], -2).reshape(-1, 3, 2), np.stack([ua, ud, uc], -2).reshape(-1, 3, 2)]), 2)
            if ci + 1 < ncol:  # arch to the next column: half torus
                nu2, nv2 = 3 * k, 2 * k
                ph, ps = np.meshgrid(np.linspace(0, np.pi, nu2 + 1), np.linspace(0, 2 * np.pi, nv2 + 1), indexing="ij")
                R0, r0 = L / ncol / 2, 0.12
                ax = cx + R0 - (R0 + r0 * np.cos(ps)) * np.cos(ph)
                ay = 3.6 + (R0 + r0 * np.cos(ps)) * np.sin(ph) * 0.8
                az = cz + r0 * np.sin(ps)
                P = np.stack([ax, ay, az], -1)
                UV = np.stack([ph / np.pi * 3, ps / (2 * np.pi)], -1)
                a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
                ua, ub, uc, ud = UV[:-1, :-1], UV[1:, :-1], UV[1:, 1:], UV[:-1, 1:]
                add(np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 3), np.stack([a, c, d], -2).reshape(-1, 3, 3)]),
                    np.concatenate([np.stack([ua, ub, uc], -2).reshape(-1, 3, 2), np.stack([ua, uc, ud], -2).reshape(-1, 3, 2)]), 2)
    for ci in range(5):  # curtains
        cx = -L / 2 + (ci + 1) * L / 6 + rng.uniform(-0.2, 0.2)
        ph0 = rng.uniform(0, 6.28)
        fold = lambda s, t, ph0=ph0: np.stack([np.zeros_like(s), np.zeros_like(s), 0.12 * np.sin(18 * s + ph0) * (0.3 + t)], -1)  # noqa: E731
        add(*_grid_patch(np.array([cx - 0.7, 4.8, 0.0]), np.array([1.4, 0, 0.0]), np.array([0, -2.6, 0.0]), 5 * k, 5 * k, fold), 3)
    pos = np.concatenate(parts).astype(np.float32)
    uv = np.concatenate(uvs).astype(np.float32)
    mat_id = np.concatenate(mats)
    materials = [make_material(diffuse=(0.55, 0.52, 0.48, 1), specular=(0.05, 0.05, 0.05), glossiness=10.0),
                 make_material(diffuse=(0.70, 0.66, 0.58, 1), specular=(0.02, 0.02, 0.02), glossiness=5.0),
                 make_material(diffuse=(0.75, 0.73, 0.70, 1), specular=(0.15, 0.15, 0.15), glossiness=50.0),
                 make_material(diffuse=(0.65, 0.12, 0.10, 1), specular=(0.05, 0.05, 0.05), glossiness=20.0)]
    return dict(pos=pos, nrm=_smooth_normals_flat(pos), uv=uv, mat_id=mat_id, materials=materials)


def atrium_view(width, height):
    """Camera + two area lights for atrium_scene (looking down the hall)."""
    cam = make_camera(eye=(-5.2, 1.7, 0.1), forward=(1.0, 0.08, -0.02), up=(0, 1, 0), fov_deg=65.0, near=0.01, far=30.0, width=width, height=height)
    lights = [make_light(position=(0.0, 5.6, 0.0), col0=(1, 0, 0), col1=(0, 0, -1), col2=(0, 1, 0), size=(2.5, 0.8), E=(30, 28, 25)),
              make_light(position=(3.5, 4.5, 0.0), col0=(1, 0, 0), col1=(0, 0, -1), col2=(0, 1, 0), size=(0.5, 0.5), E=(60, 50, 40), intensity=1.5)]
    return cam, lights


def tetra_cloud_scene(num_tris=10_000_000, seed=11):
    """BVH-bandwidth stress scene (BASELINE configs[4]): a jittered grid of small tetrahedra in the unit box
    [-0.5,0.5]^3 plus a floor.  num_tris is rounded down to a multiple of 4.  Returns dict(pos, nrm, uv, mat_id, materials)."""
    n_tet = num_tris // 4
    g = int(np.ceil(n_tet ** (1.0 / 3.0)))
    rng = np.random.RandomState(seed)
    idx = np.arange(n_tet)
    cell = np.stack([idx % g, (idx // g) % g, idx // (g * g)], 1).astype(np.float32)
    c = (cell + 0.5 + rng.uniform(-0.35, 0.35, (n_tet, 3)).astype(np.float32)) / g - 0.5
    r = 0.45 / g
    v = c[:, None, :] + rng.uniform(-1, 1, (n_tet, 4, 3)).astype(np.float32) * r
    faces = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]])
    pos = v[:, faces, :].reshape(-1, 3, 3).astype(np.float32)
    uv = np.tile(np.array([[0, 0], [1, 0], [0, 1]], np.float32), (pos.shape[0], 1, 1))
    mat_id = (np.arange(pos.shape[0]) // 4 % 3).astype(np.int32)
    materials = [make_material(diffuse=(0.8, 0.3, 0.25, 1), specular=(0.1, 0.1, 0.1), glossiness=30.0),
                 make_material(diffuse=(0.3, 0.75, 0.35, 1), specular=(0.1, 0.1, 0.1), glossiness=30.0),
                 make_material(diffuse=(0.3, 0.4, 0.85, 1), specular=(0.1, 0.1, 0.1), glossiness=30.0)]
    return dict(pos=pos, nrm=flat_normals(pos), uv=uv, mat_id=mat_id, materials=materials)


def tetra_cloud_view(width, height):
    cam = make_camera(eye=(0.0, 0.15, 1.25), forward=(0.0, -0.1, -1.0), up=(0, 1, 0), fov_deg=50.0, near=0.01, far=4.0, width=width, height=height)
    lights = [make_light(position=(0.0, 0.95, 0.3), col0=(1, 0, 0), col1=(0, 0, -1), col2=(0, 1, 0), size=(0.4, 0.4), E=(40, 40, 40))]
    return cam, lights
