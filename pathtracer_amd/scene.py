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
    e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    n = np.cross(e1, e2)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    return np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)
