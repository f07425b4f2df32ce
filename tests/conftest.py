import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


def _scene_dict(mesh, woop, nodes, lst, materials=None, textures=()):
    from pathtracer_amd import scene
    return dict(pos=mesh["pos"], nrm=mesh["nrm"], uv=mesh["uv"], mat_id=mesh["mat_id"], woop=woop,
                materials=materials if materials is not None else scene.mesh_materials(mesh), textures=list(textures),
                nodes=nodes, list=lst)


@pytest.fixture(scope="session")
def head(oracle):
    """head.obj scene (untextured materials) with the oracle-built SAH BVH and Woop data."""
    from pathtracer_amd import scene
    mesh = scene.load_ptmesh()
    nodes, lst, stats = oracle.bvh_build(mesh["pos"], True)
    woop = oracle.woop_from_positions(mesh["pos"])
    sd = _scene_dict(mesh, woop, nodes, lst)
    return dict(mesh=mesh, nodes=nodes, list=lst, stats=stats, woop=woop, sd=sd)


@pytest.fixture(scope="session")
def soup(oracle):
    """Small random triangle soup (256 triangles) with a median-split BVH."""
    from pathtracer_amd import scene
    pos = scene.grid_scene(8, 8, seed=3)
    rng = np.random.RandomState(7)
    mesh = dict(pos=pos, nrm=scene.flat_normals(pos), uv=rng.rand(pos.shape[0], 3, 2).astype(np.float32),
                mat_id=np.zeros(pos.shape[0], np.int32))
    nodes, lst, stats = oracle.bvh_build(pos, False)
    woop = oracle.woop_from_positions(pos)
    sd = _scene_dict(mesh, woop, nodes, lst, materials=[scene.make_material(specular=(0.15, 0.15, 0.15), glossiness=50.0)])
    return dict(mesh=mesh, nodes=nodes, list=lst, stats=stats, woop=woop, sd=sd)


@pytest.fixture(scope="session")
def gpu_ctx():
    from pathtracer_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
