"""Device results against the REFERENCE'S OWN CODE, not the restatement: oracle/_ref/libptref.so is the reference's src/sobol.cpp,
src/sobol.h and src/rtIntersect.inl compiled where they lie (oracle/Makefile; the library travels to the GPU box, the sources do not).

  * ptgpu_sobol_sample_batch -- the device functions every kernel draws its Sobol points with (byte-indexed pair table + bit loop) --
    equals sobol::sample (reference src/sobol.h:39-49) bit for bit over all 1024 dimensions, on small, 24-bit-crossing, 32-bit and
    52-bit indices;
  * ptgpu_intersect_triangle1_batch equals intersect_triangle1 (reference src/rtIntersect.inl:215-259) on 100,000 random (ray,
    triangle) pairs, hits, misses and near-degenerate cases alike (the committed goldens are 300 of these).
These are the two pieces of the path that compile from the reference's tree without the absent FW framework."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref/libptref.so is not in this tree (it is built where /root/reference exists)")
    return r


def test_device_sobol_equals_the_reference_build(gpu_ctx, ref):
    rng = np.random.RandomState(17)
    n = 120_000
    idx = np.concatenate([
        np.arange(0, 20_000, dtype=np.uint64),                                  # the first points of the sequence
        rng.randint(0, 1 << 24, 30_000).astype(np.uint64),                      # pair table only
        rng.randint(1 << 24, 1 << 31, 30_000).astype(np.uint64),                # pair table + bit loop (what a deep render forms)
        rng.randint(0, 1 << 32, 20_000, dtype=np.int64).astype(np.uint64),      # any 32-bit index
        (rng.randint(0, 1 << 52, 19_990, dtype=np.int64)).astype(np.uint64),    # the reference's 52-bit range
        np.array([0, 1, 255, 256, 65535, 65536, (1 << 24) - 1, 1 << 24, (1 << 32) - 1, 1 << 32], np.uint64)])
    assert idx.shape[0] == n
    dim = rng.randint(0, 1024, n).astype(np.uint32)
    dim[:2048] = np.arange(2048) % 1024  # every dimension at least twice
    got = gpu_ctx.sobol_sample(idx, dim)
    want = np.array([ref.ptref_sobol_sample(int(i), int(d), 0) for i, d in zip(idx, dim)], np.float32)
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, "%d of %d Sobol values differ from the reference build, first at index %d dimension %d" % (
        bad.size, n, int(idx[bad[0]]), int(dim[bad[0]]))
    assert want.max() <= 1.0 and 0.49 < float(want.mean()) < 0.51
    # known answers from the reference's code (dimension 0 is the bit reversal of the index)
    assert gpu_ctx.sobol_sample(np.array([1, 2, 3, 8, 123456], np.uint64), np.zeros(5, np.uint32)).tolist() == [0.5, 0.25, 0.75, 0.0625, np.float32(0.00890350342)]
    with pytest.raises(Exception, match="1024"):
        gpu_ctx.sobol_sample(np.array([1], np.uint64), np.array([1024], np.uint32))


def test_device_intersect_triangle1_equals_the_reference_build(gpu_ctx, ref):
    rng = np.random.RandomState(23)
    n = 100_000
    v0, v1, v2 = (rng.uniform(-1, 1, (n, 3)).astype(np.float32) for _ in range(3))
    o = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    # half of the rays aim at a point of their triangle (hits from both sides), a quarter just past an edge, the rest anywhere
    a, b = rng.uniform(0, 1, n).astype(np.float32), rng.uniform(0, 1, n).astype(np.float32)
    flip = a + b > 1
    a[flip], b[flip] = 1 - a[flip], 1 - b[flip]
    edge = rng.rand(n) < 0.33
    b[edge] = (1 - a[edge]) * np.float32(1.000001)
    tgt = v0 + a[:, None] * (v1 - v0) + b[:, None] * (v2 - v0)
    d = (tgt - o).astype(np.float32) * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    anywhere = rng.rand(n) < 0.25
    d[anywhere] = rng.normal(size=(int(anywhere.sum()), 3)).astype(np.float32)
    v2[::97] = v1[::97]          # degenerate triangles: determinant 0
    d[5::101] = 0.0              # zero directions
    hit, t, u, v = gpu_ctx.intersect_triangle1(o, d, v0, v1, v2)
    rt, ru, rv = C.c_float(), C.c_float(), C.c_float()
    P = lambda x, i: x[i].ctypes.data  # noqa: E731
    want_hit = np.zeros(n, np.int32)
    want = np.zeros((n, 3), np.float32)
    for i in range(n):
        rt.value = ru.value = rv.value = 0.0
        want_hit[i] = ref.ptref_intersect_triangle1(P(o, i), P(d, i), P(v0, i), P(v1, i), P(v2, i), C.byref(rt), C.byref(ru), C.byref(rv))
        if want_hit[i]:
            want[i] = (rt.value, ru.value, rv.value)
    assert 0.3 < want_hit.mean() < 0.8
    assert np.array_equal(hit, want_hit), "%d hit flags differ from the reference build" % int((hit != want_hit).sum())
    got = np.stack([t, u, v], 1)
    m = want_hit == 1
    assert np.array_equal(got[m].view(np.uint32), want[m].view(np.uint32))
