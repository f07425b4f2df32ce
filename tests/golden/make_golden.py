#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own code (oracle/_ref/libptref.so = /root/reference/src/sobol.cpp and
rtIntersect.inl compiled where they lie).  Authoring container only; the JSON outputs are committed so the GPU box can
check the oracle and the kernels without the reference tree.  Usage: python tests/golden/make_golden.py"""
import ctypes as C
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po  # noqa: E402


def main():
    ref = po.ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    mats = np.ctypeslib.as_array(ref.ptref_sobol_matrices(), shape=(1024 * 52,)).astype("<u4")
    rng = np.random.RandomState(1234)
    samples = []
    for idx, dim in list(zip(rng.randint(0, 2 ** 31, 400).tolist(), rng.randint(0, 1024, 400).tolist())) + \
            [(0, 0), (1, 0), (2 ** 32 - 1, 0), (5, 1023), (123456, 3), (2 ** 31 - 1, 1021)]:
        bits = int(np.float32(ref.ptref_sobol_sample(idx, dim, 0)).view(np.uint32))
        samples.append([idx, dim, bits])
    json.dump(dict(source="reference src/sobol.cpp via oracle/_ref", matrices_crc32=zlib.crc32(mats.tobytes()), samples=samples),
              open(os.path.join(HERE, "sobol_golden.json"), "w"))
    cases = []
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    for i in range(300):
        tri = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
        o = rng.uniform(-2, 2, 3).astype(np.float32)
        target = (tri.mean(0) + rng.uniform(-0.6, 0.6, 3)).astype(np.float32)
        d = (target - o).astype(np.float32) * np.float32(rng.uniform(0.5, 2.0))
        hit = ref.ptref_intersect_triangle1(o.ctypes.data, d.ctypes.data, tri[0].ctypes.data, tri[1].ctypes.data, tri[2].ctypes.data,
                                            C.byref(t), C.byref(u), C.byref(v))
        c = dict(o=o.tolist(), d=d.tolist(), v0=tri[0].tolist(), v1=tri[1].tolist(), v2=tri[2].tolist(), hit=int(hit))
        if hit:
            c["tuv_bits"] = np.array([t.value, u.value, v.value], np.float32).view(np.uint32).tolist()
        cases.append(c)
    json.dump(dict(source="reference src/rtIntersect.inl intersect_triangle1 via oracle/_ref", cases=cases),
              open(os.path.join(HERE, "rtintersect_golden.json"), "w"))
    print("sobol samples:", len(samples), "triangle cases:", len(cases), "hits:", sum(c["hit"] for c in cases))


if __name__ == "__main__":
    main()
