"""MI355X-native implementation of Esgrove/pathtracer's per-sample path-tracing hot path.

Layout: csrc/ = HIP kernels + the C ABI (include/ptgpu.h -> libptgpu.so); host/ = C++ mirror of the reference's
Bvh / RayTracer / PathTraceRenderer / AreaLight classes over that ABI (libpthost.so); capi.py / scene.py = ctypes
plumbing for tests and bench.py.  There is no CPU fallback anywhere in this package.
"""
