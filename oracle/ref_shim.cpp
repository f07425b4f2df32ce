// ORACLE -- TEST INFRASTRUCTURE ONLY.
// C-ABI shim around the parts of the reference that compile from their own source files with no
// stand-in for the absent FW framework: sobol.cpp/sobol.h (third-party Sobol sampler) and
// rtIntersect.inl (plain-float Moller-Trumbore / tri-box tests).  The reference sources are
// compiled WHERE THEY LIE (-I/root/reference/src); nothing is copied.  Output: oracle/_ref/libptref.so
// (git-ignored, travels to the GPU box).  Everything else in the reference's hot path includes
// FW headers that are not in its tree => unbuildable here without stand-ins.
#include <math.h>

#include "sobol.h"
#define __forceinline inline  // MSVC keyword used at rtIntersect.inl:1
#include "rtIntersect.inl"

extern "C" {
float ptref_sobol_sample(unsigned long long index, unsigned dimension, unsigned scramble) {
    return sobol::sample(index, dimension, scramble);
}
const unsigned* ptref_sobol_matrices(void) { return sobol::Matrices::matrices; }
int ptref_intersect_triangle1(const float* o, const float* d, const float* v0, const float* v1, const float* v2,
                              float* t, float* u, float* v) {
    return intersect_triangle1(o, d, v0, v1, v2, *t, *u, *v) ? 1 : 0;
}
int ptref_intersect_triangle2(const float* o, const float* d, const float* v0, const float* v1, const float* v2,
                              float* t, float* u, float* v) {
    return intersect_triangle2(o, d, v0, v1, v2, *t, *u, *v) ? 1 : 0;
}
int ptref_tri_box_overlap(const float* boxcenter, const float* boxhalfsize, const float* v0, const float* v1, const float* v2) {
    return triBoxOverlap(boxcenter, boxhalfsize, v0, v1, v2) ? 1 : 0;
}
}
