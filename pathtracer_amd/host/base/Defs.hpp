// Minimal stand-in for the scalar typedefs/macros of the external FW framework that the reference's
// hot-path headers use.  Host-side only.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace FW {
typedef float F32;
typedef double F64;
typedef int32_t S32;
typedef uint32_t U32;
typedef int64_t S64;
typedef uint64_t U64;
}  // namespace FW

#define FW_PI 3.14159265358979323846f
#define FW_ASSERT(X) assert(X)
