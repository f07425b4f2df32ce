// Host-side Sobol' sampler with the reference's call signature (reference src/sobol.h:28-49): 1024 dimensions x 52
// generator-matrix columns, 32-bit output scaled by 2^-32.  The matrix table is generated at build time by
// tools/gen_sobol_table.py from the Joe-Kuo direction numbers (never copied from the reference).
#pragma once
#include <cassert>

namespace sobol {

struct Matrices {
    static const unsigned num_dimensions = 1024;
    static const unsigned size = 52;
    static const unsigned matrices[];
};

inline float sample(unsigned long long index, const unsigned dimension, const unsigned scramble = 0U) {
    assert(dimension < Matrices::num_dimensions);
    unsigned bits = scramble;
    const unsigned* column = Matrices::matrices + dimension * Matrices::size;
    while (index) {
        if (index & 1ULL) bits ^= *column;
        index >>= 1;
        ++column;
    }
    return bits * (1.f / (1ULL << 32));
}

}  // namespace sobol
