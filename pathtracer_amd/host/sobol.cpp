#include "sobol.h"

namespace sobol {
const unsigned Matrices::num_dimensions;
const unsigned Matrices::size;
const unsigned Matrices::matrices[Matrices::num_dimensions * Matrices::size] = {
#include "../csrc/_gen/sobol_matrices.inc"
};
}  // namespace sobol
