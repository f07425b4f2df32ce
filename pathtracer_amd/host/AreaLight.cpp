#include "AreaLight.hpp"

#include "util.hpp"

namespace FW {

namespace {
// Sobol dimensions 4+4b / 5+4b for bounce b; the table has 1024 dimensions, so b saturates at 254
inline int lightDim(int bounce) { return 4 + 4 * (bounce > 254 ? 254 : bounce); }
}  // namespace

void AreaLight::sample(float& pdf, Vec3f& p, int base, int dimension, Random& rnd) {
    const Vec3f tangent = m_xform.getCol(0).getXYZ(), bitangent = m_xform.getCol(1).getXYZ();
    const int dim = lightDim(dimension);
    const int r = rnd.getS32(1, 10000);
    const unsigned long long idx = (unsigned long long)(long long)(base + r);
    float x = (2.0f * sobol::sample(idx, dim) - 1.0f) * m_size.x;
    float y = (2.0f * sobol::sample(idx, dim + 1) - 1.0f) * m_size.y;
    p = getPosition() + tangent * x + bitangent * y;
    pdf = 1.0f / (4.0f * m_size.x * m_size.y);
}

void AreaLight::sampleDisk(float& pdf, Vec3f& p, int base, int dimension, Random& rnd) {
    const Vec3f tangent = m_xform.getCol(0).getXYZ(), bitangent = m_xform.getCol(1).getXYZ();
    const int dim = lightDim(dimension);
    const int r = rnd.getS32(1, 1000);
    const unsigned long long idx = (unsigned long long)(long long)(base + r);
    float a = (2.0f * sobol::sample(idx, dim) - 1.0f) * m_size.x;
    float b = (2.0f * sobol::sample(idx, dim + 1) - 1.0f) * m_size.y;
    float x, y;
    mapToDisk(a, b, x, y);
    p = getPosition() + tangent * x + bitangent * y;
    pdf = 1.0f / (FW_PI * m_size.x * m_size.y);
}

}  // namespace FW
