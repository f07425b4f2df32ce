#include "util.hpp"

#include <cmath>

#include "base/Defs.hpp"

void FW::mapToDisk(const float a, const float b, float& x, float& y) {
    // four wedges of the square; r = distance of the wedge's outer edge, phi proportional to the position along it
    float r, phi;
    const float q = FW_PI / 4.0f;
    if (a > -b) {
        if (a > b) { r = a; phi = q * (b / a); }
        else { r = b; phi = q * (2 - (a / b)); }
    } else if (a < b) {
        r = -a; phi = q * (4.0f + (b / a));
    } else {
        r = -b;
        phi = (b != 0) ? q * (6.0f - (a / b)) : 0;
    }
    x = r * std::cos(phi);
    y = r * std::sin(phi);
}
