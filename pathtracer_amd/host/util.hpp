#pragma once
namespace FW {
// Shirley & Chiu concentric square -> disk map, (a,b) in [-1,1]^2 (reference src/util.cpp:22-47)
void mapToDisk(const float a, const float b, float& x, float& y);
}
