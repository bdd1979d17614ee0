// dtof_half.h -- host side of the half-float node records (DNode16, dtof_scene.h): outward rounding of a box coordinate to IEEE binary16.
#pragma once
#include <stdint.h>
#include <cmath>
#include <cstring>

namespace dtof {

// IEEE binary16 bits of the nearest half BELOW (up = false) or ABOVE (up = true) a float of magnitude <= 65 000 (exact values map to themselves): the float is
// truncated to 10 fraction bits (toward zero) and stepped one half away from zero where truncation went the wrong way; halves below 2^-14 are subnormal.
// (tests/test_loader_and_abi.py checks both directions, exactness and tightness against every kind of input.)
inline uint16_t half_toward(float x, bool up) {
    if (x == 0.f) return 0;
    const bool neg = x < 0.f; const float a = std::fabs(x);
    uint32_t bits;                                   // magnitude, rounded toward zero
    bool exact;
    if (a < 6.103515625e-5f) {                       // below 2^-14: multiples of 2^-24
        const float q = a * 16777216.f; const uint32_t m = (uint32_t) q; bits = m; exact = (float) m == q;
    } else {
        uint32_t u; memcpy(&u, &a, 4);
        const uint32_t e = (u >> 23) - 127 + 15, m = (u >> 13) & 0x3ffu;
        bits = (e << 10) | m; exact = (u & 0x1fffu) == 0;
    }
    const bool away = neg != up;                     // up && positive, or down && negative: the magnitude has to grow
    if (!exact && away) ++bits;                      // (a carry out of the fraction moves into the exponent: still the next half)
    return (uint16_t) (bits | (neg ? 0x8000u : 0u));
}

}  // namespace dtof
