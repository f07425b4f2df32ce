// FW::Random stand-in.  The reference's generator lives in the absent FW framework (algorithm unknown,
// "parity unpinned"), so this build defines one: a counter-based hash stream (key from up to four seed words, draw n =
// mix(key ^ mix(n * golden + c))).  Counter-based so the GPU can give every path sample its own independent stream;
// the kernels and the CPU oracle implement the same function bit for bit.
#pragma once
#include "Defs.hpp"

namespace FW {

class Random {
public:
    explicit Random(U32 seed = 0) { reset(seed, 0, 0, 0); }
    Random(U32 k0, U32 k1, U32 k2, U32 k3) { reset(k0, k1, k2, k3); }
    void reset(U32 k0, U32 k1 = 0, U32 k2 = 0, U32 k3 = 0) {
        m_key = mix(k3 + mix(k2 + mix(k1 + mix(k0 + 0x9E3779B9u))));
        m_ctr = 0;
    }
    U32 getU32() { return at(m_ctr++); }
    U32 getU32(U32 lo, U32 hi) { return (hi <= lo) ? lo : lo + getU32() % (hi - lo); }
    S32 getS32(S32 lo, S32 hi) { return (hi <= lo) ? lo : lo + (S32)(getU32() % (U32)(hi - lo)); }  // hi exclusive
    F32 getF32() { return (F32)getU32() * (1.0f / 4294967296.0f); }
    F32 getF32(F32 lo, F32 hi) { return getF32() * (hi - lo) + lo; }
    U32 key() const { return m_key; }
    U32 counter() const { return m_ctr; }
    void setCounter(U32 c) { m_ctr = c; }  // the device hands back where a sample left the stream (ptgpu_trace_samples)

private:
    static U32 mix(U32 x) {
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
    }
    U32 at(U32 n) const { return mix(m_key ^ mix(n * 0x9E3779B9u + 0x7F4A7C15u)); }
    U32 m_key, m_ctr;
};

}  // namespace FW
