// sin / cos of x * 2^f for the positional encoding (Embedder, run_nerf_helpers.py:41-48: the frequency bands are
// exact powers of two), one argument reduction per coordinate instead of one per frequency.
//
// x * 2/pi is kept as an unevaluated fp32 sum h + l (product error recovered with an fma, ~2^-47 relative), so
// scaling by 2^f is exact for every band; per band: q = rint(h 2^f), r = (h 2^f - q) + l 2^f (rounded once more: l 2^f is not small against 1
// when ulp(h 2^f) nears 1/4) in [-1/2, 1/2]
// quarter turns, then the Cephes single-precision kernels on r pi/2 in [-pi/4, pi/4] and a rotation by q mod 4.
// Measured against sin/cos evaluated in double precision of the same fp32 argument (tests/test_host_cpu.py,
// tools/probes/pe_sincos_check.cpp): <= 1.2e-7 absolute (2 ulp of one) for |x 2^f| < 2^22; beyond that callers
// fall back to sincosf.  ~30 VALU instructions per band against ~65 for a library sincosf.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PE_HD __host__ __device__ __forceinline__
#else
#define PE_HD inline
#endif

struct PeTurns {
    float h, l;   // x * 2/pi = h + l
};

constexpr float PE_FAST_LIMIT = 4194304.0f;   // 2^22: |x 2^f| below this takes the shared reduction

PE_HD PeTurns pe_turns(float x) {
    const float C_HI = 0.636619746685028076171875f;      // fl32(2/pi)
    const float C_LO = 2.5682553267268e-08f;             // 2/pi - C_HI
    PeTurns t;
    t.h = x * C_HI;
    t.l = fmaf(x, C_LO, fmaf(x, C_HI, -t.h));
    return t;
}

// sin(x 2^f), cos(x 2^f) from t = pe_turns(x); scale = 2^f as a float
PE_HD void pe_sincos(const PeTurns t, const float scale, float* s_out, float* c_out) {
    const float h = t.h * scale, l = t.l * scale;        // exact
    const float q1 = rintf(h);
    const float r1 = (h - q1) + l;                       // h - q1 is exact; |l| reaches 1/4 once ulp(h) does,
    const float q2 = rintf(r1);                          // so round once more
    const float r = r1 - q2;
    const float PIO2_HI = 1.57079637050628662109375f, PIO2_LO = -4.37113900018624283e-08f;
    const float a = fmaf(r, PIO2_HI, r * PIO2_LO);
    const float z = a * a;
    // Cephes sinf / cosf kernels on [-pi/4, pi/4]
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float s = fmaf(a * z, ps, a);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float c = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int n = (int)q1 + (int)q2;
    const float ss = (n & 1) ? c : s, cc = (n & 1) ? s : c;
    *s_out = (n & 2) ? -ss : ss;
    *c_out = ((n + 1) & 2) ? -cc : cc;
}
