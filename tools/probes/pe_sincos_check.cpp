// Host check of pl-nerf_amd/csrc/pe_sincos.h against double-precision sin/cos of the same fp32 argument.
//   g++ -O2 -ffp-contract=off -I pl-nerf_amd/csrc tools/probes/pe_sincos_check.cpp -o /tmp/pe_check && /tmp/pe_check
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "pe_sincos.h"

int main() {
    std::mt19937_64 rng(1);
    double worst_s = 0, worst_c = 0, worst_lib = 0; float wx = 0; int wf = 0;
    auto one = [&](float x, int f) {
        const float sc = (float)(1 << f);
        const float theta = x * sc;
        if (!(fabsf(theta) < PE_FAST_LIMIT)) return;   // callers use sincosf there
        float s, c;
        pe_sincos(pe_turns(x), sc, &s, &c);
        const double es = fabs((double)s - sin((double)theta)), ec = fabs((double)c - cos((double)theta));
        if (es > worst_s) { worst_s = es; wx = x; wf = f; }
        if (ec > worst_c) worst_c = ec;
        const double el = fabs((double)sinf(theta) - sin((double)theta));
        if (el > worst_lib) worst_lib = el;
    };
    std::uniform_real_distribution<float> u10(-10.f, 10.f), u1(-1.f, 1.f), ubig(-8000.f, 8000.f);
    for (int i = 0; i < 4000000; ++i) {
        const int f = i % 10;
        one(u10(rng), f); one(u1(rng), f); one(ubig(rng), f);
    }
    // arguments next to multiples of pi/2 (cancellation in the reduction)
    for (int k = -200000; k <= 200000; ++k) {
        const float x = (float)(k * 1.5707963267948966);
        for (int d = -2; d <= 2; ++d) {
            float y = x;
            for (int j = 0; j < abs(d); ++j) y = nextafterf(y, d > 0 ? 1e30f : -1e30f);
            for (int f = 0; f < 10; ++f) one(y, f);
        }
    }
    one(0.0f, 0); one(-0.0f, 3); one(4194303.0f / 512.f, 9); one(1e-30f, 9);
    printf("max abs err: sin %.3e  cos %.3e   (libm sinf on the same arguments: %.3e; 1 ulp of 1 = 5.96e-8)\n",
           worst_s, worst_c, worst_lib);
    printf("worst sin at x = %.9g (%a), f = %d\n", wx, wx, wf);
    return (worst_s < 1.2e-7 && worst_c < 1.2e-7) ? 0 : 1;
}
