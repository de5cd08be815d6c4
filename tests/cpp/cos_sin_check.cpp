// Host check of sd_cos_sin_f32 (slam-dynamic_amd/csrc/sd_trig.h) against the C library: usage  cos_sin_check [stride]
// stride 1 = every f32 angle in degrees [0.001, 360] (154 M values, ~7 s); the test suite runs stride 61.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "sd_trig.h"
int main(int argc, char** argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 61;
    const float factorPI = (float)(M_PI / 180.f);
    long bad = 0, n = 0;
    uint32_t lo, hi;
    const float flo = 1e-3f, fhi = 360.f;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    for (uint32_t bits = lo; bits <= hi; bits += stride) {
        float deg; memcpy(&deg, &bits, 4);
        const float ang = deg * factorPI;                    // computeOrbDescriptor: angle * (float)(CV_PI / 180.f)
        const sd_cs g = sd_cos_sin_f32((double)ang);
        const float rc = (float)cos((double)ang), rs = (float)sin((double)ang);
        n++;
        if (memcmp(&g.c, &rc, 4) || memcmp(&g.s, &rs, 4)) { if (bad++ < 5) printf("deg %.9g: cos %.9g vs %.9g, sin %.9g vs %.9g\n", deg, g.c, rc, g.s, rs); }
    }
    printf("%ld angles, %ld differ\n", n, bad);
    return bad ? 1 : 0;
}
