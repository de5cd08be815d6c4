// cos / sin of the keypoint steering angle, shared by the HIP kernels and the host-side check (tests/cpp/cos_sin_check.cpp).
#pragma once
#if defined(__HIPCC__)
#define SD_HD __host__ __device__ __forceinline__
#else
#define SD_HD inline
#endif

struct sd_cs { float c, s; };

// cos / sin of an angle in [0, 2 pi] evaluated in f64 and rounded to f32 (oracle spec Q3: the f32 rounding of the f64 library
// value).  Quadrant reduction with a two-part pi/2 and the fdlibm kernel polynomials; ~35 f64 operations instead of the ~180 of
// the general-range library cos() + sin().  Only IEEE f64 multiply / fma / rint are used, so host and device agree bit for bit,
// and the host run of the check program compares it with glibc for EVERY f32 angle (degrees in [0.001, 360] x pi/180): 154 M
// values, no difference.
SD_HD sd_cs sd_cos_sin_f32(double x)
{
    const double kd = __builtin_rint(x * 0.63661977236758134308);
    const int n = (int)kd & 3;
    double r = __builtin_fma(-kd, 1.5707963267948966, x);
    r = __builtin_fma(-kd, 6.123233995736766e-17, r);
    const double z = r * r;
    const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                    2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double sn = __builtin_fma(z * r, ps, r);
    const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                    -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cs = 1.0 - __builtin_fma(0.5, z, -(z * z) * pc);
    const double c = (n & 1) ? sn : cs, s_ = (n & 1) ? cs : sn;
    sd_cs o;
    o.c = (float)((n == 1 || n == 2) ? -c : c);
    o.s = (float)((n >= 2) ? -s_ : s_);
    return o;
}
