/* vslam_trig.h -- float sin/cos that reproduce glibc >= 2.28 sinf/cosf bit for bit.
 *
 * Why: the reference rotates the rBRIEF pattern with (float)cos(angle), (float)sin(angle)
 * (src/geometry/fextractor.cpp:103-104), i.e. the host libm's cosf/sinf, and then rounds sample
 * coordinates with cvRound -- a one-ulp difference in a or b can move a sample by a pixel.  Device
 * libm (ocml) is not glibc, so the kernel evaluates glibc's own algorithm: the ARM "optimized
 * routines" sincosf (sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h}): argument reduction and
 * polynomial in double, one final rounding to float.  The constants below are the published
 * __sincosf_table values (also readable in libm.so.6's .rodata).
 *
 * Domain used here: 0 <= y < 120 (angles are degrees in [0,360] times pi/180).  Larger inputs take
 * glibc's reduce_large path, which this header does not restate; they return NaN so a misuse is loud.
 *
 * Must be compiled with -ffp-contract=off (glibc's generic build does not contract; its FMA ifunc
 * variant differs from this in < 1e-8 of all inputs, see tests/test_trig.py).
 */
#ifndef VSLAM_TRIG_H
#define VSLAM_TRIG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define VSLAM_HD __host__ __device__ __forceinline__
#else
#define VSLAM_HD static inline
#endif

namespace vslam_trig {

VSLAM_HD uint32_t as_u32(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
}
VSLAM_HD uint32_t abstop12(float x) { return (as_u32(x) >> 20) & 0x7ff; }

/* polynomial coefficients: c0..c4 cosine, s1..s3 sine; table 1 = cosine coefficients negated */
#define VSLAM_HPI_INV 0x1.45F306DC9C883p+23 /* 2/pi * 2^24 */
#define VSLAM_HPI 0x1.921FB54442D18p0       /* pi/2 */
#define VSLAM_C0 0x1p0
#define VSLAM_C1 -0x1.ffffffd0c621cp-2
#define VSLAM_C2 0x1.55553e1068f19p-5
#define VSLAM_C3 -0x1.6c087e89a359dp-10
#define VSLAM_C4 0x1.99343027bf8c3p-16
#define VSLAM_S1 -0x1.555545995a603p-3
#define VSLAM_S2 0x1.1107605230bc4p-7
#define VSLAM_S3 -0x1.994eb3774cf24p-13

/* sinf_poly(x, x2, p, n): n odd -> cosine polynomial; neg selects table 1 (negated cosine). */
VSLAM_HD float sinf_poly(double x, double x2, int neg, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = VSLAM_S2 + x2 * VSLAM_S3;
        double x7 = x3 * x2;
        double s = x + x3 * VSLAM_S1;
        return (float)(s + x7 * s1);
    } else {
        const double sg = neg ? -1.0 : 1.0; /* exact sign flips of the constants */
        double x4 = x2 * x2;
        double c2 = sg * VSLAM_C3 + x2 * (sg * VSLAM_C4);
        double c1 = sg * VSLAM_C0 + x2 * (sg * VSLAM_C1);
        double x6 = x4 * x2;
        double c = c1 + x4 * (sg * VSLAM_C2);
        return (float)(c + x6 * c2);
    }
}

VSLAM_HD double reduce_fast(double x, int* np) {
    double r = x * VSLAM_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * VSLAM_HPI;
}

VSLAM_HD float quiet_nan() {
    union { uint32_t u; float f; } c;
    c.u = 0x7fc00000u;
    return c.f;
}

VSLAM_HD float glibc_sinf(float y) {
    double x = y;
    if (abstop12(y) < 0x3f4) { /* abstop12(pi/4) */
        double s = x * x;
        if (abstop12(y) < 0x398) return y; /* |y| < 2^-12 */
        return sinf_poly(x, s, 0, 0);
    } else if (abstop12(y) < 0x42f) { /* |y| < 120 */
        int n;
        x = reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; /* sign[] = {1,-1,-1,1} */
        return sinf_poly(x * s, x * x, (n & 2) != 0, n);
    }
    return quiet_nan();
}

VSLAM_HD float glibc_cosf(float y) {
    double x = y;
    if (abstop12(y) < 0x3f4) {
        double x2 = x * x;
        if (abstop12(y) < 0x398) return 1.0f;
        return sinf_poly(x, x2, 0, 1);
    } else if (abstop12(y) < 0x42f) {
        int n;
        x = reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return sinf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    }
    return quiet_nan();
}

/* glibc logf (sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c, glibc >= 2.27): MapPoint::PredictScale's
 * `log(ratio)` on a float (mappoint.cpp:514, std::log(float)).  Normal positive inputs only (ratio of two
 * positive distances); zero, negative, subnormal, inf and NaN return NaN so a misuse is loud.  The same text in
 * oracle/orb_oracle.cpp was compared with the platform libm for every positive finite float. */
VSLAM_HD float glibc_logf(float x) {
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    const uint32_t ix = as_u32(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return quiet_nan();
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    union { uint32_t u; float f; } zc;
    zc.u = ix - (tmp & (0x1ffu << 23));
    const double z = (double)zc.f;
    const double r = z * invc[i] - 1;
    const double y0 = logc[i] + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

} // namespace vslam_trig
#endif
