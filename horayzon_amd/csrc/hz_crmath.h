/* hz_crmath.h -- the five float libm calls of the refraction branch (shadow_comp.cpp:135-159, :430-446:
 * acos, tan, pow, cos, sin on float arguments = the float overloads) as self-contained functions.
 *
 * Why not the platform's libm: the reference's result depends on it.  glibc 2.35's acosf / tanf are not
 * correctly rounded (0.25 % / 2 % of the arguments this path uses are off by one float ulp, measured by
 * tests/test_oracle.py::test_crmath_*), they come in FMA and non-FMA builds selected at run time, and other
 * platforms ship other routines -- there is no single "reference value" to be bit-equal to.  The contract
 * here is the CORRECTLY ROUNDED float result.  Each function evaluates in float64 with plain + - * / sqrt in
 * a fixed order (both users compile with -ffp-contract=off) to a relative error < 1e-14 and rounds once, so
 *   - the HIP kernels and the CPU oracle, which both include this header, agree bit for bit by construction;
 *   - the result is the correctly rounded float except when the exact value lies within 1e-14 (relative) of a
 *     rounding boundary (about 1 argument in 1e7).
 * Only the argument ranges of this path are covered accurately: acos [-1, 1]; tan [0, 1.6] rad; sin / cos
 * |x| <= pi/4 (here |x| < 0.02); pow base > 0 with |y log x| < 700.
 */
#ifndef HZ_CRMATH_H
#define HZ_CRMATH_H

#ifdef __HIPCC__
#define HZ_CRM __host__ __device__ static inline
#else
#define HZ_CRM static inline
#endif

/* Horner steps: HZ_CRM_FMA(p, z, c) = p z + c with two roundings (the contract).  HZ_CRM_FMAK is the same step for a coefficient
 * that is a literal: in the HIP kernels the coefficient is materialised into a scalar register pair AT the step (an empty asm
 * with an "s" constraint).  hipcc otherwise hoists the ~60 coefficients of the refraction branch out of k_shadow_refill's cell
 * loop into vector registers that live through the traversal -- 29 spilled VGPRs at the 72 registers of 7 workgroups per CU;
 * pinned: 1 -- and config 4 with refraction runs 3.6 % faster (140.6 -> 135.5 ms per 144 sun positions,
 * profiles/r05/ab_crmath_sgpr_pin.log).  Same instructions on the same values: results unchanged.
 * (Fused steps were measured in round 5 and dropped: profiles/r05/ab_crmath_fused_horner.log; the switch is in profiles/r06/lab_probes.patch.) */
#ifndef HZ_CRM_FMA
#if defined(__HIP_DEVICE_COMPILE__)
#define HZ_CRM_PINNED(c_) asm volatile("" : "+s"(c_))
#else
#define HZ_CRM_PINNED(c_) ((void)0)
#endif
#define HZ_CRM_FMA(a, b, c) ((a) * (b) + (c))
#define HZ_CRM_FMAK(a, b, c) ({ double c_ = (c); HZ_CRM_PINNED(c_); (a) * (b) + c_; })
#endif

/* x / c for a CONSTANT c with rc = the double nearest to 1 / c: q = x rc, r = x - c q (exact: one fused multiply-add),
 * q + r rc (+ the sign of x) -- four instructions instead of the eleven of a float64 division (v_div_scale x 2, v_rcp_f64, four Newton FMAs,
 * v_mul, v_fma, v_div_fmas, v_div_fixup).  This IS the correctly rounded quotient (Markstein's correction step with a correctly
 * rounded reciprocal), which is what the reference's `/ 180.0` and `/ M_PI` produce (deg2rad / rad2deg, shadow_comp.cpp:43-62:
 * float in, double arithmetic).  Only the HIP kernels use it; the oracle divides.  Not taken on trust:
 * tests/test_oracle.py::test_division_by_a_constant_is_the_ieee_quotient compares it with the IEEE division for EVERY finite
 * float x (the arguments are floats promoted to double) and both constants.  x = +-inf gives NaN here and +-inf there: the
 * refraction branch turns either into NaN directions (cos / sin of a non-finite angle) and the same shadow code. */
HZ_CRM double hz_crm_div_const(double x, double c, double rc) {
    const double q = x * rc;
    const double r = __builtin_fma(-c, q, x);
    return __builtin_copysign(__builtin_fma(r, rc, q), x);     /* (c > 0.  The sign: x = -0 would come out as +0 -- the only float the sweep found) */
}

#define HZ_CRM_PI_HI 3.141592653589793116      /* double nearest to pi          */
#define HZ_CRM_PI_LO 1.2246467991473532e-16    /* pi - HZ_CRM_PI_HI             */
#define HZ_CRM_LN2_HI 0.6931471803691238       /* ln 2, upper 33 bits           */
#define HZ_CRM_LN2_LO 1.9082149292705877e-10   /* ln 2 - HZ_CRM_LN2_HI          */

/* sin(r), cos(r) for |r| <= pi/4.  Round 5 (the refraction branch cost 40 % of a sun position): the
 * 10-term Taylor series of rounds 2-4 are replaced by
 *   |r| <= 2^-5 (the rotation angle of the refraction, < 0.015 rad): 5 / 5 Taylor terms, remainder < 1e-23 relative;
 *   else: the degree-13 / degree-14 minimax polynomials of fdlibm's k_sin.c / k_cos.c (|error| < 2^-58 on [-pi/4, pi/4]).
 * Both stay far inside the 1e-14 of the contract above. */
HZ_CRM double hz_crm_sin_k(double r) {
    const double z = r * r;
    double p;
    if (z <= 0.0009765625) {                                 /* |r| <= 2^-5 */
        p = -1.0 / 362880.0;                                   /* -1/9!  */
        p = HZ_CRM_FMAK(p, z, 1.0 / 5040.0);                              /*  1/7!  */
        p = HZ_CRM_FMAK(p, z, -1.0 / 120.0);                               /* -1/5!  */
        p = HZ_CRM_FMAK(p, z, 1.0 / 6.0);                                 /*  1/3!  */
        return r - (r * z) * p;
    }
    p = 1.58969099521155010221e-10;
    p = HZ_CRM_FMAK(p, z, -2.50507602534068634195e-08);
    p = HZ_CRM_FMAK(p, z, 2.75573137070700676789e-06);
    p = HZ_CRM_FMAK(p, z, -1.98412698298579493134e-04);
    p = HZ_CRM_FMAK(p, z, 8.33333333332248946124e-03);
    p = HZ_CRM_FMAK(p, z, -1.66666666666666324348e-01);
    return r + (r * z) * p;
}
HZ_CRM double hz_crm_cos_k(double r) {
    const double z = r * r;
    double p;
    if (z <= 0.0009765625) {
        p = -1.0 / 3628800.0;                                  /* -1/10! */
        p = HZ_CRM_FMAK(p, z, 1.0 / 40320.0);                             /*  1/8!  */
        p = HZ_CRM_FMAK(p, z, -1.0 / 720.0);                               /* -1/6!  */
        p = HZ_CRM_FMAK(p, z, 1.0 / 24.0);                                /*  1/4!  */
        p = HZ_CRM_FMAK(p, z, -0.5);                                       /* -1/2!  */
        return 1.0 + z * p;
    }
    p = -1.13596475577881948265e-11;
    p = HZ_CRM_FMAK(p, z, 2.08757232129817482790e-09);
    p = HZ_CRM_FMAK(p, z, -2.75573143513906633035e-07);
    p = HZ_CRM_FMAK(p, z, 2.48015872894767294178e-05);
    p = HZ_CRM_FMAK(p, z, -1.38888888888741095749e-03);
    p = HZ_CRM_FMAK(p, z, 4.16666666666666019037e-02);
    p = HZ_CRM_FMAK(p, z, -0.5);
    return 1.0 + z * p;
}

HZ_CRM float hz_crm_sinf(float x) { return (float)hz_crm_sin_k((double)x); }
HZ_CRM float hz_crm_cosf(float x) { return (float)hz_crm_cos_k((double)x); }

/* tan on [0, ~pi/2 + 0.03]: direct below pi/4, cotangent of the complement above (the float argument makes
 * pi/2 - x exact to 1e-16 absolute, and |pi/2 - x| >= 4e-8 for every float x) */
HZ_CRM float hz_crm_tanf(float xf) {
    const double x = (double)xf;
    if (!(x == x)) return xf;
    if (x < 0.0) return -hz_crm_tanf(-xf);
    if (x <= 0.78539816339744828) return (float)(hz_crm_sin_k(x) / hz_crm_cos_k(x));
    const double r = (0.5 * HZ_CRM_PI_HI - x) + 0.5 * HZ_CRM_PI_LO;     /* |r| <= pi/4 for x <= 3 pi/4 */
    return (float)(hz_crm_cos_k(r) / hz_crm_sin_k(r));
}

/* asin(t) for |t| <= 0.5: t + t^3 g(t^2), g = the degree-12 polynomial that interpolates (asin(t) / t - 1) / t^2 at the
 * Chebyshev nodes of t^2 in [0, 0.25] (computed in 80-bit arithmetic; |asin error| < 3e-16 relative, measured, double
 * rounding included).  Rounds 2-4 summed 26 Taylor terms. */
HZ_CRM double hz_crm_asin_k(double t) {
    const double z = t * t;
    double p = 0.034553784590501055;
    p = HZ_CRM_FMAK(p, z, -0.02364695072174073);
    p = HZ_CRM_FMAK(p, z, 0.023283466983300003);
    p = HZ_CRM_FMAK(p, z, 0.0031765613418358813);
    p = HZ_CRM_FMAK(p, z, 0.010889791739128478);
    p = HZ_CRM_FMAK(p, z, 0.011384931937545141);
    p = HZ_CRM_FMAK(p, z, 0.013981803080779488);
    p = HZ_CRM_FMAK(p, z, 0.017351599310672417);
    p = HZ_CRM_FMAK(p, z, 0.022372210988752014);
    p = HZ_CRM_FMAK(p, z, 0.030381943060510043);
    p = HZ_CRM_FMAK(p, z, 0.044642857161844712);
    p = HZ_CRM_FMAK(p, z, 0.074999999999905781);
    p = HZ_CRM_FMAK(p, z, 0.16666666666666666);
    return t + (t * z) * p;
}

HZ_CRM float hz_crm_acosf(float xf) {
    const double x = (double)xf;
    if (!(x >= -1.0 && x <= 1.0)) return (float)((x - x) / (x - x));   /* NaN, as acosf outside [-1, 1] */
    if (x >= -0.5 && x <= 0.5) return (float)((0.5 * HZ_CRM_PI_HI - hz_crm_asin_k(x)) + 0.5 * HZ_CRM_PI_LO);
    if (x > 0.5) return (float)(2.0 * hz_crm_asin_k(__builtin_sqrt((1.0 - x) * 0.5)));
    return (float)((HZ_CRM_PI_HI - 2.0 * hz_crm_asin_k(__builtin_sqrt((1.0 + x) * 0.5))) + HZ_CRM_PI_LO);
}

/* 2^k as a double for -1022 <= k <= 1023, built from its bit pattern */
HZ_CRM double hz_crm_pow2i(int k) {
    union { unsigned long long u; double d; } v;
    v.u = (unsigned long long)(k + 1023) << 52;
    return v.d;
}

/* log(x), x > 0 finite: x = m 2^e with m in [0.75, 1.5); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.2 */
HZ_CRM double hz_crm_log(double x) {
    union { double d; unsigned long long u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    if (e == -1023) {                                   /* subnormal: scale up */
        v.d = x * 18014398509481984.0;                  /* 2^54 */
        e = (int)((v.u >> 52) & 0x7ff) - 1023 - 54;
    }
    v.u = (v.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;   /* m in [1, 2) */
    double m = v.d;
    if (m >= 1.5) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 37.0;
    for (int n = 35; n >= 3; n -= 2) p = HZ_CRM_FMA(p, z, 1.0 / (double)n);
    const double lm = 2.0 * (s + (s * z) * p);
    return ((double)e * HZ_CRM_LN2_HI + lm) + (double)e * HZ_CRM_LN2_LO;
}

/* exp(t) for |t| < 700: t = k ln2 + r, |r| <= 0.35; Taylor to r^17 */
HZ_CRM double hz_crm_exp(double t) {
    const double kf = __builtin_floor(t * 1.4426950408889634 + 0.5);
    const int k = (int)kf;
    const double r = (t - kf * HZ_CRM_LN2_HI) - kf * HZ_CRM_LN2_LO;
    double p = 1.0 / 355687428096000.0;                  /* 1/17! */
    const double inv[16] = {1.0 / 20922789888000.0, 1.0 / 1307674368000.0, 1.0 / 87178291200.0, 1.0 / 6227020800.0,
                            1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                            1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0};
    for (int n = 0; n < 16; n++) p = p * r + inv[n];
    return (1.0 + r * p) * hz_crm_pow2i(k);
}

/* powf for the pressure formula (base = temperature ratio near 1, exponent 5.26): special cases as C99 pow
 * for the inputs this path can produce (non-positive or non-finite base) */
HZ_CRM float hz_crm_powf(float xf, float yf) {
    const double x = (double)xf, y = (double)yf;
    if (!(x == x) || !(y == y)) return xf + yf;
    if (x < 0.0) return (float)((x - x) / (x - x));       /* negative base, non-integer exponent: NaN */
    if (x == 0.0) return (y > 0.0) ? 0.0f : (float)(1.0 / (x * x));
    if (x > 1.7976931348623157e308) return (y > 0.0) ? xf : 0.0f;
    const double t = y * hz_crm_log(x);
    if (t > 700.0) return (float)(1.0e300 * 1.0e300);
    if (t < -700.0) return 0.0f;
    return (float)hz_crm_exp(t);
}

#endif /* HZ_CRMATH_H */
