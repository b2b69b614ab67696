/*
 * sogm_detmath.h — deterministic fp64 cbrt / cos / acos built only from IEEE-754 correctly rounded
 * operations (+ - * / sqrt) and exact exponent manipulation.
 *
 * Why: the hybrid A* heuristic (path_searching/src/fake_risk_hybrid_a_star.cpp:525-587) calls
 * cbrt/acos/cos.  glibc and the ROCm device library round these differently in the last ulp, and
 * the north_star demands bit-exact A* expansions between the CPU oracle and the HIP path.  Both
 * sides therefore evaluate these three functions with this one header (plain C++, no algorithm
 * from the reference in it); tests/test_detmath.py bounds their error against libm.
 * Compile with -ffp-contract=off on both sides.
 */
#ifndef SOGM_DETMATH_H
#define SOGM_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SOGM_HD __host__ __device__ inline
#else
#define SOGM_HD inline
#endif

namespace sogm_det {

SOGM_HD double sqrt_rn(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __dsqrt_rn(x);
#else
  return __builtin_sqrt(x);
#endif
}

SOGM_HD uint64_t bits_of(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
SOGM_HD double from_bits(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}

/* x * 2^k for normal results, exact */
SOGM_HD double scale2(double x, int k) { return x * from_bits((uint64_t)(1023 + k) << 52); }

SOGM_HD double fabs_(double x) { return x < 0 ? -x : x; }

/* cube root: exponent split, then division-free Newton on t = m^(-1/3) and one Newton step on y = m t^2.
 * (A Newton iteration on y itself divides by 3 y^2 every step: seven dependent fp64 divisions cost the A* child
 * evaluation ~0.5 us per call on the GPU.)  Initial guess: a quadratic per octave of m in [1,8), relative error
 * 1.8e-3; t <- t + t (1 - m t^3) / 3 squares the error twice (-> 8e-11); the closing step
 * y <- y - (y^3 - m) t^2 / 3 leaves the rounding of its own five operations: <= 0.96 ulp from the exact root over
 * 2e7 samples, exact on perfect cubes (tests/test_detmath.py). */
SOGM_HD double cbrt(double x) {
  if (x == 0.0 || x != x) return x;
  const bool neg = x < 0;
  double     a   = neg ? -x : x;
  uint64_t   u   = bits_of(a);
  int        e   = (int)(u >> 52) & 0x7ff;
  if (e == 0x7ff) return x; /* inf */
  int shift = 0;
  if (e == 0) { /* subnormal: scale up by 2^54 (multiple of 3) */
    a     = a * 18014398509481984.0;
    u     = bits_of(a);
    e     = (int)(u >> 52) & 0x7ff;
    shift = -18;
  }
  const int    ex = e - 1023;                            /* a = m * 2^ex, m in [1,2) */
  const int    q  = ex >= 0 ? ex / 3 : -((-ex + 2) / 3); /* floor(ex / 3) */
  const int    r  = ex - 3 * q;                          /* 0,1,2 */
  const double m  = from_bits((u & 0x000fffffffffffffULL) | ((uint64_t)(1023 + r) << 52)); /* [2^r, 2^(r+1)) */
  const double c2 = r == 0 ? 0.09339756192967737 : (r == 1 ? 0.018532423507304427 : 0.00367729857137654);
  const double c1 = r == 0 ? -0.4833214344056348 : (r == 1 ? -0.19180623835357227 : -0.07611835613412744);
  const double c0 = r == 0 ? 1.38815612815435 : (r == 1 ? 1.1017802490641597 : 0.8744835632011081);
  const double third = 1.0 / 3.0;
  double       t     = (c2 * m + c1) * m + c0;
  for (int i = 0; i < 2; ++i) {
    const double t3 = (t * t) * t;
    t               = t + t * ((1.0 - m * t3) * third);
  }
  const double t2 = t * t;
  double       y  = m * t2;
  const double y2 = y * y;
  y               = y - (y2 * y - m) * (t2 * third);
  y               = scale2(y, q + shift);
  return neg ? -y : y;
}

/* sin and cos on |r| <= pi/4 (Taylor, fixed Horner order) */
SOGM_HD double sin_k(double r) {
  const double z = r * r;
  double       p = -7.6471637318198164759e-13;         /* -1/15! */
  p              = p * z + 1.6059043836821614599e-10;  /*  1/13! */
  p              = p * z + -2.5052108385441718775e-08; /* -1/11! */
  p              = p * z + 2.7557319223985890653e-06;  /*  1/9!  */
  p              = p * z + -1.9841269841269841270e-04; /* -1/7!  */
  p              = p * z + 8.3333333333333333333e-03;  /*  1/5!  */
  p              = p * z + -1.6666666666666666667e-01; /* -1/3!  */
  return r + r * (z * p);
}
SOGM_HD double cos_k(double r) {
  const double z = r * r;
  double       p = 4.7794773323873852974e-14;          /*  1/16! */
  p              = p * z + -1.1470745597729724714e-11; /* -1/14! */
  p              = p * z + 2.0876756987868098979e-09;  /*  1/12! */
  p              = p * z + -2.7557319223985890653e-07; /* -1/10! */
  p              = p * z + 2.4801587301587301587e-05;  /*  1/8!  */
  p              = p * z + -1.3888888888888888889e-03; /* -1/6!  */
  p              = p * z + 4.1666666666666666667e-02;  /*  1/4!  */
  return 1.0 - 0.5 * z + z * (z * p);
}

/* cos for |x| < ~1e5 (three-term Cody-Waite reduction by pi/2) */
SOGM_HD double cos(double x) {
  const double a       = fabs_(x);
  const double kf      = (double)(long long)(a * 0.63661977236758134308 + 0.5);
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  const double PIO2_LL = 2.02226624879595063154e-21;
  double       r       = a - kf * PIO2_HI;
  r                    = r - kf * PIO2_LO;
  r                    = r - kf * PIO2_LL;
  const int k          = (int)((long long)kf & 3);
  switch (k) {
    case 0: return cos_k(r);
    case 1: return -sin_k(r);
    case 2: return -cos_k(r);
    default: return sin_k(r);
  }
}

/* asin on [0, ~0.51]: odd Taylor series, 26 terms, Horner in z = x^2 */
SOGM_HD double asin_small(double x) {
  const double z = x * x;
  /* c_n = (2n)! / (4^n (n!)^2 (2n+1)), n = 0..25: the doubles produced by the recurrence
   * c_n = c_{n-1} (2n-1)^2 / (2n (2n+1)) evaluated in IEEE double, written out so that no
   * divisions run per call (tests/test_detmath.py re-derives them) */
  const double c[26] = {1.0,
                        0.16666666666666666,
                        0.075,
                        0.044642857142857144,
                        0.030381944444444444,
                        0.022372159090909092,
                        0.017352764423076924,
                        0.01396484375,
                        0.011551800896139705,
                        0.009761609529194078,
                        0.008390335809616815,
                        0.0073125258735988454,
                        0.006447210311889649,
                        0.005740037670841924,
                        0.005153309682319905,
                        0.004660143486915096,
                        0.004240907093679363,
                        0.003880964558837669,
                        0.0035692053938259347,
                        0.003297059503473485,
                        0.0030578216492580306,
                        0.002846178401108942,
                        0.00265787063820729,
                        0.0024894486782468836,
                        0.002338091892111975,
                        0.0022014739737101384};
  double p = c[25];
  for (int n = 24; n >= 0; --n) p = p * z + c[n];
  return x * p;
}

SOGM_HD double acos(double x) {
  const double PI      = 3.14159265358979311600e+00;
  const double PIO2_HI = 1.57079632679489655800e+00;
  const double PIO2_LO = 6.12323399573676603587e-17;
  if (x != x) return x;
  if (x >= 1.0) return 0.0;
  if (x <= -1.0) return PI;
  const double a = fabs_(x);
  if (a <= 0.5) {
    const double s = asin_small(a);
    return x >= 0 ? (PIO2_HI - (s - PIO2_LO)) : (PIO2_HI + (s + PIO2_LO));
  }
  /* acos(a) = 2 asin(sqrt((1-a)/2)) for a in (0.5, 1) */
  const double t = sqrt_rn((1.0 - a) * 0.5);
  const double s = 2.0 * asin_small(t);
  return x >= 0 ? s : (PI - s);
}

/* natural log for finite x > 0: x = m 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh((m-1)/(m+1)) */
SOGM_HD double log(double x) {
  if (!(x > 0.0)) return x == 0.0 ? -1.0 / 0.0 : (x - x) / (x - x);
  uint64_t u = bits_of(x);
  int      e = (int)(u >> 52) & 0x7ff;
  if (e == 0x7ff) return x;
  int adj = 0;
  if (e == 0) {
    x   = x * 18014398509481984.0; /* 2^54 */
    u   = bits_of(x);
    e   = (int)(u >> 52) & 0x7ff;
    adj = -54;
  }
  int    ex = e - 1023 + adj;
  double m  = from_bits((u & 0x000fffffffffffffULL) | ((uint64_t)1023 << 52)); /* [1,2) */
  if (m > 1.41421356237309514547) {
    m  = m * 0.5;
    ex = ex + 1;
  }
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
  double       p = 1.0 / 27.0;
  p              = p * z + 1.0 / 25.0;
  p              = p * z + 1.0 / 23.0;
  p              = p * z + 1.0 / 21.0;
  p              = p * z + 1.0 / 19.0;
  p              = p * z + 1.0 / 17.0;
  p              = p * z + 1.0 / 15.0;
  p              = p * z + 1.0 / 13.0;
  p              = p * z + 1.0 / 11.0;
  p              = p * z + 1.0 / 9.0;
  p              = p * z + 1.0 / 7.0;
  p              = p * z + 1.0 / 5.0;
  p              = p * z + 1.0 / 3.0;
  const double lm    = 2.0 * (s + s * (z * p));
  const double LN2_H = 6.93147180369123816490e-01;
  const double LN2_L = 1.90821492927058770002e-10;
  const double ef    = (double)ex;
  return ef * LN2_H + (lm + ef * LN2_L);
}

/* exp(x) for a float x <= 0, returned as a float: the Gaussian weight of ParticleATC::getParticlesWithRisk's resample
 * branch (particles.cpp:399).  exp(x) = 2^k exp(r), k = nearest integer to x / ln 2, r = x - k ln 2 in two parts,
 * exp(r) by its Taylor series to r^13 (|r| <= 0.35: remainder < 1e-17) — fp64 operations only, so the oracle and the
 * device agree bit for bit; against libm's expf the float result differs by at most one ulp. */
SOGM_HD float expf_neg(float xf) {
  if (!(xf <= 0.0f)) return 1.0f;  /* (also NaN) */
  if (xf < -104.0f) return 0.0f;
  const double x     = (double)xf;
  const double LOG2E = 1.44269504088896338700e+00;
  const double LN2_H = 6.93147180369123816490e-01;
  const double LN2_L = 1.90821492927058770002e-10;
  const double kd    = (double)(long long)(x * LOG2E - 0.5);  /* x <= 0: truncation after -0.5 rounds to nearest */
  const double r     = (x - kd * LN2_H) - kd * LN2_L;
  double       p     = 1.0 / 6227020800.0;
  p                  = p * r + 1.0 / 479001600.0;
  p                  = p * r + 1.0 / 39916800.0;
  p                  = p * r + 1.0 / 3628800.0;
  p                  = p * r + 1.0 / 362880.0;
  p                  = p * r + 1.0 / 40320.0;
  p                  = p * r + 1.0 / 5040.0;
  p                  = p * r + 1.0 / 720.0;
  p                  = p * r + 1.0 / 120.0;
  p                  = p * r + 1.0 / 24.0;
  p                  = p * r + 1.0 / 6.0;
  p                  = p * r + 0.5;
  p                  = p * r + 1.0;
  p                  = p * r + 1.0;
  const int k        = (int)kd;  /* >= -151 */
  /* 2^k in two exact factors (k may be below the normal exponent range of one factor for a float-denormal result) */
  const int k1 = k / 2, k2 = k - k1;
  return (float)((p * from_bits((uint64_t)(1023 + k1) << 52)) * from_bits((uint64_t)(1023 + k2) << 52));
}

}  // namespace sogm_det
#endif
