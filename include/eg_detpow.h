/*
 * eg_detpow.h — x^p for x in (0, 1], p in [1, 8], from IEEE + - * / and integer bit operations only.
 *
 * The reference's stalled-policy sampler raises every weight to a power (w.powf(power_scaling),
 * ai/learning/weights/sampling.rs:190-220).  libm's pow is not available bit-for-bit on the GPU, so the CPU oracle
 * and the HIP kernel both use this routine instead: compiled without FMA contraction it returns the same bits on
 * x86-64 and gfx950, and it agrees with glibc's pow to < 2e-14 relative on the weight domain (tests/test_detpow.py),
 * i.e. far inside the 1e-5 tolerance of the float terms.  exp(p * ln x) with
 *   ln x  = e*ln2 + 2*atanh((m-1)/(m+1)),  m in [sqrt(1/2), sqrt(2))   (odd series to s^23)
 *   exp r = 2^k * sum_{n<=14} r^n / n!,    |r| <= ln2/2
 * Define EG_DETPOW_QUAL before including (e.g. `static inline` or `__device__ __forceinline__`).
 */
#ifndef EG_DETPOW_H
#define EG_DETPOW_H

#ifndef EG_DETPOW_QUAL
#define EG_DETPOW_QUAL static inline
#endif

EG_DETPOW_QUAL double eg_detpow_from_bits(unsigned long long b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
EG_DETPOW_QUAL unsigned long long eg_detpow_to_bits(double d) { unsigned long long b; __builtin_memcpy(&b, &d, 8); return b; }

EG_DETPOW_QUAL double eg_detlog(double x) {   /* x normal and positive */
  unsigned long long bits = eg_detpow_to_bits(x);
  int e = (int)((bits >> 52) & 0x7FFull) - 1023;
  double m = eg_detpow_from_bits((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);   /* [1, 2) */
  if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }                                   /* [sqrt(1/2), sqrt(2)) */
  const double s = (m - 1.0) / (m + 1.0);
  const double s2 = s * s;
  double t = 1.0 / 23.0;
  t = t * s2 + 1.0 / 21.0; t = t * s2 + 1.0 / 19.0; t = t * s2 + 1.0 / 17.0; t = t * s2 + 1.0 / 15.0;
  t = t * s2 + 1.0 / 13.0; t = t * s2 + 1.0 / 11.0; t = t * s2 + 1.0 / 9.0;  t = t * s2 + 1.0 / 7.0;
  t = t * s2 + 1.0 / 5.0;  t = t * s2 + 1.0 / 3.0;  t = t * s2 + 1.0;
  const double lnm = 2.0 * s * t;
  const double ed = (double)e;
  return (ed * 6.93147180369123816490e-01 + lnm) + ed * 1.90821492927058770002e-10;       /* ln2 = hi + lo */
}

EG_DETPOW_QUAL double eg_detexp(double y) {   /* |y| < 700 */
  const double kf = y * 1.4426950408889634;   /* y / ln2 */
  const int k = (int)(kf < 0.0 ? kf - 0.5 : kf + 0.5);
  const double kd = (double)k;
  const double r = (y - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
  double t = 1.0 / 87178291200.0;             /* 1/14! */
  t = t * r + 1.0 / 6227020800.0; t = t * r + 1.0 / 479001600.0; t = t * r + 1.0 / 39916800.0; t = t * r + 1.0 / 3628800.0;
  t = t * r + 1.0 / 362880.0;     t = t * r + 1.0 / 40320.0;     t = t * r + 1.0 / 5040.0;     t = t * r + 1.0 / 720.0;
  t = t * r + 1.0 / 120.0;        t = t * r + 1.0 / 24.0;        t = t * r + 1.0 / 6.0;        t = t * r + 0.5;
  t = t * r + 1.0;                t = t * r + 1.0;
  return t * eg_detpow_from_bits((unsigned long long)(k + 1023) << 52);
}

EG_DETPOW_QUAL double eg_detpow(double x, double p) { return eg_detexp(p * eg_detlog(x)); }

#endif
