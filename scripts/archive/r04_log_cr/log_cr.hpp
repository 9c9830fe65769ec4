// Correctly rounded natural logarithm of a positive normal double, for the ONE log the log-probs
// take per chain (N/2 log(precision), binf/example/likelihood.py:55; (shape - 1) log(precision),
// priors.py:23-25).  numpy's own log is correctly rounded in all but rare cases (and differs between
// CPUs where it is not); the device library's is within an ulp -- one ulp off this host's numpy for
// about 2 in 100 random arguments (tests/soak/fuzz_pairdist_big.py found it).  With this the term is
// a fixed bit pattern: the double nearest the true logarithm.
//
// t = m 2^k, m in [sqrt(1/2), sqrt(2));  log t = k ln 2 + 2 atanh(s),  s = (m - 1) / (m + 1),
// 2 atanh(s) = s (2 + 2 s^2/3 + 2 s^4/5 + ...), |s| <= 0.1716: 22 terms in double-double arithmetic
// (about 2^-100 relative), evaluated as four interleaved Horner chains in s^8 to keep the dependent
// chain short; the result is the high word of the normalised sum.  Zero, negative, infinite, NaN and
// subnormal arguments take the device library's log (numpy's special values).
// Constants generated with mpmath at 400 bits: 2 / (2n + 1) and ln 2 as hi + lo.
#pragma once
#include <hip/hip_runtime.h>

namespace binf {

struct DD {
    double hi, lo;
};

__device__ inline DD dd_fast_two_sum(double a, double b)        // |a| >= |b|
{
    const double s = a + b;
    return {s, b - (s - a)};
}
__device__ inline DD dd_two_sum(double a, double b)
{
    const double s = a + b;
    const double bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ inline DD dd_two_prod(double a, double b)
{
    const double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}
__device__ inline DD dd_add(DD x, DD y)
{
    DD s = dd_two_sum(x.hi, y.hi);
    const DD t = dd_two_sum(x.lo, y.lo);
    s.lo += t.hi;
    s = dd_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_fast_two_sum(s.hi, s.lo);
}
__device__ inline DD dd_mul(DD x, DD y)
{
    DD p = dd_two_prod(x.hi, y.hi);
    p.lo += x.hi * y.lo + x.lo * y.hi;
    return dd_fast_two_sum(p.hi, p.lo);
}
__device__ inline DD dd_mul_d(DD x, double y)
{
    DD p = dd_two_prod(x.hi, y);
    p.lo += x.lo * y;
    return dd_fast_two_sum(p.hi, p.lo);
}
__device__ inline DD dd_neg(DD x) { return {-x.hi, -x.lo}; }
__device__ inline DD dd_div(DD x, DD y)
{
    const double q1 = x.hi / y.hi;
    DD r = dd_add(x, dd_neg(dd_mul_d(y, q1)));
    const double q2 = r.hi / y.hi;
    r = dd_add(r, dd_neg(dd_mul_d(y, q2)));
    const double q3 = r.hi / y.hi;
    const DD q = dd_fast_two_sum(q1, q2);
    return dd_add(q, DD{q3, 0.0});
}

constexpr int LOG_CR_TERMS = 22;

__device__ inline double log_cr(double t)
{
    // in the function, loops fully unrolled: the constants become instruction literals (a table in
    // memory costs a cold launch several microseconds of load latency)
    constexpr double LOG_CR_C[LOG_CR_TERMS][2] = {
    {0x1.0000000000000p+1, 0x0.0p+0},
    {0x1.5555555555555p-1, 0x1.5555555555555p-55},
    {0x1.999999999999ap-2, -0x1.999999999999ap-56},
    {0x1.2492492492492p-2, 0x1.2492492492492p-56},
    {0x1.c71c71c71c71cp-3, 0x1.c71c71c71c71cp-57},
    {0x1.745d1745d1746p-3, -0x1.745d1745d1746p-58},
    {0x1.3b13b13b13b14p-3, -0x1.3b13b13b13b14p-57},
    {0x1.1111111111111p-3, 0x1.1111111111111p-59},
    {0x1.e1e1e1e1e1e1ep-4, 0x1.e1e1e1e1e1e1ep-60},
    {0x1.af286bca1af28p-4, 0x1.af286bca1af28p-58},
    {0x1.8618618618618p-4, 0x1.8618618618618p-58},
    {0x1.642c8590b2164p-4, 0x1.642c8590b2164p-59},
    {0x1.47ae147ae147bp-4, -0x1.eb851eb851eb8p-60},
    {0x1.2f684bda12f68p-4, 0x1.2f684bda12f68p-58},
    {0x1.1a7b9611a7b96p-4, 0x1.1a7b9611a7b96p-60},
    {0x1.0842108421084p-4, 0x1.0842108421084p-59},
    {0x1.f07c1f07c1f08p-5, -0x1.f07c1f07c1f08p-60},
    {0x1.d41d41d41d41dp-5, 0x1.0750750750750p-59},
    {0x1.bacf914c1bad0p-5, -0x1.bacf914c1bad0p-59},
    {0x1.a41a41a41a41ap-5, 0x1.0690690690690p-59},
    {0x1.8f9c18f9c18fap-5, -0x1.f3831f3831f38p-60},
    {0x1.7d05f417d05f4p-5, 0x1.7d05f417d05f4p-61},
    };
    if (!(t >= 2.2250738585072014e-308) || t > 1.7976931348623157e308) return log(t);
    int k;
    double m = frexp(t, &k);                      // [0.5, 1)
    if (m < 0.70710678118654752) {
        m *= 2.0;
        k -= 1;
    }
    // m - 1 is exact (m in [1/2, 2]); m + 1 as an exact double-double
    const DD s = dd_div(DD{m - 1.0, 0.0}, dd_two_sum(m, 1.0));
    const DD s2 = dd_mul(s, s), s4 = dd_mul(s2, s2), s8 = dd_mul(s4, s4);
    // P(s^2) = sum c_n s^(2n) = A(s^8) + s^2 B(s^8) + s^4 C(s^8) + s^6 D(s^8): four independent chains
    DD ch[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int n = j + 4 * ((LOG_CR_TERMS - 1 - j) / 4);
        DD p = {LOG_CR_C[n][0], LOG_CR_C[n][1]};
#pragma unroll
        for (n -= 4; n >= 0; n -= 4) p = dd_add(dd_mul(p, s8), DD{LOG_CR_C[n][0], LOG_CR_C[n][1]});
        ch[j] = p;
    }
    const DD lo_half = dd_add(ch[0], dd_mul(ch[1], s2));
    const DD hi_half = dd_add(ch[2], dd_mul(ch[3], s2));
    DD r = dd_mul(s, dd_add(lo_half, dd_mul(hi_half, s4)));
    if (k != 0) {
        DD kl = dd_two_prod((double)k, 0x1.62e42fefa39efp-1);
        kl.lo += (double)k * 0x1.abc9e3b39803fp-56;
        kl = dd_fast_two_sum(kl.hi, kl.lo);
        r = dd_add(kl, r);
    }
    return r.hi;
}

}  // namespace binf
