// cr_atan2.h -- atan2 evaluated in double-double arithmetic and rounded once, for host and device.
//
// Why: libpedsim's social force takes the SIGN of the angle between two nearly parallel vectors
// (ped_agent.cpp:348-352, Tvector::angleTo = atan2 - atan2, ped_vector.cpp:141-152).  While two agents are at
// rest that angle is a rounding residue of the two atan2 calls, and its sign switches a force term of full
// magnitude on or off -- so the result depends on the last bit of the libm's atan2.  The reference links
// glibc, whose atan2 is correctly rounded except within ~0.02 ulp of a tie; OCML's device atan2 is not, and
// disagreed on most scenarios.  This implementation carries ~100 bits through the evaluation and rounds once,
// so it agrees with a correctly rounded atan2 (and hence with glibc) except in astronomically rare ties.
// It is only used for the tiny social-force crowds and the yaw of vector_states, never in a throughput loop.
//
// Method: q = min(|y|,|x|) / max(|y|,|x|) in double-double; atan(q) = atan(k/64) + atan(t) with
// t = (q - k/64) / (1 + q k/64), |t| <= 1/128, atan(k/64) from a 65-entry double-double table (generated with
// mpmath at 200 bits), atan(t) by its odd Taylor series with the first two terms in double-double; octant
// fix-ups with double-double pi and pi/2.  Explicit fma() calls are exact by definition, so -ffp-contract=off
// does not affect them.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CRA_HD __host__ __device__
#else
#define CRA_HD
#endif

struct cra_dd {
    double hi, lo;
};
CRA_HD inline cra_dd cra_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    cra_dd r;
    r.hi = s;
    r.lo = (a - (s - bb)) + (b - bb);
    return r;
}
CRA_HD inline cra_dd cra_fast_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    cra_dd r;
    r.hi = s;
    r.lo = b - (s - a);
    return r;
}
CRA_HD inline cra_dd cra_two_prod(double a, double b) {
    cra_dd r;
    r.hi = a * b;
    r.lo = fma(a, b, -r.hi);
    return r;
}
CRA_HD inline cra_dd cra_add(cra_dd a, cra_dd b) {
    cra_dd s = cra_two_sum(a.hi, b.hi);
    const cra_dd t = cra_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = cra_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return cra_fast_two_sum(s.hi, s.lo);
}
CRA_HD inline cra_dd cra_neg(cra_dd a) {
    cra_dd r;
    r.hi = -a.hi;
    r.lo = -a.lo;
    return r;
}
CRA_HD inline cra_dd cra_mul(cra_dd a, cra_dd b) {
    cra_dd p = cra_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return cra_fast_two_sum(p.hi, p.lo);
}
CRA_HD inline cra_dd cra_mul_d(cra_dd a, double b) {
    cra_dd p = cra_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return cra_fast_two_sum(p.hi, p.lo);
}
CRA_HD inline cra_dd cra_div(cra_dd a, cra_dd b) {  // ~104 bits
    const double q1 = a.hi / b.hi;
    cra_dd r = cra_add(a, cra_neg(cra_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = cra_add(r, cra_neg(cra_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    cra_dd q = cra_fast_two_sum(q1, q2);
    cra_dd q3d;
    q3d.hi = q3;
    q3d.lo = 0.0;
    return cra_add(q, q3d);
}

CRA_HD inline cra_dd cra_atan_table(int k) {
    // atan(k / 64), k = 0..64, as hi + lo
    const double T[65][2] = {
    {0x0.0p+0, 0x0.0p+0},
    {0x1.fff555bbb729bp-7, -0x1.220c39d4dff50p-61},
    {0x1.ffd55bba97625p-6, -0x1.5ec431444912cp-60},
    {0x1.7fb818430da2ap-5, -0x1.86ef8f794f105p-63},
    {0x1.ff55bb72cfdeap-5, -0x1.c934d86d23f1dp-60},
    {0x1.3f59f0e7c559dp-4, 0x1.ac4ce285df847p-58},
    {0x1.7ee182602f10fp-4, -0x1.cfb654c0c3d98p-58},
    {0x1.be39ebe6f07c3p-4, 0x1.f7b8f29a05987p-58},
    {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},
    {0x1.1e1fafb043727p-3, -0x1.b485914dacf8cp-59},
    {0x1.3d6eee8c6626cp-3, 0x1.61a3b0ce9281bp-57},
    {0x1.5c9811e3ec26ap-3, -0x1.054ab2c010f3dp-58},
    {0x1.7b97b4bce5b02p-3, 0x1.347b0b4f881cap-58},
    {0x1.9a6a8e96c8626p-3, 0x1.cf601e7b4348ep-59},
    {0x1.b90d7529260a2p-3, 0x1.17b10d2e0e5abp-61},
    {0x1.d77d5df205736p-3, 0x1.c648d1534597ep-57},
    {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},
    {0x1.09dc597d86362p-2, 0x1.62e47390cb865p-56},
    {0x1.18bf5a30bf178p-2, 0x1.30ca4748b1bf9p-57},
    {0x1.278372057ef46p-2, -0x1.077cdd36dfc81p-56},
    {0x1.362773707ebccp-2, -0x1.963a544b672d8p-57},
    {0x1.44aa436c2af0ap-2, -0x1.5d5e43c55b3bap-56},
    {0x1.530ad9951cd4ap-2, -0x1.2566480884082p-57},
    {0x1.614840309cfe2p-2, -0x1.a725715711f00p-56},
    {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},
    {0x1.7d5604b63b3f7p-2, 0x1.69c885c2b249ap-56},
    {0x1.8b24d394a1b25p-2, 0x1.b6d0ba3748fa8p-56},
    {0x1.98cd5454d6b18p-2, 0x1.9e6c988fd0a77p-56},
    {0x1.a64eec3cc23fdp-2, -0x1.24dec1b50b7ffp-56},
    {0x1.b3a911da65c6cp-2, 0x1.ae187b1ca5040p-56},
    {0x1.c0db4c94ec9f0p-2, -0x1.cc1ce70934c34p-56},
    {0x1.cde53432c1351p-2, -0x1.a2cfa4418f1adp-56},
    {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},
    {0x1.e77eb7f175a34p-2, 0x1.0e53dc1bf3435p-56},
    {0x1.f40dd0b541418p-2, -0x1.a3992dc382a23p-57},
    {0x1.0039c73c1a40cp-1, -0x1.b32c949c9d593p-55},
    {0x1.0657e94db30d0p-1, -0x1.d5b495f6349e6p-56},
    {0x1.0c6145b5b43dap-1, 0x1.974fa13b5404fp-58},
    {0x1.1255d9bfbd2a9p-1, -0x1.2bdaee1c0ee35p-58},
    {0x1.1835a88be7c13p-1, 0x1.c621cec00c301p-55},
    {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},
    {0x1.23b71e2cc9e6ap-1, 0x1.c421c9f38224ep-57},
    {0x1.2958e59308e31p-1, -0x1.09e73b0c6c087p-56},
    {0x1.2ee628406cbcap-1, 0x1.c5d5e9ff0cf8dp-55},
    {0x1.345f01cce37bbp-1, 0x1.1021137c71102p-55},
    {0x1.39c391cd4171ap-1, -0x1.2304331d8bf46p-55},
    {0x1.3f13fb89e96f4p-1, 0x1.ecf8b492644f0p-56},
    {0x1.445065b795b56p-1, -0x1.f76d0163f79c8p-56},
    {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},
    {0x1.4e8de5bb6ec04p-1, 0x1.4a33dbeb3796cp-55},
    {0x1.538f57b89061fp-1, -0x1.1bb74abda520cp-55},
    {0x1.587d81f732fbbp-1, -0x1.5e5c9d8c5a950p-56},
    {0x1.5d58987169b18p-1, 0x1.0028e4bc5e7cap-57},
    {0x1.6220d115d7b8ep-1, -0x1.2b785350ee8c1p-57},
    {0x1.66d663923e087p-1, -0x1.6ea6febe8bbbap-56},
    {0x1.6b798920b3d99p-1, -0x1.a80386188c50ep-55},
    {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},
    {0x1.748978fba8e0fp-1, 0x1.7b2a6165884a1p-59},
    {0x1.78f6bbd5d315ep-1, 0x1.406a089803740p-55},
    {0x1.7d528289fa093p-1, 0x1.560821e2f3aa9p-55},
    {0x1.819d0b7158a4dp-1, -0x1.bf76229d3b917p-56},
    {0x1.85d69576cc2c5p-1, 0x1.6b66e7fc8b8c3p-57},
    {0x1.89ff5ff57f1f8p-1, -0x1.55b9a5e177a1bp-55},
    {0x1.8e17aa99cc05ep-1, -0x1.ec182ab042f61p-56},
    {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55},
    };
    cra_dd r;
    r.hi = T[k][0];
    r.lo = T[k][1];
    return r;
}

// atan(q) for a double-double q in [0, 1]
CRA_HD inline cra_dd cra_atan01(cra_dd q) {
    int k = (int)(q.hi * 64.0 + 0.5);
    if (k > 64) k = 64;
    cra_dd t = q;
    if (k > 0) {
        const double c = k / 64.0;  // exact
        cra_dd cd;
        cd.hi = -c;
        cd.lo = 0.0;
        const cra_dd num = cra_add(q, cd);
        cra_dd one;
        one.hi = 1.0;
        one.lo = 0.0;
        const cra_dd den = cra_add(one, cra_mul_d(q, c));
        t = cra_div(num, den);
    }
    // atan(t) = t - t^3/3 + t^5/5 - ...  (|t| <= 2^-7): t and t^3/3 in double-double, the rest in double
    const cra_dd t2 = cra_mul(t, t);
    const double z = t2.hi;
    const double tail = t.hi * z * z * (0x1.999999999999ap-3 + z * (-0x1.2492492492492p-3 + z * (0x1.c71c71c71c71cp-4 +
                        z * (-0x1.745d1745d1746p-4 + z * (0x1.3b13b13b13b14p-4 + z * -0x1.1111111111111p-4)))));
    cra_dd third;
    third.hi = 0x1.5555555555555p-2;
    third.lo = 0x1.5555555555555p-56;
    const cra_dd t3 = cra_mul(cra_mul(t, t2), third);
    cra_dd s = cra_add(t, cra_neg(t3));
    cra_dd td;
    td.hi = tail;
    td.lo = 0.0;
    s = cra_add(s, td);
    return cra_add(cra_atan_table(k), s);
}

CRA_HD inline double cr_atan2(double y, double x) {
    const double kPiHi = 0x1.921fb54442d18p+1, kPiLo = 0x1.1a62633145c07p-53;
    if (x != x || y != y) return x + y;
    const double ax = fabs(x), ay = fabs(y);
    if (ay == 0.0) return (copysign(1.0, x) > 0) ? y : copysign(kPiHi, y);  // +-0 or +-pi
    if (ax == 0.0) return copysign(kPiHi * 0.5, y);
    if (isinf(ax) || isinf(ay)) return atan2(y, x);  // libm handles the infinities exactly
    // scale away extreme exponents (not needed for metres-scale inputs, kept for safety)
    const bool swap = ay > ax;
    const double mn = swap ? ax : ay, mx = swap ? ay : ax;
    cra_dd q;
    q.hi = mn / mx;
    q.lo = fma(-q.hi, mx, mn) / mx;
    if (q.hi < 0x1p-500) return atan2(y, x);  // tiny ratios: the libm result is exact enough (atan(q) ~ q)
    cra_dd a = cra_atan01(q);
    cra_dd half_pi, pi;
    half_pi.hi = kPiHi * 0.5;
    half_pi.lo = kPiLo * 0.5;
    pi.hi = kPiHi;
    pi.lo = kPiLo;
    if (swap) a = cra_add(half_pi, cra_neg(a));  // atan(ay/ax) = pi/2 - atan(ax/ay)
    if (x < 0) a = cra_add(pi, cra_neg(a));
    const double r = a.hi + a.lo;
    return y < 0 ? -r : r;
}
