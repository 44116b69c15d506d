// Per-cell arithmetic of one forward-Euler sweep, in the reference's operation order.
//
// Each cell's update splits into
//   phase A  -- everything that depends only on the cell's own reconstruction stencil
//               (itself + its face neighbours): limited face values, face viscosities and
//               pressures, the complete left-hand-side sum and unknown coefficient of every
//               equation, and the scaled products that neighbouring cells difference;
//   phase B  -- the right-hand-side sums, which difference phase-A products of the cell and
//               of its face neighbours, followed by the solve.
// A kernel evaluates phase A once per cell and hands the products to the neighbours, so
// nothing is reconstructed twice.
//
// Reference functions collapsed here (paths into the reference tree):
//   propLimitedSlope_adj, superbee        src/SolutionDomain.hs:644-659, 869-897
//   prop_adj (Mew, Pressure)              src/SolutionDomain.hs:569-574
//   approximateDerivative_adj             src/SolutionDomain.hs:506-538
//   integSurface_adj, multTerm            src/CalculusUtils.hs:58-91
//   solveUnknown                          src/CalculusUtils.hs:36-52
//   continuity .. energy                  src/Main.hs:89-175
//
// Identities used (all exact in IEEE binary64, so results keep the reference's bits):
//   x*1 = x;  x*(-1) = -x;  (-a)*b = -(a*b);  (-a)-(-b) = b-a;  s + (-y) = s - y;
//   a sum that starts at +0 never becomes -0, so adding a +-0 term is a no-op.
// t**1.5 is evaluated as t*sqrt(t).
#pragma once

#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define CFD_HD __host__ __device__ __forceinline__
#else
#define CFD_HD inline
#endif

namespace cfd {

struct SweepConsts {
    double Ae, An, At;      // face areas: East/West, North/South, Top/Bottom
    double vol, dt;
    double hx, hy, hz;      // average [h, h_nb]          (same-direction interval)
    double Hx, Hy, Hz;      // 2 * average [h, h_n1, h_n2] (cross-direction interval)
    double rhx, rhy, rhz;   // correctly rounded reciprocals of the six intervals (for div_by)
    double rHx, rHy, rHz;
    double lhx, lhy, lhz;   // and their low parts: RN(1/h - rh), so that rh + lh is 1/h to about 106 bits
    double lHx, lHy, lHz;
    double lambda, suth, R, cv, k, dyef;
    double m23;             // -(2/3)
    double c83h;            // (8/3)/2
    double c83n;            // (8/3)*(-1)
};

// Low part of the reciprocal of h, given rh = RN(1/h): (1 - rh h) / h, whose numerator fma forms exactly.
inline double recip_low(double h, double rh) { return fma(-rh, h, 1.0) / h; }

// Host-side construction in the reference's fold orders (avMap/slMap, average).
inline SweepConsts make_consts(double dx, double dy, double dz, double dt, double lambda,
                               double suth, double R, double cv, double k, double dyef) {
    SweepConsts c;
    c.Ae = (1.0 * dy) * dz;
    c.An = (1.0 * dx) * dz;
    c.At = (1.0 * dx) * dy;
    c.vol = ((1.0 * dx) * dy) * dz;
    c.dt = dt;
    const double len[3] = {dx, dy, dz};
    double hs[3], hc[3];
    for (int d = 0; d < 3; ++d) {
        hs[d] = (0.0 + len[d] / 2.0) + len[d] / 2.0;
        hc[d] = 2.0 * (((0.0 + len[d] / 3.0) + len[d] / 3.0) + len[d] / 3.0);
    }
    c.hx = hs[0]; c.hy = hs[1]; c.hz = hs[2];
    c.Hx = hc[0]; c.Hy = hc[1]; c.Hz = hc[2];
    c.rhx = 1.0 / c.hx; c.rhy = 1.0 / c.hy; c.rhz = 1.0 / c.hz;
    c.rHx = 1.0 / c.Hx; c.rHy = 1.0 / c.Hy; c.rHz = 1.0 / c.Hz;
    c.lhx = recip_low(c.hx, c.rhx); c.lhy = recip_low(c.hy, c.rhy); c.lhz = recip_low(c.hz, c.rhz);
    c.lHx = recip_low(c.Hx, c.rHx); c.lHy = recip_low(c.Hy, c.rHy); c.lHz = recip_low(c.Hz, c.rHz);
    c.lambda = lambda; c.suth = suth; c.R = R; c.cv = cv; c.k = k; c.dyef = dyef;
    c.m23 = -(2.0 / 3.0);
    c.c83h = (8.0 / 3.0) / 2.0;
    c.c83n = (8.0 / 3.0) * (-1.0);
    return c;
}

// Upper 32 bits of a double (sign, exponent, top of the mantissa).
CFD_HD int hi_word(double x) {
    long long b;
    memcpy(&b, &x, sizeof b);
    return (int)(b >> 32);
}

CFD_HD double from_bits(long long b) {
    double x;
    memcpy(&x, &b, sizeof x);
    return x;
}

CFD_HD long long to_bits(double x) {
    long long b;
    memcpy(&b, &x, sizeof b);
    return b;
}

// 0.5 * superbee(a, b), where superbee a b is zero unless a and b are both > 0 or both < 0 and
// otherwise sign * min(max(|a|,|b|), min(2|a|,2|b|)).  Branch-free and select-free:
//   sb = b with its sign flipped when a is negative, so that "same sign" reads sb > 0;
//   0.5*min(hi, 2 lo) == min(0.5 hi, lo) exactly (scaling by two is exact), with hi = max(|a|, sb)
//   and lo = min(|a|, sb); for sb > 0 that is the limiter's magnitude;
//   for sb <= 0 (opposite signs, or b == 0) lo = sb <= 0 makes the minimum <= 0, and for a == 0
//   lo <= 0 as well: the outer max(0, .) returns the required zero;
//   the sign of a is then OR-ed onto the non-negative magnitude.
// A zero result may carry a sign bit; it is only ever added to / subtracted from the cell value,
// where +0 and -0 act alike.  (For NaN inputs the reference's getSign calls NaN negative; NaN
// fields are outside what this code reproduces.)
CFD_HD double half_superbee_plain(double a, double b) {
    const long long sign = to_bits(a) & (long long)0x8000000000000000ull;
    const double sb = from_bits(to_bits(b) ^ sign);
    const double aa = fabs(a);
    const double m = fmax(0.0, fmin(0.5 * fmax(aa, sb), fmin(aa, sb)));
    return from_bits(to_bits(m) | sign);
}

// The same value with the four maxima / minima written as single instructions.  Given fmax / fmin the compiler
// first quiets a possible signalling NaN in sb (which comes out of an integer operation) with a v_max_f64 sb, sb
// of its own: one instruction in nine.  NaN fields are outside what this code reproduces, and for every other
// input the two forms return the same bits.  Where a kernel at the register limit spills with this form it keeps
// the plain one (x_half_slopes2<true>).
CFD_HD double half_superbee(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const long long sign = to_bits(a) & (long long)0x8000000000000000ull;
    const double sb = from_bits(to_bits(b) ^ sign);
    double hi, lo, m;
    asm("v_max_f64 %0, |%1|, %2" : "=v"(hi) : "v"(a), "v"(sb));
    asm("v_min_f64 %0, |%1|, %2" : "=v"(lo) : "v"(a), "v"(sb));
    hi *= 0.5;
    asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(hi), "v"(lo));
    asm("v_max_f64 %0, %1, 0" : "=v"(m) : "v"(m));
    return from_bits(to_bits(m) | sign);
#else
    return half_superbee_plain(a, b);
#endif
}

CFD_HD double superbee(double a, double b) { return 2.0 * half_superbee(a, b); }

// 0.5 * superbee(up - vc, vc - lo): the half-slope added to / subtracted from vc.
CFD_HD double half_slope(double up, double vc, double lo) {
    return half_superbee(up - vc, vc - lo);
}
CFD_HD double half_slope_plain(double up, double vc, double lo) {
    return half_superbee_plain(up - vc, vc - lo);
}

// x / h, correctly rounded, for a divisor h that is known in advance, with rh = RN(1/h) and rl = RN(1/h - rh).
//   q0 = x rh + RN(x rl)   one rounding of a product taken with a 106-bit reciprocal: within half an ulp (and a
//                          2^-50 ulp) of x/h, i.e. a faithful quotient;
//   r  = x - q0 h          exact in one fused multiply-add, because q0 is faithful;
//   q  = q0 + r rh         Markstein's theorem: a faithful quotient corrected once with the exact residual and the
//                          correctly rounded reciprocal is the correctly rounded quotient --
// the same bits as the IEEE division the reference performs, in 4 multiply-add operations instead of the
// ~12-instruction division sequence.  Holds for finite x with |x| >= 2^-970 or x == 0 (below that the residual
// underflows and is no longer exact); a zero quotient may lose its sign.  The numerators it is used on are
// differences of viscous / conductive products of velocity and temperature scale; the Dye terms, whose
// numerators do decay into that range, use the plain division.
CFD_HD double div_by(double x, double h, double rh, double rl) {
    const double q0 = fma(x, rh, x * rl);
    return fma(fma(-q0, h, x), rh, q0);
}

// n / d and sqrt(x), correctly rounded, for operands in the middle of the exponent range: the instruction sequences
// the compiler emits for an IEEE division / square root (reciprocal or reciprocal-square-root estimate, Newton steps
// with fused residuals, one final correction) WITHOUT the operand scaling and the special-value fix-up that surround
// them (v_div_scale x2, v_div_fixup; v_cmp / v_ldexp x2 / v_cmp_class / v_cndmask x2) -- 8 instead of 11 and 10
// instead of 18 instructions.  Identical bits to `/` and sqrt wherever that scaling would not have engaged:
//   div_normal:  d and n / d finite, non-zero and normal, no exponent of d, n or n / d within ~50 of the range ends;
//   sqrt_normal: x >= 2^-767, finite.
// Used where the operands are Temperature, Density, viscosity and the cell volume -- positive physical quantities --
// never on velocity components or Dye, which pass through zero and the subnormal range.
CFD_HD double div_normal(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
#else
    return n / d;
#endif
}
CFD_HD double sqrt_normal(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
#else
    return sqrt(x);
#endif
}

CFD_HD double mu_of(const SweepConsts& k, double t) {
    return div_normal(k.lambda * (t * sqrt_normal(t)), t + k.suth);
}

// c ? x : y on doubles.  On the device the condition is materialised as a lane mask in a scalar
// register pair and applied with the three-operand form of the conditional move: measured on
// gfx950, the two-operand form that reads the mask from VCC issues about three times slower.
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long lane_mask;
CFD_HD lane_mask mask_of(bool c) { return __builtin_amdgcn_ballot_w64(c); }
CFD_HD double select(lane_mask m, double x, double y) {
    const unsigned long long xb = (unsigned long long)to_bits(x), yb = (unsigned long long)to_bits(y);
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)yb), "v"((unsigned)xb), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((unsigned)(yb >> 32)), "v"((unsigned)(xb >> 32)), "s"(m));
    return from_bits((long long)(((unsigned long long)hi << 32) | lo));
}
CFD_HD unsigned select_u32(lane_mask m, unsigned x, unsigned y) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(y), "v"(x), "s"(m));
    return r;
}
#else
typedef bool lane_mask;
CFD_HD lane_mask mask_of(bool c) { return c; }
CFD_HD double select(lane_mask m, double x, double y) { return m ? x : y; }
CFD_HD unsigned select_u32(lane_mask m, unsigned x, unsigned y) { return m ? x : y; }
#endif

// The two ways a kernel can have its conditional moves made.  SelMask: the helpers above.  SelPlain:
// ordinary ?: left to the compiler.  Which is faster is a property of the whole kernel's register
// allocation and schedule, so it is chosen per kernel by measurement (same GPU, alternating runs):
// the single-sweep kernels run 4 % faster with SelMask, the two-sweeps kernel 5 % faster with SelPlain.
struct SelMask {
    typedef lane_mask mask;
    CFD_HD static mask of(bool c) { return mask_of(c); }
    CFD_HD static double pick(mask m, double x, double y) { return select(m, x, y); }
    CFD_HD static unsigned pick32(mask m, unsigned x, unsigned y) { return select_u32(m, x, y); }
};
struct SelPlain {
    typedef bool mask;
    CFD_HD static mask of(bool c) { return c; }
    CFD_HD static double pick(mask m, double x, double y) { return m ? x : y; }
    CFD_HD static unsigned pick32(mask m, unsigned x, unsigned y) { return m ? x : y; }
};

// F * (1/phi), with the reference's NaN -> 1 substitution.
// The same for an unknown that is a positive physical quantity (Density, Temperature): see div_normal.
CFD_HD double unknown_coeff_normal(double F, double phi) {
    const double u0 = F * div_normal(1.0, phi);
    return select(mask_of(u0 != u0), 1.0, u0);
}

CFD_HD double unknown_coeff(double F, double phi) {
    const double u0 = F * (1.0 / phi);
    return select(mask_of(u0 != u0), 1.0, u0);
}

// ------------------------------------------------------------------------------------------
// 2-D case: nz == 1 and W identically zero.  Top/Bottom neighbours are the cell itself, every
// z-flux carries the factor W = 0 and every z-difference is of identical values, so those
// terms are +-0 and drop out; W's own equation evaluates to exactly +0.  The one z-term that
// survives is the +X, -X pair of d/dz(-d(mu U)/dx) in the energy equation.
// Field order of the 2-D arrays: U, V, Density, Temperature, Dye.
// ------------------------------------------------------------------------------------------
enum { F2_U = 0, F2_V, F2_RHO, F2_T, F2_DYE, NF2 };
enum { E2_RHO = 0, E2_U, E2_V, E2_T, E2_DYE };     // equation slots in lhs[] / aU[]

// Scaled centre products differenced along x (by this cell and its East/West neighbours).
struct ShareX2 {
    double aX;      // ((mu U) Ae) dt
    double aX23;    // (((mu U) m23) Ae) dt
    double bX;      // ((mu V) Ae) dt
    double tX;      // ((T k) Ae) dt
    double a2X83;   // ((((mu U) mu) U) c83h) Ae) dt
    double aX83;    // (((mu U) c83n) Ae) dt
    double b2Xh;    // ((((mu V) mu) V) 0.5) Ae) dt
    double gX;      // (((mu Dye) dyef) Ae) dt
    double aZ;      // ((mu U) At) dt
    // North/South face products of this cell, differenced by its East/West neighbours
    double bN, bS;      // ((mu_f V_f) An) dt
    double aN23, aS23;  // (((mu_f U_f) m23) An) dt
    double cN, cS;      // (((mu_f U_f) V_f) An) dt
    double aN, aS;      // ((mu_f U_f) An) dt
    double vN, vS;      // (V_f An) dt
};

// Scaled centre products differenced along y (by this cell and its North/South neighbours).
struct ShareY2 {
    double aY;      // ((mu U) An) dt
    double bY;      // ((mu V) An) dt
    double bY23;    // (((mu V) m23) An) dt
    double tY;      // ((T k) An) dt
    double b2Y83;   // ((((mu V) mu) V) c83h) An) dt
    double bY83;    // (((mu V) c83n) An) dt
    double a2Yh;    // ((((mu U) mu) U) 0.5) An) dt
    double gY;      // (((mu Dye) dyef) An) dt
    // East/West face products of this cell, differenced by its North/South neighbours
    double bE23, bW23;  // (((mu_f V_f) m23) Ae) dt
    double aE, aW;      // ((mu_f U_f) Ae) dt
};

struct PhaseA2 {
    double lhs[5];      // complete left-hand-side constant sums
    double aU[5];       // unknown coefficients (already times the cell volume)
    double own[3];      // the cell's stored U, V, Dye (kept where the cell is inactive)
    ShareX2 sx;
    ShareY2 sy;
};

// x half-slopes of the five stored values: 0.5*superbee(e - c, c - w).  Split out of phase A so that
// a kernel can retire the West row's registers (and reuse them for the next load) early.
// PLAIN: the limiter's portable form (the two-sweeps kernel, which runs close to the register limit, spills with
// the single-instruction form here; its y slopes take that form without harm).
template <bool PLAIN = false>
CFD_HD void x_half_slopes2(const double c[NF2], const double e[NF2], const double w[NF2], double hx[NF2]) {
    for (int q = 0; q < NF2; ++q) hx[q] = PLAIN ? half_slope_plain(e[q], c[q], w[q]) : half_slope(e[q], c[q], w[q]);
}

// c,n,s: the five stored values (F2_* order) of the cell and of its clamped North/South neighbours;
// hx: its x half-slopes.
CFD_HD void phaseA2(const SweepConsts& k, const double c[NF2], const double hx[NF2],
                    const double n[NF2], const double s[NF2], PhaseA2& o) {
    // limited face values: vc +- 0.5*superbee(up - vc, vc - lo)
    double fE[NF2], fW[NF2], fN[NF2], fS[NF2];
    for (int q = 0; q < NF2; ++q) {
        const double hy = half_slope(n[q], c[q], s[q]);
        fE[q] = c[q] + hx[q]; fW[q] = c[q] - hx[q];
        fN[q] = c[q] + hy; fS[q] = c[q] - hy;
    }
    const double muC = mu_of(k, c[F2_T]);
    const double muE = mu_of(k, fE[F2_T]), muW = mu_of(k, fW[F2_T]);
    const double muN = mu_of(k, fN[F2_T]), muS = mu_of(k, fS[F2_T]);
    const double pE = (k.R * fE[F2_T]) * fE[F2_RHO], pW = (k.R * fW[F2_T]) * fW[F2_RHO];
    const double pN = (k.R * fN[F2_T]) * fN[F2_RHO], pS = (k.R * fS[F2_T]) * fS[F2_RHO];

    const double rE = fE[F2_RHO], rW = fW[F2_RHO], rN = fN[F2_RHO], rS = fS[F2_RHO];
    const double uE = fE[F2_U], uW = fW[F2_U], uN = fN[F2_U], uS = fS[F2_U];
    const double vE = fE[F2_V], vW = fW[F2_V], vN = fN[F2_V], vS = fS[F2_V];
    const double tE = fE[F2_T], tW = fW[F2_T], tN = fN[F2_T], tS = fS[F2_T];
    const double dE = fE[F2_DYE], dW = fW[F2_DYE], dN = fN[F2_DYE], dS = fS[F2_DYE];
    const double Ae = k.Ae, An = k.An, dt = k.dt;

    // mass fluxes shared by several equations: (rho_f u_f) on E/W, (rho_f v_f) on N/S
    const double mE = rE * uE, mW = rW * uW, mN = rN * vN, mS = rS * vS;

    o.own[0] = c[F2_U]; o.own[1] = c[F2_V]; o.own[2] = c[F2_DYE];
    double F, l;
    // ---- continuity: d(rho)/dt + div(rho u)
    F = c[F2_RHO];
    o.aU[E2_RHO] = unknown_coeff_normal(F, c[F2_RHO]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    l = l + (mE * Ae) * dt;  l = l - (mW * Ae) * dt;
    l = l + (mN * An) * dt;  l = l - (mS * An) * dt;
    o.lhs[E2_RHO] = l;
    // ---- U momentum: d(rho U)/dt + div(rho U u) + dP/dx
    F = c[F2_RHO] * c[F2_U];
    o.aU[E2_U] = unknown_coeff(F, c[F2_U]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    l = l + ((mE * uE) * Ae) * dt;          l = l - ((mW * uW) * Ae) * dt;
    l = l + (((rN * uN) * vN) * An) * dt;   l = l - (((rS * uS) * vS) * An) * dt;
    l = l + (pE * Ae) * dt;                 l = l - (pW * Ae) * dt;
    o.lhs[E2_U] = l;
    // ---- V momentum: d(rho V)/dt + div(rho V u) + dP/dy
    F = c[F2_RHO] * c[F2_V];
    o.aU[E2_V] = unknown_coeff(F, c[F2_V]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    l = l + (((rE * vE) * uE) * Ae) * dt;   l = l - (((rW * vW) * uW) * Ae) * dt;
    l = l + ((mN * vN) * An) * dt;          l = l - ((mS * vS) * An) * dt;
    l = l + (pN * An) * dt;                 l = l - (pS * An) * dt;
    o.lhs[E2_V] = l;
    // ---- energy: d(rho T cv)/dt + div(rho T u cv) + div(P u)
    F = (c[F2_RHO] * c[F2_T]) * k.cv;
    o.aU[E2_T] = unknown_coeff_normal(F, c[F2_T]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    l = l + ((((rE * tE) * uE) * k.cv) * Ae) * dt;  l = l - ((((rW * tW) * uW) * k.cv) * Ae) * dt;
    l = l + ((((rN * tN) * vN) * k.cv) * An) * dt;  l = l - ((((rS * tS) * vS) * k.cv) * An) * dt;
    l = l + ((pE * uE) * Ae) * dt;                  l = l - ((pW * uW) * Ae) * dt;
    l = l + ((pN * vN) * An) * dt;                  l = l - ((pS * vS) * An) * dt;
    o.lhs[E2_T] = l;
    // ---- dye: d(rho Dye)/dt + div(rho Dye u)
    F = c[F2_RHO] * c[F2_DYE];
    o.aU[E2_DYE] = unknown_coeff(F, c[F2_DYE]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    l = l + (((rE * dE) * uE) * Ae) * dt;   l = l - (((rW * dW) * uW) * Ae) * dt;
    l = l + (((rN * dN) * vN) * An) * dt;   l = l - (((rS * dS) * vS) * An) * dt;
    o.lhs[E2_DYE] = l;

    // ---- products the right-hand sides difference
    const double a = muC * c[F2_U], b = muC * c[F2_V], g = muC * c[F2_DYE];
    const double a2 = (a * muC) * c[F2_U], b2 = (b * muC) * c[F2_V];
    const double tk = c[F2_T] * k.k, gd = g * k.dyef;
    const double a23 = a * k.m23, b23 = b * k.m23;
    ShareX2& X = o.sx;
    X.aX = (a * Ae) * dt;           X.aX23 = (a23 * Ae) * dt;
    X.bX = (b * Ae) * dt;           X.tX = (tk * Ae) * dt;
    X.a2X83 = ((a2 * k.c83h) * Ae) * dt;
    X.aX83 = ((a * k.c83n) * Ae) * dt;
    X.b2Xh = ((b2 * 0.5) * Ae) * dt;
    X.gX = (gd * Ae) * dt;
    X.aZ = (a * k.At) * dt;
    const double afN = muN * uN, afS = muS * uS, bfN = muN * vN, bfS = muS * vS;
    X.bN = (bfN * An) * dt;                 X.bS = (bfS * An) * dt;
    X.aN23 = ((afN * k.m23) * An) * dt;     X.aS23 = ((afS * k.m23) * An) * dt;
    X.cN = ((afN * vN) * An) * dt;          X.cS = ((afS * vS) * An) * dt;
    X.aN = (afN * An) * dt;                 X.aS = (afS * An) * dt;
    X.vN = (vN * An) * dt;                  X.vS = (vS * An) * dt;
    ShareY2& Y = o.sy;
    Y.aY = (a * An) * dt;           Y.bY = (b * An) * dt;
    Y.bY23 = (b23 * An) * dt;       Y.tY = (tk * An) * dt;
    Y.b2Y83 = ((b2 * k.c83h) * An) * dt;
    Y.bY83 = ((b * k.c83n) * An) * dt;
    Y.a2Yh = ((a2 * 0.5) * An) * dt;
    Y.gY = (gd * An) * dt;
    const double afE = muE * uE, afW = muW * uW, bfE = muE * vE, bfW = muW * vW;
    Y.bE23 = ((bfE * k.m23) * Ae) * dt;     Y.bW23 = ((bfW * k.m23) * Ae) * dt;
    Y.aE = (afE * Ae) * dt;                 Y.aW = (afW * Ae) * dt;
}

// Right-hand sides and solve.  C = this cell's phase A; XE/XW = ShareX2 of the East/West
// neighbours, YN/YS = ShareY2 of the North/South neighbours (a clamped neighbour is the cell
// itself, which makes its differences vanish as in the reference).
// out[] in E2_* order.
CFD_HD void phaseB2(const SweepConsts& k, const PhaseA2& C, const ShareX2& XE, const ShareX2& XW,
                    const ShareY2& YN, const ShareY2& YS, double out[5]) {
    const ShareX2& X = C.sx;
    const ShareY2& Y = C.sy;
    const double hx = k.hx, hy = k.hy, Hx = k.Hx, Hy = k.Hy;
    double r;
    // ---- continuity: right-hand side is the single constant 0
    r = 0.0 + 0.0;
    out[E2_RHO] = div_normal(r - C.lhs[E2_RHO], C.aU[E2_RHO]);

    // ---- U momentum
    {
        const double d1u = div_by(XE.aX - X.aX, hx, k.rhx, k.lhx), d1l = div_by(XW.aX - X.aX, hx, k.rhx, k.lhx);   // d/dx(d(mu U)/dx)
        r = 0.0 + d1u;  r = r + d1l;
        r = r + div_by(YN.aY - Y.aY, hy, k.rhy, k.lhy);   r = r + div_by(YS.aY - Y.aY, hy, k.rhy, k.lhy);           // d/dy(d(mu U)/dy)
        r = r + d1u;    r = r + d1l;                                           // d/dx(d(mu U)/dx) again
        r = r + div_by(XE.bN - XW.bN, Hx, k.rHx, k.lHx);  r = r + div_by(XW.bS - XE.bS, Hx, k.rHx, k.lHx);           // d/dy(d(mu V)/dx)
        r = r + div_by(XE.aX23 - X.aX23, hx, k.rhx, k.lhx);  r = r + div_by(XW.aX23 - X.aX23, hx, k.rhx, k.lhx);     // -2/3 d/dx(d(mu U)/dx)
        r = r + div_by(YN.bE23 - YS.bE23, Hy, k.rHy, k.lHy); r = r + div_by(YS.bW23 - YN.bW23, Hy, k.rHy, k.lHy);    // -2/3 d/dx(d(mu V)/dy)
        out[E2_U] = (r - C.lhs[E2_U]) / C.aU[E2_U];
    }
    // ---- V momentum
    {
        const double d2u = div_by(YN.bY - Y.bY, hy, k.rhy, k.lhy), d2l = div_by(YS.bY - Y.bY, hy, k.rhy, k.lhy);   // d/dy(d(mu V)/dy)
        r = 0.0 + div_by(XE.bX - X.bX, hx, k.rhx, k.lhx);  r = r + div_by(XW.bX - X.bX, hx, k.rhx, k.lhx);           // d/dx(d(mu V)/dx)
        r = r + d2u;    r = r + d2l;
        r = r + div_by(YN.aE - YS.aE, Hy, k.rHy, k.lHy);   r = r + div_by(YS.aW - YN.aW, Hy, k.rHy, k.lHy);          // d/dx(d(mu U)/dy)
        r = r + d2u;    r = r + d2l;                                           // d/dy(d(mu V)/dy) again
        r = r + div_by(XE.aN23 - XW.aN23, Hx, k.rHx, k.lHx);  r = r + div_by(XW.aS23 - XE.aS23, Hx, k.rHx, k.lHx);   // -2/3 d/dy(d(mu U)/dx)
        r = r + div_by(YN.bY23 - Y.bY23, hy, k.rhy, k.lhy);   r = r + div_by(YS.bY23 - Y.bY23, hy, k.rhy, k.lhy);    // -2/3 d/dy(d(mu V)/dy)
        out[E2_V] = (r - C.lhs[E2_V]) / C.aU[E2_V];
    }
    // ---- energy
    {
        r = 0.0 + div_by(XE.tX - X.tX, hx, k.rhx, k.lhx);  r = r + div_by(XW.tX - X.tX, hx, k.rhx, k.lhx);           // k lap(T), x
        r = r + div_by(YN.tY - Y.tY, hy, k.rhy, k.lhy);    r = r + div_by(YS.tY - Y.tY, hy, k.rhy, k.lhy);           // k lap(T), y
        // squareDerivative [Mew,U] (8/3) X
        r = r + div_by(XE.a2X83 - X.a2X83, hx, k.rhx, k.lhx);  r = r + div_by(XW.a2X83 - X.a2X83, hx, k.rhx, k.lhx);
        r = r + div_by(XE.aX83 - X.aX83, hx, k.rhx, k.lhx);    r = r + div_by(XW.aX83 - X.aX83, hx, k.rhx, k.lhx);
        // squareDerivative [Mew,V] (8/3) Y
        r = r + div_by(YN.b2Y83 - Y.b2Y83, hy, k.rhy, k.lhy);  r = r + div_by(YS.b2Y83 - Y.b2Y83, hy, k.rhy, k.lhy);
        r = r + div_by(YN.bY83 - Y.bY83, hy, k.rhy, k.lhy);    r = r + div_by(YS.bY83 - Y.bY83, hy, k.rhy, k.lhy);
        // squareDerivative [Mew,U] 1 Y   (second term has factor -1: differences flip)
        r = r + div_by(YN.a2Yh - Y.a2Yh, hy, k.rhy, k.lhy);    r = r + div_by(YS.a2Yh - Y.a2Yh, hy, k.rhy, k.lhy);
        r = r + div_by(Y.aY - YN.aY, hy, k.rhy, k.lhy);        r = r + div_by(Y.aY - YS.aY, hy, k.rhy, k.lhy);
        // squareDerivative [Mew,V] 1 X
        r = r + div_by(XE.b2Xh - X.b2Xh, hx, k.rhx, k.lhx);    r = r + div_by(XW.b2Xh - X.b2Xh, hx, k.rhx, k.lhx);
        r = r + div_by(X.bX - XE.bX, hx, k.rhx, k.lhx);        r = r + div_by(X.bX - XW.bX, hx, k.rhx, k.lhx);
        // pairedMultipliedDerivatives [Mew,U] [V] X Y
        r = r + div_by(XE.cN - XW.cN, Hx, k.rHx, k.lHx);       r = r + div_by(XW.cS - XE.cS, Hx, k.rHx, k.lHx);
        r = r + div_by(XW.aN - XE.aN, Hx, k.rHx, k.lHx);       r = r + div_by(XE.aS - XW.aS, Hx, k.rHx, k.lHx);
        r = r + div_by(XW.vN - XE.vN, Hx, k.rHx, k.lHx);       r = r + div_by(XE.vS - XW.vS, Hx, k.rHx, k.lHx);
        // pairedMultipliedDerivatives [Mew,U] [W] X Z: the surviving +X, -X pair
        const double zl = div_by(XW.aZ - XE.aZ, Hx, k.rHx, k.lHx);
        r = r + zl;                         r = r - zl;
        out[E2_T] = div_normal(r - C.lhs[E2_T], C.aU[E2_T]);
    }
    // ---- dye
    {
        // plain divisions: Dye decays towards zero away from its source and these numerators reach
        // the subnormal range, where div_by is not exact
        r = 0.0 + (XE.gX - X.gX) / hx;  r = r + (XW.gX - X.gX) / hx;
        r = r + (YN.gY - Y.gY) / hy;    r = r + (YS.gY - Y.gY) / hy;
        out[E2_DYE] = (r - C.lhs[E2_DYE]) / C.aU[E2_DYE];
    }
}


// ------------------------------------------------------------------------------------------
// 2-D case, re-associated arithmetic ("fast").  Same equations, same limiter, same NaN -> 1 rule,
// but each equation is solved in increment form
//     phi_new = phi - (div(flux) - rhs/vol) / coefficient
// with the geometric factors folded into per-direction constants, every second difference taken
// once on a per-cell combination of products, divisions by the constant intervals replaced by
// multiplications, and mu and 1/rho evaluated by Newton iterations on the hardware estimates.
// Mathematically identical to the strict form; rounding differs at the level of the strict
// form's own rounding noise (<= 1e-13 of the field scale per sweep).
// ------------------------------------------------------------------------------------------
struct FastConsts {
    double rcx, rcy, rcz;   // dt/dx, dt/dy, dt/dz           (flux divergence per unit volume)
    double rcx2, rcy2, rcz2;    // 2 dt/dx, ...             (streaming forms: face differences come out halved)
    double kxx, kyy, kzz;   // dt/dx^2, dt/dy^2, dt/dz^2
    double kxy, kxz, kyz;   // dt/(2 dx dy), dt/(2 dx dz), dt/(2 dy dz)
    double kxxm, kyym, kzzm;    // (2 + m23) * kxx, ...
    double kxym, kxzm, kyzm;    // m23 * kxy, ...
    double lambda, suth, R, cv, rcv, k, dyef;
    double c83h, c83n;
    double cp;              // cv + R (streaming forms: convective + pressure-work energy flux in one product)
    double il, sl;          // 1 / lambda, suth / lambda (mu_stream)
    // streaming form (2-D): centre coefficients and Dye constants with the Dye factor folded in
    double cUa, cVb;        // 2 (kxxm + kyy), 2 (kxx + kyym)
    double kxx2, kyy2;      // 2 kxx, 2 kyy
    double kxxd, kyyd, cDg; // dyef kxx, dyef kyy, 2 dyef (kxx + kyy)
    // streaming form (3-D)
    double kzz2, kzzd;      // 2 kzz, dyef kzz
    double cUa3, cVb3, cWc3, cDg3;  // 2 (kxxm + kyy + kzz), 2 (kxx + kyym + kzz), 2 (kxx + kyy + kzzm), 2 dyef (kxx + kyy + kzz)
};

inline FastConsts make_fast_consts(const SweepConsts& c, double dx, double dy, double dz = 1.0) {
    FastConsts f;
    f.rcx = c.dt / dx;  f.rcy = c.dt / dy;  f.rcz = c.dt / dz;
    f.rcx2 = 2.0 * f.rcx;  f.rcy2 = 2.0 * f.rcy;  f.rcz2 = 2.0 * f.rcz;
    f.kxx = c.dt / (dx * dx);  f.kyy = c.dt / (dy * dy);  f.kzz = c.dt / (dz * dz);
    f.kxy = c.dt / (2.0 * dx * dy);  f.kxz = c.dt / (2.0 * dx * dz);  f.kyz = c.dt / (2.0 * dy * dz);
    f.kxxm = (2.0 + c.m23) * f.kxx;  f.kyym = (2.0 + c.m23) * f.kyy;  f.kzzm = (2.0 + c.m23) * f.kzz;
    f.kxym = c.m23 * f.kxy;  f.kxzm = c.m23 * f.kxz;  f.kyzm = c.m23 * f.kyz;
    f.lambda = c.lambda; f.suth = c.suth; f.R = c.R; f.cv = c.cv; f.rcv = 1.0 / c.cv; f.k = c.k; f.dyef = c.dyef;
    f.c83h = c.c83h; f.c83n = c.c83n;
    f.cp = c.cv + c.R;
    f.il = 1.0 / c.lambda;  f.sl = c.suth / c.lambda;
    f.cUa = 2.0 * (f.kxxm + f.kyy);  f.cVb = 2.0 * (f.kxx + f.kyym);
    f.kxx2 = 2.0 * f.kxx;  f.kyy2 = 2.0 * f.kyy;
    f.kxxd = c.dyef * f.kxx;  f.kyyd = c.dyef * f.kyy;  f.cDg = 2.0 * (f.kxxd + f.kyyd);
    f.kzz2 = 2.0 * f.kzz;  f.kzzd = c.dyef * f.kzz;
    f.cUa3 = 2.0 * (f.kxxm + f.kyy + f.kzz);  f.cVb3 = 2.0 * (f.kxx + f.kyym + f.kzz);
    f.cWc3 = 2.0 * (f.kxx + f.kyy + f.kzzm);  f.cDg3 = 2.0 * (f.kxxd + f.kyyd + f.kzzd);
    return f;
}

// 1/x and sqrt(x) to full double accuracy (not correctly rounded) from the hardware estimates
// (relative error ~5e-8 on gfx950) by one third-order correction each:
//   1/x     = r (1 + e + e^2),          e = 1 - x r
//   sqrt(x) = g (1 + d + 1.5 d^2),      g = x y,  d = (1 - g y)/2,  y ~ 1/sqrt(x)
CFD_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
#else
    return 1.0 / x;
#endif
}

CFD_HD double fast_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double d = fma(-0.5 * g, y, 0.5);
    return fma(g, fma(1.5 * d, d, d), g);
#else
    return sqrt(x);
#endif
}

CFD_HD double mu_fast(const FastConsts& k, double t) {
    return (k.lambda * t) * (fast_sqrt(t) * fast_rcp(t + k.suth));
}

// What the East/West neighbours need from a cell.
struct ShareXF {
    double a, b, g;     // mu U, mu V, mu Dye at the centre
    double px;          // k T + c83h a^2 + c83n a + b^2/2 - b
    double q;           // N-minus-S difference of (mu_f U_f V_f - mu_f U_f - V_f)
    double dnb, dna;    // N-minus-S difference of mu_f V_f, of mu_f U_f
};
// What the North/South neighbours need from a cell.
struct ShareYF {
    double a, b, g;
    double py;          // k T + c83h b^2 + c83n b + a^2/2 - a
    double deb, dea;    // E-minus-W difference of mu_f V_f, of mu_f U_f
};
struct PhaseAF {
    double div[5];      // flux divergence per unit volume (E2_* order)
    double phi[5];      // the cell's own rho, U, V, T, Dye
    double rinv;        // 1/rho
    ShareXF sx;
    ShareYF sy;
};

// Rule for this section: a difference of two products that cancels exactly in the strict form
// (equal face states on opposite faces, equal neighbours) must cancel exactly here too, because
// the unknown's coefficient switches between 1 and rho on "phi == 0" and an update that turns an
// exact zero into 1e-17 changes the next sweep at O(1).  So products are rounded before they are
// subtracted (no contraction of a*b - c*d); fma is used only to combine already-formed
// differences with constants, and inside per-cell functions of a single state.
CFD_HD void phaseA2f(const FastConsts& k, const double c[NF2], const double hx[NF2],
                     const double n[NF2], const double s[NF2], PhaseAF& o) {
    double fE[NF2], fW[NF2], fN[NF2], fS[NF2];
    for (int q = 0; q < NF2; ++q) {
        const double hy = half_slope(n[q], c[q], s[q]);
        fE[q] = c[q] + hx[q]; fW[q] = c[q] - hx[q];
        fN[q] = c[q] + hy; fS[q] = c[q] - hy;
    }
    const double muC = mu_fast(k, c[F2_T]);
    const double muE = mu_fast(k, fE[F2_T]), muW = mu_fast(k, fW[F2_T]);
    const double muN = mu_fast(k, fN[F2_T]), muS = mu_fast(k, fS[F2_T]);
    const double pE = (k.R * fE[F2_T]) * fE[F2_RHO], pW = (k.R * fW[F2_T]) * fW[F2_RHO];
    const double pN = (k.R * fN[F2_T]) * fN[F2_RHO], pS = (k.R * fS[F2_T]) * fS[F2_RHO];
    const double mE = fE[F2_RHO] * fE[F2_U], mW = fW[F2_RHO] * fW[F2_U];
    const double mN = fN[F2_RHO] * fN[F2_V], mS = fS[F2_RHO] * fS[F2_V];

    // flux differences per direction (products rounded, then subtracted)
    const double dRx = mE - mW, dRy = mN - mS;
    const double dUx = (mE * fE[F2_U] - mW * fW[F2_U]) + (pE - pW), dUy = mN * fN[F2_U] - mS * fS[F2_U];
    const double dVx = mE * fE[F2_V] - mW * fW[F2_V], dVy = (mN * fN[F2_V] - mS * fS[F2_V]) + (pN - pS);
    const double dTx = k.cv * (mE * fE[F2_T] - mW * fW[F2_T]) + (pE * fE[F2_U] - pW * fW[F2_U]);
    const double dTy = k.cv * (mN * fN[F2_T] - mS * fS[F2_T]) + (pN * fN[F2_V] - pS * fS[F2_V]);
    const double dDx = mE * fE[F2_DYE] - mW * fW[F2_DYE], dDy = mN * fN[F2_DYE] - mS * fS[F2_DYE];
    o.div[E2_RHO] = fma(k.rcx, dRx, k.rcy * dRy);
    o.div[E2_U] = fma(k.rcx, dUx, k.rcy * dUy);
    o.div[E2_V] = fma(k.rcx, dVx, k.rcy * dVy);
    o.div[E2_T] = fma(k.rcx, dTx, k.rcy * dTy);
    o.div[E2_DYE] = fma(k.rcx, dDx, k.rcy * dDy);
    o.phi[E2_RHO] = c[F2_RHO]; o.phi[E2_U] = c[F2_U]; o.phi[E2_V] = c[F2_V];
    o.phi[E2_T] = c[F2_T];     o.phi[E2_DYE] = c[F2_DYE];
    o.rinv = fast_rcp(c[F2_RHO]);

    const double a = muC * c[F2_U], b = muC * c[F2_V], g = muC * c[F2_DYE];
    const double a2 = a * a, b2 = b * b, kt = k.k * c[F2_T];
    o.sx.a = a; o.sx.b = b; o.sx.g = g;
    o.sy.a = a; o.sy.b = b; o.sy.g = g;
    o.sx.px = (kt + fma(k.c83h, a2, k.c83n * a)) + fma(0.5, b2, -b);
    o.sy.py = (kt + fma(k.c83h, b2, k.c83n * b)) + fma(0.5, a2, -a);
    const double aN = muN * fN[F2_U], aS = muS * fS[F2_U], bN = muN * fN[F2_V], bS = muS * fS[F2_V];
    const double aE = muE * fE[F2_U], aW = muW * fW[F2_U], bE = muE * fE[F2_V], bW = muW * fW[F2_V];
    o.sx.dna = aN - aS;
    o.sx.dnb = bN - bS;
    o.sx.q = ((aN * fN[F2_V] - aS * fS[F2_V]) - o.sx.dna) - (fN[F2_V] - fS[F2_V]);
    o.sy.dea = aE - aW;
    o.sy.deb = bE - bW;
}

CFD_HD void phaseB2f(const FastConsts& k, const PhaseAF& C, const ShareXF& XE, const ShareXF& XW,
                     const ShareYF& YN, const ShareYF& YS, double out[5]) {
    const ShareXF& X = C.sx;
    const ShareYF& Y = C.sy;
    // second differences and cross differences, per unit volume
    const double rU = fma(k.kxxm, (XE.a + XW.a) - 2.0 * X.a,
                      fma(k.kyy, (YN.a + YS.a) - 2.0 * Y.a,
                      fma(k.kxy, XE.dnb - XW.dnb, k.kxym * (YN.deb - YS.deb))));
    const double rV = fma(k.kxx, (XE.b + XW.b) - 2.0 * X.b,
                      fma(k.kyym, (YN.b + YS.b) - 2.0 * Y.b,
                      fma(k.kxy, YN.dea - YS.dea, k.kxym * (XE.dna - XW.dna))));
    const double rT = fma(k.kxx, (XE.px + XW.px) - 2.0 * X.px,
                      fma(k.kyy, (YN.py + YS.py) - 2.0 * Y.py, k.kxy * (XE.q - XW.q)));
    const double rD = k.dyef * fma(k.kxx, (XE.g + XW.g) - 2.0 * X.g, k.kyy * ((YN.g + YS.g) - 2.0 * Y.g));
    const double gU = C.div[E2_U] - rU, gV = C.div[E2_V] - rV, gT = C.div[E2_T] - rT, gD = C.div[E2_DYE] - rD;
    // increment form; where phi is exactly zero the unknown's coefficient is 1 instead of rho
    out[E2_RHO] = C.phi[E2_RHO] - C.div[E2_RHO];
    out[E2_U] = fma(-gU, select(mask_of(C.phi[E2_U] != 0.0), C.rinv, 1.0), C.phi[E2_U]);
    out[E2_V] = fma(-gV, select(mask_of(C.phi[E2_V] != 0.0), C.rinv, 1.0), C.phi[E2_V]);
    out[E2_T] = fma(-gT, select(mask_of(C.phi[E2_T] != 0.0), C.rinv * k.rcv, 1.0), C.phi[E2_T]);
    out[E2_DYE] = fma(-gD, select(mask_of(C.phi[E2_DYE] != 0.0), C.rinv, 1.0), C.phi[E2_DYE]);
}

// ------------------------------------------------------------------------------------------
// 2-D case, streaming form of the re-associated arithmetic.  Same mathematics as the section
// above, arranged so that a marching kernel carries almost nothing from row to row: each cell
// turns its phase-A results at once into
//   * q = phi - coef * (div(flux) + centre part of the viscous/conductive terms), and
//   * one pre-scaled "message" per neighbour: the part of that neighbour's right-hand side this
//     cell contributes (second-difference term plus or minus the cross-difference term),
// and a cell's new value is  q + coef * (sum of the four messages addressed to it).
// coef is 1/rho (1/(rho cv) for Temperature), 1 where phi == 0 (the reference's NaN -> 1 rule) and 0
// for the fields an inactive cell keeps.  The exact-cancellation rule of the section above holds:
// a field that is identically zero yields zero messages, zero divergence and an unchanged zero.
// mu is evaluated with one combined second-order correction of the hardware reciprocal and
// reciprocal-square-root estimates (relative error ~1e-14; mu only enters terms that are many
// orders of magnitude below the field scale).
// ------------------------------------------------------------------------------------------
enum { S2_U = 0, S2_V, S2_T, S2_DYE };

CFD_HD double mu_stream(const FastConsts& k, double t) {
#if defined(__HIP_DEVICE_COMPILE__)
    // lambda t^1.5 / (t + S) = t^2 / sqrt(t ((t + S) / lambda)^2): one reciprocal-square-root estimate and
    // one Newton correction of it; lambda rides on the shifted temperature, which costs nothing
    const double z = fma(t, k.il, k.sl);            // (t + S) / lambda
    const double p = t * (z * z);
    const double y = __builtin_amdgcn_rsq(p);       // ~ lambda / (sqrt(t) (t + S))
    const double d2 = fma(-(p * y), y, 1.0);        // 1 - p y^2 = twice the relative error of y
    const double base = (t * t) * y;
    return fma(base, 0.5 * d2, base);
#else
    return (k.lambda * t) * (sqrt(t) / (t + k.suth));
#endif
}

struct StreamA2 {
    double rho;         // new Density (complete)
    double q[4];        // S2_* order
    double coef[4];
    double me[4];       // message to the West neighbour (this cell is its East)
    double mw[3];       // message to the East neighbour: U, V, T (its Dye entry equals me's)
    double ms[4];       // message to the South neighbour (this cell is its North)
    double mn[2];       // message to the North neighbour: U, V (its T and Dye entries equal ms's)
};

// Face differences without face values.  With the two limited states of a direction written as c + h and
// c - h (h = the half-slope), every upper-minus-lower difference of a product of them has a closed form
// with no cancellation:
//     (r+hr)(u+hu)         - (r-hr)(u-hu)         = 2 B,                B = r hu + hr u
//     (r+hr)(u+hu)(x+hx)   - (r-hr)(u-hu)(x-hx)   = 2 (B x + A hx),     A = r u + hr hu
//     mu+ (x+hx)           - mu- (x-hx)           = D x + S hx,         S = mu+ + mu-,  D = mu+ - mu-
//     mu+ (u+hu)(v+hv)     - mu- (u-hu)(v-hv)     = D (u v + hu hv) + S (u hv + hu v)
// (the factor 2 rides on the direction's constant).  Fewer operations than forming both face products and
// subtracting them, and no face value of Density, Dye or a velocity component is ever formed.  The
// exact-cancellation rule holds term by term: zero half-slopes give B = 0, D = 0 and exact zeros
// throughout, and a field that is identically zero (x = hx = 0) contributes exact zeros.
struct FluxDir {
    double A, B, dP, S, D;
};
// r, t: Density and Temperature of the cell, hr, ht: their half-slopes; u, hu: the velocity component normal
// to the faces.  dP = upper-minus-lower difference of T rho, halved.
CFD_HD FluxDir flux_dir(const FastConsts& k, double r, double hr, double t, double ht, double u, double hu) {
    FluxDir f;
    f.A = fma(hr, hu, r * u);
    f.B = fma(r, hu, hr * u);
    f.dP = fma(t, hr, r * ht);
    const double mU = mu_stream(k, t + ht), mL = mu_stream(k, t - ht);
    f.S = mU + mL;
    f.D = mU - mL;
    return f;
}
// half the upper-minus-lower difference of (rho u_normal x) over the direction's two faces
CFD_HD double flux_of(const FluxDir& f, double x, double hx) { return fma(f.B, x, f.A * hx); }
// upper-minus-lower difference of mu_f x_f
CFD_HD double visc_of(const FluxDir& f, double x, double hx) { return fma(f.D, x, f.S * hx); }
// upper-minus-lower difference of (mu_f u_f) v_f - mu_f u_f - v_f; dmu_u = visc_of(f, u, hu)
CFD_HD double paired_of(const FluxDir& f, double u, double hu, double v, double hv, double dmu_u) {
    const double p1 = fma(hu, hv, u * v), p2 = fma(hu, v, u * hv);
    return fma(-2.0, hv, fma(f.S, p2, f.D * p1) - dmu_u);
}

// Progress hook: a kernel whose waves run the same row in lock-step can have a wave lower its issue priority as it
// gets through the row (PhaseHook::at(1..3) after the x faces, the y faces and the messages), so that the waves
// sharing a SIMD finish the row together instead of oldest first.
struct NoPhaseHook { CFD_HD static void at(int) {} };

template <class S = SelMask, class H = NoPhaseHook>
CFD_HD void phaseA2s(const FastConsts& k, const double c[NF2], const double hx[NF2],
                     const double n[NF2], const double s[NF2], typename S::mask act, StreamA2& o) {
    const double rho = c[F2_RHO], T = c[F2_T];
    double div[5];
    // ---- x faces (pressure p = R (T rho): the face difference is taken on T rho and scaled by R once;
    // energy flux cv m T + p u = (cv + R) m T: times cp below, once)
    const FluxDir X = flux_dir(k, rho, hx[F2_RHO], T, hx[F2_T], c[F2_U], hx[F2_U]);
    div[E2_RHO] = k.rcx2 * X.B;
    div[E2_U] = k.rcx2 * fma(k.R, X.dP, flux_of(X, c[F2_U], hx[F2_U]));
    div[E2_V] = k.rcx2 * flux_of(X, c[F2_V], hx[F2_V]);
    div[E2_T] = k.rcx2 * flux_of(X, T, hx[F2_T]);
    div[E2_DYE] = k.rcx2 * flux_of(X, c[F2_DYE], hx[F2_DYE]);
    const double dea = visc_of(X, c[F2_U], hx[F2_U]), deb = visc_of(X, c[F2_V], hx[F2_V]);
    H::at(1);
    // ---- y faces
    double hy[NF2];
    for (int q = 0; q < NF2; ++q) hy[q] = half_slope(n[q], c[q], s[q]);
    const FluxDir Y = flux_dir(k, rho, hy[F2_RHO], T, hy[F2_T], c[F2_V], hy[F2_V]);
    div[E2_RHO] = fma(k.rcy2, Y.B, div[E2_RHO]);
    div[E2_U] = fma(k.rcy2, flux_of(Y, c[F2_U], hy[F2_U]), div[E2_U]);
    div[E2_V] = fma(k.rcy2, fma(k.R, Y.dP, flux_of(Y, c[F2_V], hy[F2_V])), div[E2_V]);
    div[E2_T] = fma(k.rcy2, flux_of(Y, T, hy[F2_T]), div[E2_T]);
    div[E2_DYE] = fma(k.rcy2, flux_of(Y, c[F2_DYE], hy[F2_DYE]), div[E2_DYE]);
    const double dna = visc_of(Y, c[F2_U], hy[F2_U]), dnb = visc_of(Y, c[F2_V], hy[F2_V]);
    const double qq = paired_of(Y, c[F2_U], hy[F2_U], c[F2_V], hy[F2_V], dna);
    H::at(2);
    // ---- centre products
    const double muC = mu_stream(k, c[F2_T]);
    const double a = muC * c[F2_U], b = muC * c[F2_V], g = muC * c[F2_DYE];
    const double a2 = a * a, b2 = b * b;
    const double px = fma(k.k, c[F2_T], fma(k.c83h, a2, k.c83n * a)) + fma(0.5, b2, -b);
    const double py = fma(k.k, c[F2_T], fma(k.c83h, b2, k.c83n * b)) + fma(0.5, a2, -a);
    // ---- messages
    // second-difference part plus / minus the cross-difference part, the latter folded in with one fused
    // operation per message (a zero field still yields zero messages)
    const double sxU = k.kxxm * a, sxV = k.kxx * b, sxT = k.kxx * px;
    o.me[S2_U] = fma(k.kxy, dnb, sxU);   o.mw[S2_U] = fma(-k.kxy, dnb, sxU);
    o.me[S2_V] = fma(k.kxym, dna, sxV);  o.mw[S2_V] = fma(-k.kxym, dna, sxV);
    o.me[S2_T] = fma(k.kxy, qq, sxT);    o.mw[S2_T] = fma(-k.kxy, qq, sxT);
    o.me[S2_DYE] = k.kxxd * g;
    const double syU = k.kyy * a, syV = k.kyym * b;
    o.ms[S2_U] = fma(k.kxym, deb, syU);  o.mn[S2_U] = fma(-k.kxym, deb, syU);
    o.ms[S2_V] = fma(k.kxy, dea, syV);   o.mn[S2_V] = fma(-k.kxy, dea, syV);
    o.ms[S2_T] = k.kyy * py;
    o.ms[S2_DYE] = k.kyyd * g;
    H::at(3);
    // ---- the cell's own part
    const double gU = fma(k.cUa, a, div[E2_U]), gV = fma(k.cVb, b, div[E2_V]);
    const double gT = fma(k.kxx2, px, fma(k.kyy2, py, k.cp * div[E2_T])), gD = fma(k.cDg, g, div[E2_DYE]);
    const double rinv = fast_rcp(c[F2_RHO]);
    const double keep = S::pick(act, 1.0, 0.0);
    o.coef[S2_U] = S::pick(act & S::of(c[F2_U] != 0.0), rinv, keep);
    o.coef[S2_V] = S::pick(act & S::of(c[F2_V] != 0.0), rinv, keep);
    o.coef[S2_T] = S::pick(S::of(c[F2_T] != 0.0), rinv * k.rcv, 1.0);
    o.coef[S2_DYE] = S::pick(act & S::of(c[F2_DYE] != 0.0), rinv, keep);
    o.q[S2_U] = fma(-o.coef[S2_U], gU, c[F2_U]);
    o.q[S2_V] = fma(-o.coef[S2_V], gV, c[F2_V]);
    o.q[S2_T] = fma(-o.coef[S2_T], gT, c[F2_T]);
    o.q[S2_DYE] = fma(-o.coef[S2_DYE], gD, c[F2_DYE]);
    o.rho = c[F2_RHO] - div[E2_RHO];
}

// q += coef * (message from the West neighbour); mwD: that message's Dye entry
CFD_HD void stream_west(double q[4], const double coef[4], const double mw[3], double mwD) {
    q[S2_U] = fma(coef[S2_U], mw[S2_U], q[S2_U]);
    q[S2_V] = fma(coef[S2_V], mw[S2_V], q[S2_V]);
    q[S2_T] = fma(coef[S2_T], mw[S2_T], q[S2_T]);
    q[S2_DYE] = fma(coef[S2_DYE], mwD, q[S2_DYE]);
}

// new value of one field: q (West already folded in) plus the East, North and South messages
CFD_HD double stream_finish(double q, double coef, double from_e, double from_n, double from_s) {
    return fma(coef, (from_e + from_n) + from_s, q);
}

// ------------------------------------------------------------------------------------------
// General 3-D case.  Field order of the 3-D arrays: U, V, W, Density, Temperature, Dye.
// Directions d = 0,1,2 (X,Y,Z); faces of direction d: index 0 = upper (East/North/Top),
// 1 = lower (West/South/Bottom).  Also valid for nz == 1 with W != 0 (Top/Bottom neighbours are
// then the cell itself).
// ------------------------------------------------------------------------------------------
enum { F3_U = 0, F3_V, F3_W, F3_RHO, F3_T, F3_DYE, NF3 };
enum { E3_RHO = 0, E3_U, E3_V, E3_W, E3_T, E3_DYE, NE3 };

// Centre products scaled by (A_d dt) for one direction d; differenced by the cell and its +-d
// neighbours.  m[c] = mu u_c; sqh[c] = ((mu u_c) mu) u_c * 0.5 (used for c != d).
struct SameDir3 {
    double m[3];
    double m23;     // (mu u_d) * m23
    double t;       // T * k
    double sq83;    // (((mu u_d) mu) u_d) * c83h
    double m83;     // (mu u_d) * c83n
    double sqh[3];
    double g;       // (mu Dye) * dyef
};

// What the +-X neighbours difference: this cell's products on its Y and Z faces.
struct ShareX3 {
    SameDir3 s;
    double yMuV[2], yMuU23[2], yMuUV[2], yMuU[2], yV[2];
    double zMuW[2], zMuU23[2], zMuUW[2], zMuU[2], zW[2];
};
// What the +-Y neighbours difference: products on the X and Z faces.
struct ShareY3 {
    SameDir3 s;
    double xMuV23[2], xMuU[2];
    double zMuW[2], zMuV23[2];
};
// What the +-Z neighbours difference: products on the X and Y faces.
struct ShareZ3 {
    SameDir3 s;
    double xMuW23[2], xMuU[2];
    double yMuW23[2], yMuV[2], yMuWV[2], yMuW[2], yV[2];
};

struct PhaseA3 {
    double lhs[NE3];
    double aU[NE3];
    double own[4];      // stored U, V, W, Dye (kept by inactive cells)
    ShareX3 sx;
    ShareY3 sy;
    ShareZ3 sz;
};

// Limited half-slopes of the six stored values along the three directions; nb[2*d+s]: the values
// of the clamped neighbour on face (d, s).
CFD_HD void half_slopes3(const double c[NF3], const double nb[6][NF3], double h[3][NF3]) {
    for (int d = 0; d < 3; ++d)
        for (int q = 0; q < NF3; ++q) h[d][q] = half_slope(nb[2 * d][q], c[q], nb[2 * d + 1][q]);
}

// c: the cell's six stored values; h: their limited half-slopes (half_slopes3).
CFD_HD void phaseA3_h(const SweepConsts& k, const double c[NF3], const double h[3][NF3], PhaseA3& o) {
    const double A[3] = {k.Ae, k.An, k.At};
    const double dt = k.dt;
    double f[3][2][NF3];          // limited face values
    for (int d = 0; d < 3; ++d)
        for (int q = 0; q < NF3; ++q) {
            f[d][0][q] = c[q] + h[d][q];
            f[d][1][q] = c[q] - h[d][q];
        }
    o.own[0] = c[F3_U]; o.own[1] = c[F3_V]; o.own[2] = c[F3_W]; o.own[3] = c[F3_DYE];
    double mu[3][2], p[3][2];
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s) {
            mu[d][s] = mu_of(k, f[d][s][F3_T]);
            p[d][s] = (k.R * f[d][s][F3_T]) * f[d][s][F3_RHO];
        }
    const double muC = mu_of(k, c[F3_T]);

    double F, l;
    // ---- continuity
    F = c[F3_RHO];
    o.aU[E3_RHO] = unknown_coeff_normal(F, c[F3_RHO]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    for (int d = 0; d < 3; ++d) {
        l = l + ((f[d][0][F3_RHO] * f[d][0][F3_U + d]) * A[d]) * dt;
        l = l - ((f[d][1][F3_RHO] * f[d][1][F3_U + d]) * A[d]) * dt;
    }
    o.lhs[E3_RHO] = l;
    // ---- momentum, component cc (equation slot E3_U + cc)
    for (int cc = 0; cc < 3; ++cc) {
        F = c[F3_RHO] * c[F3_U + cc];
        o.aU[E3_U + cc] = unknown_coeff(F, c[F3_U + cc]) * k.vol;
        l = 0.0 + (-F) * k.vol;
        for (int d = 0; d < 3; ++d) {
            l = l + (((f[d][0][F3_RHO] * f[d][0][F3_U + cc]) * f[d][0][F3_U + d]) * A[d]) * dt;
            l = l - (((f[d][1][F3_RHO] * f[d][1][F3_U + cc]) * f[d][1][F3_U + d]) * A[d]) * dt;
        }
        l = l + (p[cc][0] * A[cc]) * dt;
        l = l - (p[cc][1] * A[cc]) * dt;
        o.lhs[E3_U + cc] = l;
    }
    // ---- energy
    F = (c[F3_RHO] * c[F3_T]) * k.cv;
    o.aU[E3_T] = unknown_coeff_normal(F, c[F3_T]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    for (int d = 0; d < 3; ++d) {
        l = l + ((((f[d][0][F3_RHO] * f[d][0][F3_T]) * f[d][0][F3_U + d]) * k.cv) * A[d]) * dt;
        l = l - ((((f[d][1][F3_RHO] * f[d][1][F3_T]) * f[d][1][F3_U + d]) * k.cv) * A[d]) * dt;
    }
    for (int d = 0; d < 3; ++d) {
        l = l + ((p[d][0] * f[d][0][F3_U + d]) * A[d]) * dt;
        l = l - ((p[d][1] * f[d][1][F3_U + d]) * A[d]) * dt;
    }
    o.lhs[E3_T] = l;
    // ---- dye
    F = c[F3_RHO] * c[F3_DYE];
    o.aU[E3_DYE] = unknown_coeff(F, c[F3_DYE]) * k.vol;
    l = 0.0 + (-F) * k.vol;
    for (int d = 0; d < 3; ++d) {
        l = l + (((f[d][0][F3_RHO] * f[d][0][F3_DYE]) * f[d][0][F3_U + d]) * A[d]) * dt;
        l = l - (((f[d][1][F3_RHO] * f[d][1][F3_DYE]) * f[d][1][F3_U + d]) * A[d]) * dt;
    }
    o.lhs[E3_DYE] = l;

    // ---- centre products, scaled per direction
    double a[3], a2[3];
    for (int cc = 0; cc < 3; ++cc) {
        a[cc] = muC * c[F3_U + cc];
        a2[cc] = (a[cc] * muC) * c[F3_U + cc];
    }
    const double g = muC * c[F3_DYE];
    const double tk = c[F3_T] * k.k, gd = g * k.dyef;
    SameDir3* S[3] = {&o.sx.s, &o.sy.s, &o.sz.s};
    for (int d = 0; d < 3; ++d) {
        SameDir3& s = *S[d];
        for (int cc = 0; cc < 3; ++cc) {
            s.m[cc] = (a[cc] * A[d]) * dt;
            s.sqh[cc] = ((a2[cc] * 0.5) * A[d]) * dt;
        }
        s.m23 = ((a[d] * k.m23) * A[d]) * dt;
        s.t = (tk * A[d]) * dt;
        s.sq83 = ((a2[d] * k.c83h) * A[d]) * dt;
        s.m83 = ((a[d] * k.c83n) * A[d]) * dt;
        s.g = (gd * A[d]) * dt;
    }
    // ---- face products: mf[d][s][cc] = mu_f * (u_cc)_f on face (d, s)
    double mf[3][2][3];
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s)
            for (int cc = 0; cc < 3; ++cc) mf[d][s][cc] = mu[d][s] * f[d][s][F3_U + cc];
    const double Ax = A[0], Ay = A[1], Az = A[2];
    for (int s = 0; s < 2; ++s) {
        // consumed by the +-X neighbours
        o.sx.yMuV[s] = (mf[1][s][1] * Ay) * dt;
        o.sx.yMuU23[s] = ((mf[1][s][0] * k.m23) * Ay) * dt;
        o.sx.yMuUV[s] = ((mf[1][s][0] * f[1][s][F3_V]) * Ay) * dt;
        o.sx.yMuU[s] = (mf[1][s][0] * Ay) * dt;
        o.sx.yV[s] = (f[1][s][F3_V] * Ay) * dt;
        o.sx.zMuW[s] = (mf[2][s][2] * Az) * dt;
        o.sx.zMuU23[s] = ((mf[2][s][0] * k.m23) * Az) * dt;
        o.sx.zMuUW[s] = ((mf[2][s][0] * f[2][s][F3_W]) * Az) * dt;
        o.sx.zMuU[s] = (mf[2][s][0] * Az) * dt;
        o.sx.zW[s] = (f[2][s][F3_W] * Az) * dt;
        // consumed by the +-Y neighbours
        o.sy.xMuV23[s] = ((mf[0][s][1] * k.m23) * Ax) * dt;
        o.sy.xMuU[s] = (mf[0][s][0] * Ax) * dt;
        o.sy.zMuW[s] = (mf[2][s][2] * Az) * dt;
        o.sy.zMuV23[s] = ((mf[2][s][1] * k.m23) * Az) * dt;
        // consumed by the +-Z neighbours
        o.sz.xMuW23[s] = ((mf[0][s][2] * k.m23) * Ax) * dt;
        o.sz.xMuU[s] = (mf[0][s][0] * Ax) * dt;
        o.sz.yMuW23[s] = ((mf[1][s][2] * k.m23) * Ay) * dt;
        o.sz.yMuV[s] = (mf[1][s][1] * Ay) * dt;
        o.sz.yMuWV[s] = ((mf[1][s][2] * f[1][s][F3_V]) * Ay) * dt;
        o.sz.yMuW[s] = (mf[1][s][2] * Ay) * dt;
        o.sz.yV[s] = (f[1][s][F3_V] * Ay) * dt;
    }
}

CFD_HD void phaseA3(const SweepConsts& k, const double c[NF3], const double nb[6][NF3], PhaseA3& o) {
    double h[3][NF3];
    half_slopes3(c, nb, h);
    phaseA3_h(k, c, h, o);
}

// Right-hand sides and solve.  XP/XM, YP/YM, ZP/ZM: shares of the +/- neighbours in x, y, z (a
// clamped neighbour is the cell itself).  out[] in E3_* order.
CFD_HD void phaseB3(const SweepConsts& k, const PhaseA3& C, const ShareX3& XP, const ShareX3& XM,
                    const ShareY3& YP, const ShareY3& YM, const ShareZ3& ZP, const ShareZ3& ZM, double out[NE3]) {
    const double hx = k.hx, hy = k.hy, hz = k.hz, Hx = k.Hx, Hy = k.Hy, Hz = k.Hz;
    const double rhx = k.rhx, rhy = k.rhy, rhz = k.rhz, rHx = k.rHx, rHy = k.rHy, rHz = k.rHz;
    const double lhx = k.lhx, lhy = k.lhy, lhz = k.lhz, lHx = k.lHx, lHy = k.lHy, lHz = k.lHz;
    const SameDir3 &cx = C.sx.s, &cy = C.sy.s, &cz = C.sz.s;
    const SameDir3 &xp = XP.s, &xm = XM.s, &yp = YP.s, &ym = YM.s, &zp = ZP.s, &zm = ZM.s;
    double r;
#define SAME(r, up, lo, cc, h)   do { (r) = (r) + div_by((up) - (cc), h, r##h, l##h); (r) = (r) + div_by((lo) - (cc), h, r##h, l##h); } while (0)
#define CROSS(r, n1, n2, fld, H)  do { (r) = (r) + div_by((n1).fld[0] - (n2).fld[0], H, r##H, l##H); (r) = (r) + div_by((n2).fld[1] - (n1).fld[1], H, r##H, l##H); } while (0)
#define CROSSN(r, n1, n2, fld, H) do { (r) = (r) + div_by((n2).fld[0] - (n1).fld[0], H, r##H, l##H); (r) = (r) + div_by((n1).fld[1] - (n2).fld[1], H, r##H, l##H); } while (0)
    // Quotients that a momentum equation and the energy equation use with opposite signs (the momentum equations take
    // d(mu u)/dn of the upper and lower neighbour, the energy equation's square-derivative terms the same differences
    // reversed): x - y is the exact negative of y - x and every step of div_by is odd, so the reversed quotient is the
    // negative bit for bit and adding it is subtracting this one.  (A zero quotient changes sign that way; it is only
    // ever added to a sum that started at +0, which no signed zero can turn into -0.)  Taken once, added there,
    // subtracted here: 12 divisions fewer per cell.  (The 2-D form has four such pairs; sharing them there made the
    // kernel slower -- two more live values in a kernel that is short of registers.)
    const double dYu0 = div_by(yp.m[0] - cy.m[0], hy, rhy, lhy), dYl0 = div_by(ym.m[0] - cy.m[0], hy, rhy, lhy);
    const double dZu0 = div_by(zp.m[0] - cz.m[0], hz, rhz, lhz), dZl0 = div_by(zm.m[0] - cz.m[0], hz, rhz, lhz);
    const double dXu1 = div_by(xp.m[1] - cx.m[1], hx, rhx, lhx), dXl1 = div_by(xm.m[1] - cx.m[1], hx, rhx, lhx);
    const double dZu1 = div_by(zp.m[1] - cz.m[1], hz, rhz, lhz), dZl1 = div_by(zm.m[1] - cz.m[1], hz, rhz, lhz);
    const double dXu2 = div_by(xp.m[2] - cx.m[2], hx, rhx, lhx), dXl2 = div_by(xm.m[2] - cx.m[2], hx, rhx, lhx);
    const double dYu2 = div_by(yp.m[2] - cy.m[2], hy, rhy, lhy), dYl2 = div_by(ym.m[2] - cy.m[2], hy, rhy, lhy);
#define TWO(r, u, l)    do { (r) = (r) + (u); (r) = (r) + (l); } while (0)
#define TWON(r, u, l)   do { (r) = (r) - (u); (r) = (r) - (l); } while (0)
    // ---- continuity
    r = 0.0 + 0.0;
    out[E3_RHO] = div_normal(r - C.lhs[E3_RHO], C.aU[E3_RHO]);
    // ---- U momentum
    r = 0.0;
    SAME(r, xp.m[0], xm.m[0], cx.m[0], hx);
    TWO(r, dYu0, dYl0);
    TWO(r, dZu0, dZl0);
    SAME(r, xp.m[0], xm.m[0], cx.m[0], hx);
    CROSS(r, XP, XM, yMuV, Hx);
    CROSS(r, XP, XM, zMuW, Hx);
    SAME(r, xp.m23, xm.m23, cx.m23, hx);
    CROSS(r, YP, YM, xMuV23, Hy);
    CROSS(r, ZP, ZM, xMuW23, Hz);
    out[E3_U] = (r - C.lhs[E3_U]) / C.aU[E3_U];
    // ---- V momentum
    r = 0.0;
    TWO(r, dXu1, dXl1);
    SAME(r, yp.m[1], ym.m[1], cy.m[1], hy);
    TWO(r, dZu1, dZl1);
    CROSS(r, YP, YM, xMuU, Hy);
    SAME(r, yp.m[1], ym.m[1], cy.m[1], hy);
    CROSS(r, YP, YM, zMuW, Hy);
    CROSS(r, XP, XM, yMuU23, Hx);
    SAME(r, yp.m23, ym.m23, cy.m23, hy);
    CROSS(r, ZP, ZM, yMuW23, Hz);
    out[E3_V] = (r - C.lhs[E3_V]) / C.aU[E3_V];
    // ---- W momentum
    r = 0.0;
    TWO(r, dXu2, dXl2);
    TWO(r, dYu2, dYl2);
    SAME(r, zp.m[2], zm.m[2], cz.m[2], hz);
    CROSS(r, ZP, ZM, xMuU, Hz);
    CROSS(r, ZP, ZM, yMuV, Hz);
    SAME(r, zp.m[2], zm.m[2], cz.m[2], hz);
    CROSS(r, XP, XM, zMuU23, Hx);
    CROSS(r, YP, YM, zMuV23, Hy);
    SAME(r, zp.m23, zm.m23, cz.m23, hz);
    out[E3_W] = (r - C.lhs[E3_W]) / C.aU[E3_W];
    // ---- energy
    r = 0.0;
    SAME(r, xp.t, xm.t, cx.t, hx);
    SAME(r, yp.t, ym.t, cy.t, hy);
    SAME(r, zp.t, zm.t, cz.t, hz);
    SAME(r, xp.sq83, xm.sq83, cx.sq83, hx);   SAME(r, xp.m83, xm.m83, cx.m83, hx);      // [Mew,U] 8/3 X
    SAME(r, yp.sq83, ym.sq83, cy.sq83, hy);   SAME(r, yp.m83, ym.m83, cy.m83, hy);      // [Mew,V] 8/3 Y
    SAME(r, zp.sq83, zm.sq83, cz.sq83, hz);   SAME(r, zp.m83, zm.m83, cz.m83, hz);      // [Mew,W] 8/3 Z
    SAME(r, yp.sqh[0], ym.sqh[0], cy.sqh[0], hy);  TWON(r, dYu0, dYl0);   // [Mew,U] 1 Y
    SAME(r, xp.sqh[1], xm.sqh[1], cx.sqh[1], hx);  TWON(r, dXu1, dXl1);   // [Mew,V] 1 X
    SAME(r, zp.sqh[0], zm.sqh[0], cz.sqh[0], hz);  TWON(r, dZu0, dZl0);   // [Mew,U] 1 Z
    SAME(r, xp.sqh[2], xm.sqh[2], cx.sqh[2], hx);  TWON(r, dXu2, dXl2);   // [Mew,W] 1 X
    SAME(r, zp.sqh[1], zm.sqh[1], cz.sqh[1], hz);  TWON(r, dZu1, dZl1);   // [Mew,V] 1 Z
    SAME(r, yp.sqh[2], ym.sqh[2], cy.sqh[2], hy);  TWON(r, dYu2, dYl2);   // [Mew,W] 1 Y
    CROSS(r, XP, XM, yMuUV, Hx);  CROSSN(r, XP, XM, yMuU, Hx);  CROSSN(r, XP, XM, yV, Hx);    // [Mew,U][V] X Y
    CROSS(r, XP, XM, zMuUW, Hx);  CROSSN(r, XP, XM, zMuU, Hx);  CROSSN(r, XP, XM, zW, Hx);    // [Mew,U][W] X Z
    CROSS(r, ZP, ZM, yMuWV, Hz);  CROSSN(r, ZP, ZM, yMuW, Hz);  CROSSN(r, ZP, ZM, yV, Hz);    // [Mew,W][V] Z Y
    out[E3_T] = div_normal(r - C.lhs[E3_T], C.aU[E3_T]);
    // ---- dye
    // plain divisions (see the 2-D Dye equation)
    r = 0.0;
    r = r + (xp.g - cx.g) / hx;  r = r + (xm.g - cx.g) / hx;
    r = r + (yp.g - cy.g) / hy;  r = r + (ym.g - cy.g) / hy;
    r = r + (zp.g - cz.g) / hz;  r = r + (zm.g - cz.g) / hz;
    out[E3_DYE] = (r - C.lhs[E3_DYE]) / C.aU[E3_DYE];
#undef TWO
#undef TWON
#undef SAME
#undef CROSS
#undef CROSSN
}


// ------------------------------------------------------------------------------------------
// General 3-D case, re-associated arithmetic: the increment form of the 2-D fast section with all
// three directions.  The same exact-cancellation rule applies (products are rounded before they
// are subtracted).
// ------------------------------------------------------------------------------------------
// Products a cell publishes for its +-X, +-Y, +-Z neighbours.  a[c] = mu u_c and g = mu Dye at the
// centre; p = the combination whose second difference along that direction enters the energy
// equation; d<F><q> = upper-minus-lower difference over the F faces of mu_f (u_q)_f.
struct ShareX3F { double a[3], g, p, dYv, dZw, dYu, dZu, qY, qZ; };
struct ShareY3F { double a[3], g, p, dXv, dXu, dZw, dZv; };
struct ShareZ3F { double a[3], g, p, dXw, dYw, dXu, dYv, qY; };

struct PhaseA3F {
    double div[NE3];    // flux divergence per unit volume (E3_* order)
    double phi[NE3];    // the cell's own rho, U, V, W, T, Dye
    double rinv;        // 1/rho
    ShareX3F sx;
    ShareY3F sy;
    ShareZ3F sz;
};

CFD_HD void phaseA3f_h(const FastConsts& k, const double c[NF3], const double h[3][NF3], PhaseA3F& o) {
    double f[3][2][NF3];
    for (int d = 0; d < 3; ++d)
        for (int q = 0; q < NF3; ++q) {
            f[d][0][q] = c[q] + h[d][q];
            f[d][1][q] = c[q] - h[d][q];
        }
    const double rc[3] = {k.rcx, k.rcy, k.rcz};
    double div[NE3] = {0, 0, 0, 0, 0, 0};
    double muf[3][2];
    double dmu[3][3];       // dmu[d][q] = mu_f (u_q)_f on the upper face of direction d minus the lower face
    for (int d = 0; d < 3; ++d) {
        double m[2], p[2];
        for (int s = 0; s < 2; ++s) {
            muf[d][s] = mu_fast(k, f[d][s][F3_T]);
            p[s] = (k.R * f[d][s][F3_T]) * f[d][s][F3_RHO];
            m[s] = f[d][s][F3_RHO] * f[d][s][F3_U + d];
        }
        // flux differences across direction d (products rounded, then subtracted)
        double dd[NE3];
        dd[E3_RHO] = m[0] - m[1];
        for (int cc = 0; cc < 3; ++cc) {
            dd[E3_U + cc] = m[0] * f[d][0][F3_U + cc] - m[1] * f[d][1][F3_U + cc];
            if (cc == d) dd[E3_U + cc] = dd[E3_U + cc] + (p[0] - p[1]);
        }
        dd[E3_T] = k.cv * (m[0] * f[d][0][F3_T] - m[1] * f[d][1][F3_T])
                 + (p[0] * f[d][0][F3_U + d] - p[1] * f[d][1][F3_U + d]);
        dd[E3_DYE] = m[0] * f[d][0][F3_DYE] - m[1] * f[d][1][F3_DYE];
        for (int e = 0; e < NE3; ++e) div[e] = fma(rc[d], dd[e], div[e]);
        for (int q = 0; q < 3; ++q) dmu[d][q] = muf[d][0] * f[d][0][F3_U + q] - muf[d][1] * f[d][1][F3_U + q];
    }
    for (int e = 0; e < NE3; ++e) o.div[e] = div[e];
    o.phi[E3_RHO] = c[F3_RHO]; o.phi[E3_U] = c[F3_U]; o.phi[E3_V] = c[F3_V]; o.phi[E3_W] = c[F3_W];
    o.phi[E3_T] = c[F3_T];     o.phi[E3_DYE] = c[F3_DYE];
    o.rinv = fast_rcp(c[F3_RHO]);

    const double muC = mu_fast(k, c[F3_T]);
    double a[3], a2[3];
    for (int q = 0; q < 3; ++q) { a[q] = muC * c[F3_U + q]; a2[q] = a[q] * a[q]; }
    const double g = muC * c[F3_DYE], kt = k.k * c[F3_T];
    // energy combinations: along d the component d carries 8/3 (c83h a_d^2 + c83n a_d), the other
    // two components carry 1 (a_q^2/2 - a_q)
    double pc[3];
    for (int d = 0; d < 3; ++d) {
        double v = kt + fma(k.c83h, a2[d], k.c83n * a[d]);
        for (int q = 0; q < 3; ++q)
            if (q != d) v = v + fma(0.5, a2[q], -a[q]);
        pc[d] = v;
    }
    for (int q = 0; q < 3; ++q) { o.sx.a[q] = a[q]; o.sy.a[q] = a[q]; o.sz.a[q] = a[q]; }
    o.sx.g = g; o.sy.g = g; o.sz.g = g;
    o.sx.p = pc[0]; o.sy.p = pc[1]; o.sz.p = pc[2];
    // paired terms: difference over the faces of direction dout of (mu u_a) u_b - mu u_a - u_b
    auto paired = [&](int dout, int qa, int qb) {
        const double up = (muf[dout][0] * f[dout][0][F3_U + qa]) * f[dout][0][F3_U + qb];
        const double lo = (muf[dout][1] * f[dout][1][F3_U + qa]) * f[dout][1][F3_U + qb];
        return ((up - lo) - dmu[dout][qa]) - (f[dout][0][F3_U + qb] - f[dout][1][F3_U + qb]);
    };
    // the +-X neighbours difference this cell's Y- and Z-face products
    o.sx.dYv = dmu[1][1];  o.sx.dZw = dmu[2][2];  o.sx.dYu = dmu[1][0];  o.sx.dZu = dmu[2][0];
    o.sx.qY = paired(1, 0, 1);      // [Mew,U][V] X Y
    o.sx.qZ = paired(2, 0, 2);      // [Mew,U][W] X Z
    // the +-Y neighbours: X- and Z-face products
    o.sy.dXv = dmu[0][1];  o.sy.dXu = dmu[0][0];  o.sy.dZw = dmu[2][2];  o.sy.dZv = dmu[2][1];
    // the +-Z neighbours: X- and Y-face products
    o.sz.dXw = dmu[0][2];  o.sz.dYw = dmu[1][2];  o.sz.dXu = dmu[0][0];  o.sz.dYv = dmu[1][1];
    o.sz.qY = paired(1, 2, 1);      // [Mew,W][V] Z Y
}

CFD_HD void phaseA3f(const FastConsts& k, const double c[NF3], const double nb[6][NF3], PhaseA3F& o) {
    double h[3][NF3];
    half_slopes3(c, nb, h);
    phaseA3f_h(k, c, h, o);
}

CFD_HD void phaseB3f(const FastConsts& k, const PhaseA3F& C, const ShareX3F& XP, const ShareX3F& XM,
                     const ShareY3F& YP, const ShareY3F& YM, const ShareZ3F& ZP, const ShareZ3F& ZM,
                     double out[NE3]) {
    // second differences of the centre products along x, y, z
#define D2X(fld) ((XP.fld + XM.fld) - 2.0 * C.sx.fld)
#define D2Y(fld) ((YP.fld + YM.fld) - 2.0 * C.sy.fld)
#define D2Z(fld) ((ZP.fld + ZM.fld) - 2.0 * C.sz.fld)
    const double rU = fma(k.kxxm, D2X(a[0]), fma(k.kyy, D2Y(a[0]), fma(k.kzz, D2Z(a[0]),
                      fma(k.kxy, XP.dYv - XM.dYv, fma(k.kxz, XP.dZw - XM.dZw,
                      fma(k.kxym, YP.dXv - YM.dXv, k.kxzm * (ZP.dXw - ZM.dXw)))))));
    const double rV = fma(k.kxx, D2X(a[1]), fma(k.kyym, D2Y(a[1]), fma(k.kzz, D2Z(a[1]),
                      fma(k.kxy, YP.dXu - YM.dXu, fma(k.kyz, YP.dZw - YM.dZw,
                      fma(k.kxym, XP.dYu - XM.dYu, k.kyzm * (ZP.dYw - ZM.dYw)))))));
    const double rW = fma(k.kxx, D2X(a[2]), fma(k.kyy, D2Y(a[2]), fma(k.kzzm, D2Z(a[2]),
                      fma(k.kxz, ZP.dXu - ZM.dXu, fma(k.kyz, ZP.dYv - ZM.dYv,
                      fma(k.kxzm, XP.dZu - XM.dZu, k.kyzm * (YP.dZv - YM.dZv)))))));
    const double rT = fma(k.kxx, D2X(p), fma(k.kyy, D2Y(p), fma(k.kzz, D2Z(p),
                      fma(k.kxy, XP.qY - XM.qY, fma(k.kxz, XP.qZ - XM.qZ, k.kyz * (ZP.qY - ZM.qY))))));
    const double rD = k.dyef * fma(k.kxx, D2X(g), fma(k.kyy, D2Y(g), k.kzz * D2Z(g)));
#undef D2X
#undef D2Y
#undef D2Z
    const double gU = C.div[E3_U] - rU, gV = C.div[E3_V] - rV, gW = C.div[E3_W] - rW;
    const double gT = C.div[E3_T] - rT, gD = C.div[E3_DYE] - rD;
    out[E3_RHO] = C.phi[E3_RHO] - C.div[E3_RHO];
    out[E3_U] = fma(-gU, select(mask_of(C.phi[E3_U] != 0.0), C.rinv, 1.0), C.phi[E3_U]);
    out[E3_V] = fma(-gV, select(mask_of(C.phi[E3_V] != 0.0), C.rinv, 1.0), C.phi[E3_V]);
    out[E3_W] = fma(-gW, select(mask_of(C.phi[E3_W] != 0.0), C.rinv, 1.0), C.phi[E3_W]);
    out[E3_T] = fma(-gT, select(mask_of(C.phi[E3_T] != 0.0), C.rinv * k.rcv, 1.0), C.phi[E3_T]);
    out[E3_DYE] = fma(-gD, select(mask_of(C.phi[E3_DYE] != 0.0), C.rinv, 1.0), C.phi[E3_DYE]);
}


// ------------------------------------------------------------------------------------------
// General 3-D case, streaming form: the 3-D counterpart of phaseA2s (q, coef and one pre-scaled
// message per face neighbour).  Equation slots: U, V, W, Temperature, Dye.
// ------------------------------------------------------------------------------------------
enum { S3_U = 0, S3_V, S3_W, S3_T, S3_DYE, NS3 };

struct StreamA3 {
    double rho;         // new Density (complete)
    double q[NS3];
    double coef[NS3];
    double me[NS3];     // message to the West neighbour (this cell is its East)
    double mw[4];       // message to the East neighbour: U, V, W, T (its Dye entry equals me's)
    double ms[NS3];     // message to the South neighbour (this cell is its North)
    double mn[3];       // message to the North neighbour: U, V, W (T and Dye entries equal ms's)
    double mb[NS3];     // message to the Bottom neighbour (this cell is its Top)
    double mt[4];       // message to the Top neighbour: U, V, W, T (Dye entry equals mb's)
};

// slopes(d, hd): fills hd[NF3] with the limited half-slopes of the six stored values along direction d; it is
// called once per direction, in the order 0, 1, 2, right before that direction's faces are worked on, so that a
// kernel can fetch its neighbours' values late and nothing of a direction outlives it but its sums.
template <class SlopeFn>
CFD_HD void phaseA3s_by_direction(const FastConsts& k, const double c[NF3], SlopeFn&& slopes, lane_mask act, StreamA3& o) {
    const double rc2[3] = {k.rcx2, k.rcy2, k.rcz2};
    const double rho = c[F3_RHO], T = c[F3_T];
    double div[NE3] = {0, 0, 0, 0, 0, 0};
    double dmu[3][3];       // dmu[d][q] = mu_f (u_q)_f on the upper face of direction d minus the lower face
    double pair[3];         // pair[0] = X-neighbours' Y-face term, [1] = X-neighbours' Z-face term, [2] = Z-neighbours' Y-face term
    for (int d = 0; d < 3; ++d) {
        double h[NF3];
        slopes(d, h);
        const FluxDir F = flux_dir(k, rho, h[F3_RHO], T, h[F3_T], c[F3_U + d], h[F3_U + d]);
        // flux differences across direction d (see flux_dir: halved, no face values)
        div[E3_RHO] = fma(rc2[d], F.B, div[E3_RHO]);
        for (int cc = 0; cc < 3; ++cc) {
            double dd = flux_of(F, c[F3_U + cc], h[F3_U + cc]);
            if (cc == d) dd = fma(k.R, F.dP, dd);
            div[E3_U + cc] = fma(rc2[d], dd, div[E3_U + cc]);
        }
        div[E3_T] = fma(rc2[d], flux_of(F, T, h[F3_T]), div[E3_T]);       // cv m T + p u = (cv + R) m T: times cp below, once
        div[E3_DYE] = fma(rc2[d], flux_of(F, c[F3_DYE], h[F3_DYE]), div[E3_DYE]);
        for (int q = 0; q < 3; ++q) dmu[d][q] = visc_of(F, c[F3_U + q], h[F3_U + q]);
        // paired terms: difference over the faces of d of (mu u_a) u_b - mu u_a - u_b
        if (d == 1) {
            pair[0] = paired_of(F, c[F3_U], h[F3_U], c[F3_V], h[F3_V], dmu[1][0]);     // [Mew,U][V] X Y
            pair[2] = paired_of(F, c[F3_W], h[F3_W], c[F3_V], h[F3_V], dmu[1][2]);     // [Mew,W][V] Z Y
        }
        if (d == 2) pair[1] = paired_of(F, c[F3_U], h[F3_U], c[F3_W], h[F3_W], dmu[2][0]);     // [Mew,U][W] X Z
    }
    // ---- centre products
    const double muC = mu_stream(k, c[F3_T]);
    double a[3], a2[3];
    for (int q = 0; q < 3; ++q) { a[q] = muC * c[F3_U + q]; a2[q] = a[q] * a[q]; }
    const double g = muC * c[F3_DYE];
    double pc[3];
    for (int d = 0; d < 3; ++d) {
        double v = fma(k.k, c[F3_T], fma(k.c83h, a2[d], k.c83n * a[d]));
        for (int q = 0; q < 3; ++q)
            if (q != d) v = v + fma(0.5, a2[q], -a[q]);
        pc[d] = v;
    }
    // ---- messages: second-difference part s, cross-difference part t; s + t goes to the lower
    // neighbour (this cell is its upper one), s - t to the upper neighbour.  A one-product t is folded
    // into s with one fused operation per message.
    {   // X
        const double sU = k.kxxm * a[0], sV = k.kxx * a[1], sW = k.kxx * a[2], sT = k.kxx * pc[0];
        const double tU = fma(k.kxy, dmu[1][1], k.kxz * dmu[2][2]);
        const double tT = fma(k.kxy, pair[0], k.kxz * pair[1]);
        o.me[S3_U] = sU + tU; o.me[S3_V] = fma(k.kxym, dmu[1][0], sV); o.me[S3_W] = fma(k.kxzm, dmu[2][0], sW);
        o.me[S3_T] = sT + tT; o.me[S3_DYE] = k.kxxd * g;
        o.mw[S3_U] = sU - tU; o.mw[S3_V] = fma(-k.kxym, dmu[1][0], sV); o.mw[S3_W] = fma(-k.kxzm, dmu[2][0], sW);
        o.mw[S3_T] = sT - tT;
    }
    {   // Y
        const double sU = k.kyy * a[0], sV = k.kyym * a[1], sW = k.kyy * a[2];
        const double tV = fma(k.kxy, dmu[0][0], k.kyz * dmu[2][2]);
        o.ms[S3_U] = fma(k.kxym, dmu[0][1], sU); o.ms[S3_V] = sV + tV; o.ms[S3_W] = fma(k.kyzm, dmu[2][1], sW);
        o.ms[S3_T] = k.kyy * pc[1]; o.ms[S3_DYE] = k.kyyd * g;
        o.mn[S3_U] = fma(-k.kxym, dmu[0][1], sU); o.mn[S3_V] = sV - tV; o.mn[S3_W] = fma(-k.kyzm, dmu[2][1], sW);
    }
    {   // Z
        const double sU = k.kzz * a[0], sV = k.kzz * a[1], sW = k.kzzm * a[2], sT = k.kzz * pc[2];
        const double tW = fma(k.kxz, dmu[0][0], k.kyz * dmu[1][1]);
        o.mb[S3_U] = fma(k.kxzm, dmu[0][2], sU); o.mb[S3_V] = fma(k.kyzm, dmu[1][2], sV); o.mb[S3_W] = sW + tW;
        o.mb[S3_T] = fma(k.kyz, pair[2], sT); o.mb[S3_DYE] = k.kzzd * g;
        o.mt[S3_U] = fma(-k.kxzm, dmu[0][2], sU); o.mt[S3_V] = fma(-k.kyzm, dmu[1][2], sV); o.mt[S3_W] = sW - tW;
        o.mt[S3_T] = fma(-k.kyz, pair[2], sT);
    }
    // ---- the cell's own part
    const double gU = fma(k.cUa3, a[0], div[E3_U]), gV = fma(k.cVb3, a[1], div[E3_V]), gW = fma(k.cWc3, a[2], div[E3_W]);
    const double gT = fma(k.kxx2, pc[0], fma(k.kyy2, pc[1], fma(k.kzz2, pc[2], k.cp * div[E3_T])));
    const double gD = fma(k.cDg3, g, div[E3_DYE]);
    const double rinv = fast_rcp(c[F3_RHO]);
    const double keep = select(act, 1.0, 0.0);
    o.coef[S3_U] = select(act & mask_of(c[F3_U] != 0.0), rinv, keep);
    o.coef[S3_V] = select(act & mask_of(c[F3_V] != 0.0), rinv, keep);
    o.coef[S3_W] = select(act & mask_of(c[F3_W] != 0.0), rinv, keep);
    o.coef[S3_T] = select(mask_of(c[F3_T] != 0.0), rinv * k.rcv, 1.0);
    o.coef[S3_DYE] = select(act & mask_of(c[F3_DYE] != 0.0), rinv, keep);
    o.q[S3_U] = fma(-o.coef[S3_U], gU, c[F3_U]);
    o.q[S3_V] = fma(-o.coef[S3_V], gV, c[F3_V]);
    o.q[S3_W] = fma(-o.coef[S3_W], gW, c[F3_W]);
    o.q[S3_T] = fma(-o.coef[S3_T], gT, c[F3_T]);
    o.q[S3_DYE] = fma(-o.coef[S3_DYE], gD, c[F3_DYE]);
    o.rho = c[F3_RHO] - div[E3_RHO];
}

CFD_HD void phaseA3s(const FastConsts& k, const double c[NF3], const double h[3][NF3], lane_mask act, StreamA3& o) {
    phaseA3s_by_direction(k, c, [&](int d, double hd[NF3]) { for (int q = 0; q < NF3; ++q) hd[q] = h[d][q]; }, act, o);
}

// q += coef * (message from the West neighbour); mwD: that message's Dye entry
CFD_HD void stream_west3(double q[NS3], const double coef[NS3], const double mw[4], double mwD) {
    q[S3_U] = fma(coef[S3_U], mw[S3_U], q[S3_U]);
    q[S3_V] = fma(coef[S3_V], mw[S3_V], q[S3_V]);
    q[S3_W] = fma(coef[S3_W], mw[S3_W], q[S3_W]);
    q[S3_T] = fma(coef[S3_T], mw[S3_T], q[S3_T]);
    q[S3_DYE] = fma(coef[S3_DYE], mwD, q[S3_DYE]);
}

// new value of one field: q (West folded in) plus the East, North, South, Top and Bottom messages
CFD_HD double stream_finish3(double q, double coef, double from_e, double from_n, double from_s, double from_t,
                             double from_b) {
    return fma(coef, (((from_e + from_n) + from_s) + from_t) + from_b, q);
}

}  // namespace cfd
