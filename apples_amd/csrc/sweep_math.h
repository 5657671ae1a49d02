// Arithmetic shared by the two forms of the least-squares sweep (sweep.hip: level loop over tree
// records; sweep_scan.hip: scan formulation over id-sorted leaves): tuples, lifting, the 2x2 solve, the
// residual with libm's pow bits, team-wide arg-min.  fp64, compiled with -ffp-contract=off; every
// expression is associated as the reference source line it cites.
#pragma once
#include "common.h"

#define WAVE 64
#define XE_STRIDE 18  // per-edge record: x_1, x_2, x_1_neg, x_2_neg, err, R[6], S[6], x_1-is-int  (HYBRID / inspection only)
#define INF_D __longlong_as_double(0x7ff0000000000000LL)

// tuple slots, reference attribute names per method
//   OLS/BME: 0 S   1 Sd    2 Sd2    3 SDd     4 SD2   5 SD      (apples/OLS.py:27-33, BME.py:11-17)
//   FM     : 0 S   1 Sd_D  2 Sd_D2  3 Sd2_D2  4 S1_D  5 S1_D2   (apples/FM.py:21-27)
//   BE     : 0 S   1 Sd    2 Sd_D   3 Sd2_D   4 SD    5 S1_D    (apples/BE.py:11-17)
template <int M>
__device__ __forceinline__ void leaf_tuple(double D, double *t) {
    t[0] = 1; t[1] = 0; t[2] = 0; t[3] = 0;
    if (M == APPLES_OLS || M == APPLES_BME) { t[4] = D * D; t[5] = D; }
    else if (M == APPLES_FM) { t[4] = 1.0 / D; t[5] = 1.0 / (D * D); }
    else { t[4] = D; t[5] = 1.0 / D; }
}

// what a parent adds for a child (or sibling, or its own R) tuple s over the edge e
template <int M>
__device__ __forceinline__ void lift(const double *s, double e, double *t) {
    if (M == APPLES_OLS || M == APPLES_BME) {  // apples/OLS.py:36-44
        t[0] = s[0];
        t[1] = s[0] * e + s[1];
        t[2] = s[0] * e * e + s[2] + 2 * e * s[1];
        t[3] = e * s[5] + s[3];
        t[4] = s[4];
        t[5] = s[5];
    } else if (M == APPLES_FM) {  // apples/FM.py:31-40
        t[0] = s[0];
        t[1] = e * s[4] + s[1];
        t[2] = e * s[5] + s[2];
        t[3] = s[5] * e * e + s[3] + 2 * e * s[2];
        t[4] = s[4];
        t[5] = s[5];
    } else {  // apples/BE.py:20-30
        t[0] = s[0];
        t[1] = s[0] * e + s[1];
        t[2] = e * s[5] + s[2];
        t[3] = s[5] * e * e + s[3] + 2 * e * s[2];
        t[4] = s[4];
        t[5] = s[5];
    }
}

#include "libm_pow_tables.inc"

// x ** 2 as the reference computes it: CPython's float_pow and numpy's scalar power both call libm's
// pow(x, 2.0), which is not correctly rounded (it differs from x*x in ~0.09 % of inputs).
// This is the x86-64 FMA build of GNU libm 2.35's pow specialised to y = 2 -- log of |x| in
// double-double from a 128-entry table, doubled, then exp from a 128-entry table -- with every fused
// multiply-add exactly where that build has it.  Verified bit for bit against libm on 1e8 inputs in the
// build container (the C oracle calls libm itself).  Outside [2^-500, 2^500] (and for 0, inf, nan) the
// square over/underflows and x*x gives the same result.
// `tab` = the two tables staged in LDS: [0, 384) log table rows {invc, logc, logctail}, then 256 exp-table words
__device__ __forceinline__ double pow2_libm(double x, const double *tab) {
    const double ax = fabs(x);
    if (!(ax >= 0x1p-500 && ax <= 0x1p500)) return x * x;
#ifdef SWEEP_EXP_PLAIN_SQ  // (timing experiment: plain squares, results differ in the last place: scripts/r05_plain_sq_exp.sh)
    return x * x;
#endif
    const unsigned long long ix = (unsigned long long)__double_as_longlong(ax);
    const unsigned long long tmp = ix - 0x3fe6955500000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((long long)tmp >> 52);
    const unsigned long long iz = ix - (tmp & (0xfffULL << 52));
    const double z = __longlong_as_double((long long)iz), kd = (double)k;
    const double invc = tab[3 * i], logc = tab[3 * i + 1], logctail = tab[3 * i + 2];
    const double r = fma(z, invc, -1.0);
    const double t1 = fma(kd, kPowLn2Hi, logc);
    const double t2 = t1 + r;
    const double lo1 = fma(kd, kPowLn2Lo, logctail);
    const double lo2 = t1 - t2 + r;
    const double ar = kPowLogPoly[0] * r, ar2 = r * ar, ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma(ar, r, -ar2);
    const double lo4 = t2 - hi + ar2;
    const double q1 = fma(r, kPowLogPoly[2], kPowLogPoly[1]);
    const double q2 = fma(r, kPowLogPoly[4], kPowLogPoly[3]);
    const double q3 = fma(r, kPowLogPoly[6], kPowLogPoly[5]);
    const double Q = fma(ar2, fma(ar2, q3, q2), q1);
    const double lo = fma(ar3, Q, lo1 + lo2 + lo3 + lo4);
    const double y = hi + lo;
    const double tail = hi - y + lo;
    const double ehi = y + y;
    const unsigned abstop = (unsigned)((unsigned long long)__double_as_longlong(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if ((int)(abstop - 0x3c9u) < 0) return 1.0 + ehi;
        return x * x;
    }
    double kd2 = fma(ehi, kExpInvLn2N, kExpShift);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd2);
    kd2 -= kExpShift;
    double rr = fma(kd2, kExpNegLn2LoN, fma(kd2, kExpNegLn2HiN, ehi));
    const double elo = fma(tail, 2.0, fma(y, 2.0, -ehi));
    rr = elo + rr;
    const unsigned idx = 2u * (unsigned)(ki & 127);
    const double tl = tab[384 + idx];
    const unsigned long long sbits = (unsigned long long)__double_as_longlong(tab[384 + idx + 1]) + (ki << 45);
    const double s0 = rr + tl;
    const double c23 = fma(rr, kExpPoly[1], kExpPoly[0]);
    const double r2 = rr * rr;
    const double c45 = fma(rr, kExpPoly[3], kExpPoly[2]);
    const double s1 = fma(c23, r2, s0);
    const double tmpd = fma(r2 * r2, c45, s1);
    const double scale = __longlong_as_double((long long)sbits);
    return fma(scale, tmpd, scale);
}

struct Sol {
    double x1, x2, x1n, x2n, err;
    int x1_int;
};

// placement_per_edge (apples/OLS.py:82-119 ..., util.py:6-54): x_1, x_2 of one edge; `err` is left unset
template <int M>
__device__ __forceinline__ Sol solve_x(const double *S, const double *R, double e, int negative) {
    // which tuple slots play which role (apples/OLS.py:90-96, FM.py:86-92, BE.py:61-67, BME.py:64-70)
    constexpr int IA = (M == APPLES_OLS || M == APPLES_BME) ? 0 : 5;                     // a_11 = R? + S?
    constexpr int IC = (M == APPLES_OLS || M == APPLES_BME) ? 5 : (M == APPLES_FM ? 4 : 0);  // RD / R1_D / R
    constexpr int IE = (M == APPLES_OLS || M == APPLES_BME) ? 0 : 5;                     // e * S / S1_D2 / S1_D
    constexpr int ID = (M == APPLES_OLS || M == APPLES_BME) ? 1 : 2;                     // Rd / Rd_D2 / Rd_D
    double a11 = R[IA] + S[IA];
    double a12 = R[IA] - S[IA];
    double a21 = a12, a22 = a11;
    double c1 = R[IC] + S[IC] - e * S[IE] - R[ID] - S[ID];
    double c2 = R[IC] - S[IC] + e * S[IE] - R[ID] + S[ID];
    // apples/util.py:26-50
    double det = 1 / (a11 * a22 - a12 * a21);
    Sol r;
    r.x1n = (a22 * c1 - a12 * c2) * det;
    r.x2n = (-a21 * c1 + a11 * c2) * det;
    r.x1 = r.x1n;
    r.x2 = r.x2n;
    r.x1_int = 0;
    r.err = 0;
    if (!negative) {
        // the branch table of apples/util.py:32-50, evaluated in its order with strict comparisons; the three branches that
        // divide all divide by a_11 (a_22 is the same number), so the quotient is formed once for whichever numerator the
        // lane's branch names -- the same IEEE division as in the branch, but one instruction sequence instead of three
        // when the lanes of a wavefront take different branches
        const double x1n = r.x1n, x2n = r.x2n;
        const bool n1 = x1n < 0, p1 = x1n > 0, n2 = x2n < 0, g2 = x2n > e, in2 = 0 <= x2n && x2n <= e;
        const int br = (n1 && n2) ? 1 : (p1 && n2) ? 2 : (n1 && in2) ? 3 : (n1 && g2) ? 4 : (p1 && g2) ? 5 : 0;
        if (br == 2 || br == 3 || br == 5) {
            const double num = br == 3 ? c2 * 1.0 : (br == 5 ? c1 * 1.0 - a12 * e : c1 * 1.0);
            const double qv = num / a11;
            if (br == 3) {
                r.x1 = 0; r.x1_int = 1;
                double u = qv;
                if (0 > u) u = 0;        // max(u, 0)
                r.x2 = (e < u) ? e : u;  // min(., e)
            } else {
                if (0 > qv) { r.x1 = 0; r.x1_int = 1; } else r.x1 = qv;  // max(t, 0)
                r.x2 = br == 2 ? 0 : e;
            }
        } else if (br == 1) {
            r.x1 = 0; r.x1_int = 1;
            r.x2 = 0;
        } else if (br == 4) {
            r.x1 = 0; r.x1_int = 1;
            r.x2 = e;
        }
    }
    return r;
}

// error_per_edge (apples/OLS.py:121-128, FM.py:117-124, BE.py:73-80, BME.py:76-83) at (x_1, x_2).  EXACT: the reference's bits
// (`x ** 2` is libm pow there: pow2_libm).  !EXACT: plain squares -- a bound, not a result: with *mag = the sum of the
// terms' magnitudes, |exact - this| <= 2^-49 mag (pow2_libm(x) and x * x are both within one unit in the last place of x^2: the
// two products of the C term differ by at most 2^-51 of their magnitudes + roundings, and the three additions behind C round
// partial sums that are bounded by mag: 6 x 2^-53 mag between the two evaluations); callers use 2^-40 mag.
template <int M, bool EXACT>
__device__ __forceinline__ double edge_residual(const double *S, const double *R, double e, double x1, double x2, const double *pow_tab,
                                                double *mag = nullptr) {
    constexpr int JA = (M == APPLES_FM) ? 0 : 4;
    constexpr int JB = (M == APPLES_OLS || M == APPLES_BME) ? 1 : 2;
    constexpr int JC = (M == APPLES_OLS || M == APPLES_BME) ? 0 : 5;
    constexpr int JD = (M == APPLES_OLS || M == APPLES_BME) ? 5 : (M == APPLES_FM ? 4 : 0);
    constexpr int JE = (M == APPLES_OLS || M == APPLES_BME) ? 3 : 1;
    constexpr int JF = (M == APPLES_OLS || M == APPLES_BME) ? 2 : 3;
    double up = x1 + x2;      // path through the parent side
    double dn = e + x1 - x2;  // path through the child side
    double A = R[JA] + S[JA];
    double B = 2 * up * R[JB] + 2 * dn * S[JB];
    // `x ** 2` is libm pow in the reference: same bits here
    const double cu = (EXACT ? pow2_libm(up, pow_tab) : up * up) * R[JC], cd = (EXACT ? pow2_libm(dn, pow_tab) : dn * dn) * S[JC];
    double C = cu + cd;
    double Dd = -2 * up * R[JD] - 2 * dn * S[JD];
    double E = -2 * R[JE] - 2 * S[JE];
    double F = R[JF] + S[JF];
    if (!EXACT && mag) *mag = fabs(A) + fabs(B) + fabs(cu) + fabs(cd) + fabs(Dd) + fabs(E) + fabs(F);
    return A + B + C + Dd + E + F;
}

template <int M>
__device__ __forceinline__ Sol solve_edge(const double *S, const double *R, double e, int negative, const double *pow_tab) {
    Sol r = solve_x<M>(S, R, e, negative);
    r.err = edge_residual<M, true>(S, R, e, r.x1, r.x2, pow_tab);
    return r;
}

__device__ __forceinline__ double shfl_down_f64s(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, delta, WAVE);
    hi = __shfl_down(hi, delta, WAVE);
    return __hiloint2double(hi, lo);
}

// team-wide lexicographic arg-min over (key, id); NaN keys never win.  A team is one wavefront
// (TEAM == 64: shuffles only) or the whole workgroup (TEAM == 256: shuffles + LDS).
template <int TEAM>
__device__ void team_argmin(double &d, int &i, double *shd, int *shi) {
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        double d2 = shfl_down_f64s(d, o);
        int i2 = __shfl_down(i, o, WAVE);
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
    if (TEAM == WAVE) {
        d = __hiloint2double(__shfl(__double2hiint(d), 0, WAVE), __shfl(__double2loint(d), 0, WAVE));
        i = __shfl(i, 0, WAVE);
        return;
    }
    int w = threadIdx.x / WAVE;
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) { shd[w] = d; shi[w] = i; }
    __syncthreads();
    d = shd[0]; i = shi[0];
    for (int k = 1; k < APPLES_TPB / WAVE; ++k) {
        double d2 = shd[k]; int i2 = shi[k];
        if (d2 < d || (d2 == d && i2 < i)) { d = d2; i = i2; }
    }
}

// Make the team's global/LDS writes visible to the whole team.  One wavefront executes its memory
// instructions in order, so a wavefront-scope fence is enough; a workgroup needs the barrier.
template <int TEAM>
__device__ __forceinline__ void team_sync() {
    if (TEAM == WAVE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    else __syncthreads();
}

