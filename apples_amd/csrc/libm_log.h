// log(x) with the bits of GNU libm 2.35's double-precision log on x86-64 with FMA (__log_fma: the variant the dynamic
// linker picks on every CPU with FMA3 + AVX2), for positive normal x -- what the C oracle calls, and what numpy's np.log
// (apples/distance.py:715,745) is wherever numpy has no SIMD form of its own.  The routine is glibc's
// sysdeps/ieee754/dbl-64/e_log.c (from Arm Optimized Routines): a 128-entry table {1/c, log c}, r = z/c - 1 by one fused
// multiply-add, a degree-5 polynomial; inputs within [1 - 2^-4, 1 + 0x1.09p-4) take a degree-11 polynomial of r = x - 1
// with a two-term head.  Restated operation for operation from the disassembly of the build container's libm.so.6 -- the
// compiler contracted most a * b + c of the C source into fused multiply-adds, and WHICH ones decides the last bit: every
// fma() below is one vfmadd of that code, every other operation a plain IEEE one (the build runs with -ffp-contract=off).
// Constants: apples_amd/data/libm_log_tables.txt (read from that libm).  Checked bit for bit against libm on 10^8 inputs
// on the CPU (uniform in (0,1), around 1 across the branch cut, random mantissas with exponents 2^-59..1, JC69-shaped
// arguments 1 - 4 m / (3 v)) and on the device by tests/test_gpu_parity.py.
#pragma once
#include <hip/hip_runtime.h>

#include "libm_log_tables.inc"

__device__ __forceinline__ double log_libm(double x) {
    const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) {  // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double *B = kLogB;
        const double r = x - 1.0;
        double a = fma(r, B[2], B[1]);
        double b = fma(r, B[5], B[4]);
        const double r2 = r * r;
        double c = fma(r, B[8], B[7]);
        a = fma(r2, B[3], a);
        b = fma(r2, B[6], b);
        const double r3 = r * r2;
        c = fma(r2, B[9], c);
        c = fma(r3, B[10], c);
        const double d = fma(c, r3, b);
        const double e = fma(d, r3, a);
        const double t = fma(r, 0x1p27, r);
        const double rhi = fma(-0x1p27, r, t);
        const double hh = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = fma(hh, B[0], r);
        const double t8 = r - hi;
        const double s = r + rhi;
        double lo = fma(hh, B[0], t8);
        const double m = B[0] * rlo;
        lo = fma(m, s, lo);
        const double y = fma(e, r3, lo);
        return hi + y;
    }
    const unsigned long long tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((long long)tmp >> 52);
    const unsigned long long iz = ix - (tmp & 0xfff0000000000000ull);
    const double2 tc = kLogTab[i];  // {1 / c, log c}
    const double *A = kLogA;
    const double z = __longlong_as_double((long long)iz);
    const double kd = (double)k;
    const double r = fma(z, tc.x, -1.0);
    const double w = fma(kd, kLogLn2Hi, tc.y);
    const double p1 = fma(r, A[2], A[1]);
    const double hi = r + w;
    const double r2 = r * r;
    double lo = (w - hi) + r;
    lo = fma(kd, kLogLn2Lo, lo);
    const double r3 = r * r2;
    const double q = fma(r, A[4], A[3]);
    lo = fma(r2, A[0], lo);
    const double p = fma(q, r2, p1);
    const double y = fma(r3, p, lo);
    return y + hi;
}
