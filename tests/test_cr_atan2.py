"""img_env_amd/csrc/cr_atan2.h (double-double atan2, rounded once) against the host libm: it must be within one
ulp everywhere and agree bit-for-bit with glibc in all but the rare inputs where glibc itself misrounds -- and agree
with an independent correctly rounded route (libquadmath's 113-bit atan2q rounded once to double, the evaluation the
oracle's CR mode `sfm_set_cr_atan2(1)` uses) on EVERY input."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <random>
#include <quadmath.h>
#include "%s/img_env_amd/csrc/cr_atan2.h"
static int64_t ord(double d) { int64_t i; memcpy(&i, &d, 8); return i < 0 ? INT64_MIN - i : i; }
int main() {
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(-20, 20);
    long bad = 0, far = 0, notcr = 0, n = 4000000;
    for (long i = 0; i < n; i++) {
        double y = u(g), x = u(g);
        if (i %% 5 == 0) x = y * (1 + 2.2e-16 * (double)(i %% 7));   /* nearly parallel to the diagonal */
        if (i %% 11 == 0) y = 0.0;
        if (i %% 13 == 0) x = -0.0;
        const double a = cr_atan2(y, x), b = atan2(y, x);
        if (a != b) { bad++; if (llabs(ord(a) - ord(b)) > 1) far++; }
        const double q = (double)atan2q((__float128)y, (__float128)x);
        if (a != q && !(a == 0 && q == 0)) notcr++;
    }
    printf("%%ld %%ld %%ld %%ld\n", n, bad, far, notcr);
    return 0;
}
'''


def test_cr_atan2_agrees_with_libm():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC % ROOT)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O2", "-std=gnu++17", "-ffp-contract=off", src, "-o", exe, "-lquadmath"])
        n, bad, far, notcr = map(int, subprocess.check_output([exe]).split())
    assert far == 0                      # never more than one ulp from glibc
    assert bad / n < 3e-3                # glibc 2.35 misrounds ~0.1 % of inputs (errors 0.500x ulp)
    assert notcr == 0                    # correctly rounded: identical to the 113-bit evaluation rounded once
