// fp_rows.h -- which grid cells a footprint covers, and the LAST sample that falls into each, without walking the samples.
//
// Agent::draw (agent.cpp:285-327) maps every footprint sample p of a lattice of 0.01 m pitch (agent.cpp:18-62: x-major,
// 901 samples for r = 0.17) through world = T p, cell = round(world / res) (grid_map.cpp:40-44).  At 0.25 m cells those 901
// samples land on a dozen cells.  Along one lattice ROW (fixed x index, consecutive y indices) both cell indices are monotone
// in the y index -- every operation of the fp64 chain is monotone in its varying operand -- so a row is a short staircase of
// cells, and all that has to be found are the y indices at which it steps.  They are predicted from the real-arithmetic model
//     t(n) = ((m00 px + ox) + m01 (cy + 0.01 n)) / res            (and the same for the other axis)
// and CERTIFIED: the model differs from the reference's own rounding chain by a few ulps of the world coordinate (< 1e-11 m
// for maps of a kilometre), so whenever every sample next to a predicted step -- and both row ends -- is farther than
// FPR_EPS_W (1e-9 m) from the rounding boundary, the model's cells ARE the chain's cells.  A row that cannot be certified (a
// sample within a nanometre of a cell boundary: poses on round numbers, headings along an axis with the lattice exactly on a
// boundary) makes the caller fall back to the literal walk over all samples, which stays the definition.
// Host and device share this file: tests/host/fp_rows_check.cpp holds it to the literal walk on random and adversarial poses.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FPR_HD __host__ __device__ __forceinline__
#else
#define FPR_HD inline
#endif

#define FPR_MAXC 4         // steps per axis a row may take (rows of up to ~4 cells: classes beyond that keep the sample walk)
#define FPR_EPS_W 1e-9     // metres: certification margin around a rounding boundary
#define FPR_PITCH 0.01     // the reference's sample lattice (agent.cpp:19, 52)

// one lattice row of a footprint class: samples (px, cy + 0.01 n) for n = n_lo .. n_hi are the entries q0 .. q0 + (n_hi - n_lo)
// of the class's x-major sample list
struct FpRow {
    double px;
    int n_lo, n_hi, q0;
};

// what is the same for every row of one footprint at one pose
struct FpRowsPose {
    double m00, m01, m10, m11, ox, oy;  // T (base -> world), as Tf2
    double cy;                          // the lattice's y offset (sizes[1] of a circle, 0 for a rectangle / leg discs)
    double inv_res, eps_t;              // 1 / res; FPR_EPS_W in cell units (+ a relative term for large indices)
    double sx, sy, isx, isy;            // d t / d n per axis and its reciprocal (0 where the axis does not move along a row)
};

FPR_HD FpRowsPose fpr_pose(double m00, double m01, double m10, double m11, double ox, double oy, double cy, double res) {
    FpRowsPose p;
    p.m00 = m00; p.m01 = m01; p.m10 = m10; p.m11 = m11; p.ox = ox; p.oy = oy;
    p.cy = cy;
    p.inv_res = 1.0 / res;
    // (cell indices reach 2^24 at most: 2^-52 * 2^24 = 4e-9 relative to one cell would eat the margin on absurd maps, hence the second term)
    p.eps_t = FPR_EPS_W * p.inv_res + 64.0 * 2.220446049250313e-16 * (fabs(ox) + fabs(oy) + 1.0) * p.inv_res;
    p.sx = m01 * FPR_PITCH * p.inv_res;
    p.sy = m11 * FPR_PITCH * p.inv_res;
    p.isx = p.sx != 0.0 ? 1.0 / p.sx : 0.0;
    p.isy = p.sy != 0.0 ? 1.0 / p.sy : 0.0;
    return p;
}

// the staircase of one axis along one row
struct FpAxis {
    int c0, d, cnt;        // cell at n_lo, direction (+-1), number of steps
    int f[FPR_MAXC];       // first n of the cell behind step k (sentinel n_hi + 1 for k >= cnt)
};

// certified round-half-away-from-zero of a model value
FPR_HD int fpr_round(double t, double eps_t, bool& ok) {
    const double r = rint(t);
    ok &= (0.5 - fabs(t - r)) > eps_t;  // (an exact tie has distance 0: never certified)
    return (int)r;
}

FPR_HD FpAxis fpr_axis(double t0, double s, double is, int n_lo, int n_hi, double eps_t, bool& ok) {
    FpAxis a;
    const double t_lo = t0 + s * (double)n_lo, t_hi = t0 + s * (double)n_hi;
    a.c0 = fpr_round(t_lo, eps_t, ok);
    const int c1 = fpr_round(t_hi, eps_t, ok);
    a.d = c1 >= a.c0 ? 1 : -1;
    a.cnt = (c1 - a.c0) * a.d;
    if (a.cnt > FPR_MAXC) {
        ok = false;
        a.cnt = FPR_MAXC;
    }
    const double as = fabs(s);
#pragma unroll
    for (int k = 0; k < FPR_MAXC; k++) {
        a.f[k] = n_hi + 1;
        if (k < a.cnt) {
            const double h = (double)a.c0 + (double)a.d * ((double)k + 0.5);  // the rounding boundary between step k's two cells
            const double nu = (h - t0) * is;                                   // where the row crosses it, in sample indices
            const double fl = floor(nu), frac = nu - fl;
            ok &= fmin(frac, 1.0 - frac) * as > eps_t;                         // both neighbours are clear of the boundary
            const int f = (int)fl + 1;
            ok &= f > n_lo && f <= n_hi;
            a.f[k] = f;
        }
    }
    return a;
}

// Both axes of one row.  Returns false when the row could not be certified.
FPR_HD bool fpr_row(const FpRowsPose& p, const FpRow& r, FpAxis& ax, FpAxis& ay) {
    bool ok = true;
    const double tx0 = ((p.m00 * r.px + p.ox) + p.m01 * p.cy) * p.inv_res;
    const double ty0 = ((p.m10 * r.px + p.oy) + p.m11 * p.cy) * p.inv_res;
    ax = fpr_axis(tx0, p.sx, p.isx, r.n_lo, r.n_hi, p.eps_t, ok);
    ay = fpr_axis(ty0, p.sy, p.isy, r.n_lo, r.n_hi, p.eps_t, ok);
    return ok;
}

// the pieces of a row: piece (i, j), i <= ax.cnt, j <= ay.cnt, holds the samples whose x cell is step i's and whose y cell is
// step j's.  Returns whether it is non-empty; then (m, n) is its cell and `last` the index + 1 of its last sample in the list.
FPR_HD bool fpr_piece(const FpRow& r, const FpAxis& ax, const FpAxis& ay, int i, int j, int& m, int& n, uint32_t& last) {
    // (i, j are compile-time constants in the callers' unrolled loops: the selects below fold)
    const int sx = i == 0 ? r.n_lo : ax.f[i - 1], ex = i == FPR_MAXC ? r.n_hi : ax.f[i] - 1;
    const int sy = j == 0 ? r.n_lo : ay.f[j - 1], ey = j == FPR_MAXC ? r.n_hi : ay.f[j] - 1;
    const int s = sx > sy ? sx : sy, e = ex < ey ? ex : ey;
    m = ax.c0 + ax.d * i;
    n = ay.c0 + ay.d * j;
    last = (uint32_t)(r.q0 + (e - r.n_lo) + 1);
    return i <= ax.cnt && j <= ay.cnt && s <= e;
}
