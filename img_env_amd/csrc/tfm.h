// tfm.h -- planar rigid transforms in the operation order of ROS tf / Bullet LinearMath, fp64,
// usable from host and device code.  The reference calls tf::Transform / Matrix3x3 / Quaternion
// at agent.cpp:84-131, 138-179, 295, 372, 387-388 and img_env.cpp:180-183, 570-576; tf itself is a
// third-party ROS package (noetic, tf 1.13.x, tf/LinearMath/*.h) that is not vendored in the
// reference tree, so its published formulas are restated here for roll = pitch = 0, z = 0, where
// every dropped term is an exact zero.  Compile with -ffp-contract=off: the reference is built
// without FMA contraction and cell indices come from round() of these values.
#pragma once
#include <math.h>

#include "cr_atan2.h"

#if defined(__HIPCC__)
#define TFM_HD __host__ __device__ __forceinline__
#else
#define TFM_HD inline
#endif

struct Tf2 {
    double m00, m01, m10, m11;  // basis rows
    double ox, oy;              // origin
};

// Matrix3x3::setRotation(Quaternion(0, 0, z, w))
TFM_HD void tf_set_rotation_zw(Tf2& t, double z, double w) {
    const double d = z * z + w * w;
    const double s = 2.0 / d;
    const double zs = z * s;
    const double wz = w * zs;
    const double zz = z * zs;
    t.m00 = 1.0 - zz;
    t.m01 = -wz;
    t.m10 = wz;
    t.m11 = 1.0 - zz;
}

// Quaternion::setRPY(0, 0, yaw) then setRotation
TFM_HD Tf2 tf_from_pose(double x, double y, double yaw) {
    Tf2 t;
    const double half = yaw * 0.5;
    tf_set_rotation_zw(t, sin(half), cos(half));
    t.ox = x;
    t.oy = y;
    return t;
}

// same, from the cached sin(yaw/2), cos(yaw/2) of a robot record
TFM_HD Tf2 tf_from_pose_sc(double x, double y, double sh, double ch) {
    Tf2 t;
    tf_set_rotation_zw(t, sh, ch);
    t.ox = x;
    t.oy = y;
    return t;
}

// Transform::operator()(Vector3)
TFM_HD void tf_apply(const Tf2& t, double x, double y, double& rx, double& ry) {
    rx = (t.m00 * x + t.m01 * y) + t.ox;
    ry = (t.m10 * x + t.m11 * y) + t.oy;
}

// Transform::inverse()
TFM_HD Tf2 tf_inverse(const Tf2& t) {
    Tf2 r;
    const double nx = -t.ox, ny = -t.oy;
    r.m00 = t.m00;
    r.m01 = t.m10;
    r.m10 = t.m01;
    r.m11 = t.m11;
    r.ox = r.m00 * nx + r.m01 * ny;
    r.oy = r.m10 * nx + r.m11 * ny;
    return r;
}

// Transform::operator*(Transform)
TFM_HD Tf2 tf_mul(const Tf2& a, const Tf2& b) {
    Tf2 r;
    r.m00 = b.m00 * a.m00 + b.m10 * a.m01;
    r.m01 = b.m01 * a.m00 + b.m11 * a.m01;
    r.m10 = b.m00 * a.m10 + b.m10 * a.m11;
    r.m11 = b.m01 * a.m10 + b.m11 * a.m11;
    tf_apply(a, b.ox, b.oy, r.ox, r.oy);
    return r;
}

// Matrix3x3::getRotation -> Matrix3x3(q).getRPY yaw (agent.cpp:165-168)
TFM_HD double tf_basis_yaw_via_quaternion(const Tf2& t) {
    const double trace = t.m00 + t.m11 + 1.0;
    double qz, qw;
    if (trace > 0.0) {
        double s = sqrt(trace + 1.0);
        qw = s * 0.5;
        s = 0.5 / s;
        qz = (t.m10 - t.m01) * s;
    } else {
        double s = sqrt(1.0 - t.m00 - t.m11 + 1.0);
        qz = s * 0.5;
        s = 0.5 / s;
        qw = (t.m10 - t.m01) * s;
    }
    Tf2 r;
    tf_set_rotation_zw(r, qz, qw);
#if defined(__HIP_DEVICE_COMPILE__)
    return cr_atan2(r.m10 / 1.0, r.m00 / 1.0);  // rounds like glibc's atan2 (cr_atan2.h); the host keeps libm
#else
    return atan2(r.m10 / 1.0, r.m00 / 1.0);
#endif
}

// tf::Matrix3x3(q).getRPY for a planar quaternion (img_env.cpp:180-183)
TFM_HD double tf_yaw_from_quaternion_zw(double qz, double qw) {
    Tf2 r;
    tf_set_rotation_zw(r, qz, qw);
    return atan2(r.m10 / 1.0, r.m00 / 1.0);
}

// C round() (half away from zero) as an int.  On the device: round-to-nearest-even (one instruction) is the
// same integer except on exact .5 ties, which take the rare second branch.
TFM_HD int round_away_i(double t) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = rint(t);
    if (__builtin_expect(fabs(t - r) == 0.5, 0)) r = t + copysign(0.5, t);  // t - r is exact
    return (int)r;
#else
    return (int)round(t);
#endif
}
// GridMap::world2map (grid_map.cpp:40-44): C round(), half away from zero
TFM_HD int w2m(double v, double res) { return round_away_i(v / res); }
#if defined(__HIPCC__)
// Device rounding in two steps: round-to-nearest-even now, and a flag that says whether any of the values was
// an exact .5 tie (where C round() differs); callers test the flag once per group of cells and only then
// redo the group with round_tie_fix.  Keeps the hot loops free of per-value branches.
__device__ __forceinline__ int round_even_i(double t, bool& tie) {
    const double r = rint(t);
    tie |= fabs(t - r) == 0.5;  // t - r is exact
    return (int)r;
}
__device__ __forceinline__ int round_tie_fix(double t) {
    const double r = rint(t);
    return (int)(fabs(t - r) == 0.5 ? t + copysign(0.5, t) : r);
}
#endif
// When the resolution is a power of two, v * (1/res) and v / res are the same exact scaling, so the
// (much cheaper) multiply is bit-identical to the reference's division.
template <bool POW2>
TFM_HD double w2m_scale(double v, double res, double inv_res) {
    return POW2 ? v * inv_res : v / res;
}
template <bool POW2>
TFM_HD int w2m_t(double v, double res, double inv_res) {
    return round_away_i(w2m_scale<POW2>(v, res, inv_res));
}
#if defined(__HIPCC__)
// both cell indices of a world point, one (rarely taken) branch for the pair
template <bool POW2>
__device__ __forceinline__ void w2m_pair(double wx, double wy, double res, double inv_res, int& m, int& n) {
    if (POW2) {
        const double tx = wx * inv_res, ty = wy * inv_res;
        bool tie = false;
        m = round_even_i(tx, tie);
        n = round_even_i(ty, tie);
        if (__builtin_expect(tie, 0)) {
            m = round_tie_fix(tx);
            n = round_tie_fix(ty);
        }
    } else {
        // The reference divides (grid_map.cpp:40-44), and a float64 division is ~22 instructions on this chip.  v * (1 / res)
        // differs from the correctly rounded v / res by at most 1.5 * 2^-52 |v / res| (two roundings against one), so both round
        // to the same integer unless the product lies within that distance of a half-integer; the test below leaves a margin of
        // 8 x (2^-49 |t|) and sends those values -- exact .5 ties included -- through the reference's own division.
        const double tx = wx * inv_res, ty = wy * inv_res;
        const double rx = rint(tx), ry = rint(ty);
        const double ex = fabs(fabs(tx - rx) - 0.5), ey = fabs(fabs(ty - ry) - 0.5);  // distance to the nearest half-integer (t - r is exact)
        m = (int)rx;
        n = (int)ry;
        if (__builtin_expect(ex <= fabs(tx) * 0x1p-49 || ey <= fabs(ty) * 0x1p-49, 0)) {
            m = round_tie_fix(wx / res);
            n = round_tie_fix(wy / res);
        }
    }
}
#endif
