/*
 * tfmath.h -- TEST INFRASTRUCTURE (oracle).  Planar restatement of the ROS `tf` LinearMath
 * (Bullet) operations the reference hot path calls.  tf is a third-party dependency that is
 * NOT vendored under /root/reference (ROS noetic, tf 1.13.x, tf/LinearMath/{Quaternion,
 * Matrix3x3,Transform}.h); the published formulas are restated here in the same operation
 * order, in fp64, for the planar case (roll = pitch = 0, z = 0) in which the dropped terms are
 * exact zeros.  Call sites in the reference: agent.cpp:84-88, 92-131, 138-141, 149-153,
 * 159-179, 295, 372, 387-388; img_env.cpp:180-183, 230-234, 264-267, 570-576.
 *
 * PARITY UNPINNED for this file: the reference has no tests and tf cannot be built here;
 * known-answer checks in tests/test_oracle_tf.py pin it against hand-derived values only.
 *
 * Must be compiled with -ffp-contract=off (the reference is built without FMA contraction).
 */
#ifndef ORACLE_TFMATH_H_
#define ORACLE_TFMATH_H_

#include <math.h>

typedef struct tf2d {
    double m00, m01, m10, m11; /* basis rows (m02 = m12 = m20 = m21 = 0, m22 = 1) */
    double ox, oy;             /* origin */
} tf2d;

/* Quaternion::setRPY(0,0,yaw) -> (0, 0, sin(yaw/2), cos(yaw/2)); then
 * Matrix3x3::setRotation(q): d = |q|^2, s = 2/d, zs = z*s, wz = w*zs, zz = z*zs. */
static inline void tf_set_rotation_zw(tf2d* t, double z, double w) {
    double d = z * z + w * w;
    double s = 2.0 / d;
    double zs = z * s;
    double wz = w * zs;
    double zz = z * zs;
    t->m00 = 1.0 - zz;
    t->m01 = -wz;
    t->m10 = wz;
    t->m11 = 1.0 - zz;
}

static inline void tf_set_yaw(tf2d* t, double yaw) {
    double half = yaw * 0.5;
    tf_set_rotation_zw(t, sin(half), cos(half));
}

static inline tf2d tf_from_pose(double x, double y, double yaw) {
    tf2d t;
    tf_set_yaw(&t, yaw);
    t.ox = x;
    t.oy = y;
    return t;
}

/* Transform::operator()(v): (row0.v + ox, row1.v + oy) */
static inline void tf_apply(const tf2d* t, double x, double y, double* ox, double* oy) {
    *ox = (t->m00 * x + t->m01 * y) + t->ox;
    *oy = (t->m10 * x + t->m11 * y) + t->oy;
}

/* Transform::inverse(): inv = basis^T, origin = inv * (-o) */
static inline tf2d tf_inverse(const tf2d* t) {
    tf2d r;
    double nx = -t->ox, ny = -t->oy;
    r.m00 = t->m00;
    r.m01 = t->m10;
    r.m10 = t->m01;
    r.m11 = t->m11;
    r.ox = r.m00 * nx + r.m01 * ny;
    r.oy = r.m10 * nx + r.m11 * ny;
    return r;
}

/* Transform::operator*: basis = A.basis*B.basis (Matrix3x3 operator*: element (i,j) =
 * B.tdot_j(A[i]) = B0j*Ai0 + B1j*Ai1 + B2j*Ai2), origin = A(B.origin) */
static inline tf2d tf_mul(const tf2d* a, const tf2d* b) {
    tf2d r;
    r.m00 = b->m00 * a->m00 + b->m10 * a->m01;
    r.m01 = b->m01 * a->m00 + b->m11 * a->m01;
    r.m10 = b->m00 * a->m10 + b->m10 * a->m11;
    r.m11 = b->m01 * a->m10 + b->m11 * a->m11;
    tf_apply(a, b->ox, b->oy, &r.ox, &r.oy);
    return r;
}

/* Matrix3x3::getRotation (trace branch / i==2 branch) followed by Matrix3x3(q).getRPY yaw
 * (getEulerYPR solution 1: pitch = -asin(m20) = -0, yaw = atan2(m10/cos(pitch), m00/cos(pitch))).
 * agent.cpp:165-168. */
static inline double tf_basis_yaw_via_quaternion(const tf2d* t) {
    double trace = t->m00 + t->m11 + 1.0;
    double qz, qw;
    if (trace > 0.0) {
        double s = sqrt(trace + 1.0);
        qw = s * 0.5;
        s = 0.5 / s;
        qz = (t->m10 - t->m01) * s;
    } else {
        /* largest diagonal element is m22 = 1 (i = 2, j = 0, k = 1) */
        double s = sqrt(1.0 - t->m00 - t->m11 + 1.0);
        qz = s * 0.5;
        s = 0.5 / s;
        qw = (t->m10 - t->m01) * s;
    }
    tf2d r;
    tf_set_rotation_zw(&r, qz, qw);
    return atan2(r.m10 / 1.0, r.m00 / 1.0);
}

/* tf::Matrix3x3(q).getRPY for a planar quaternion (0,0,qz,qw): img_env.cpp:180-183 */
static inline double tf_yaw_from_quaternion_zw(double qz, double qw) {
    tf2d r;
    tf_set_rotation_zw(&r, qz, qw);
    return atan2(r.m10 / 1.0, r.m00 / 1.0);
}

#endif
