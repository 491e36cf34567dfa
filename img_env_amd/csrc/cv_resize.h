// cv_resize.h -- the two OpenCV resizes of the step() path for one-channel 8-bit images.
//
//   GridMap::read_image  cv::resize(image, map_, Size(w, h))            INTER_LINEAR   grid_map.cpp:28-38   (host, at create)
//   _trans_cv2_sensor_map  cv2.resize(view, image_size, INTER_CUBIC)    INTER_CUBIC    yaml_env.py:431-438  (device, every step)
//
// OpenCV is third-party (4.2.0 on the reference's Ubuntu 20.04 / ROS noetic) and not under the reference tree; this is the
// published algorithm of its generic CPU path for CV_8UC1 (modules/imgproc/src/resize.cpp: hal::resize, interpolateCubic,
// HResizeLinear / HResizeCubic, VResizeLinear<uchar>, VResizeCubicVec_32s8u + VResizeCubic tail), built without IPP / FMA:
//   * dsize given => inv_scale = (double)dsize / ssize, scale = 1. / inv_scale; equal sizes are a copy;
//   * destination index d looks at f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s;
//   * 8-bit images use 11-bit fixed-point coefficients saturate_cast<short>(c * 2048) (nearest, ties to even);
//   * linear: c = (1 - f, f); COLUMNS before the first / behind the last pixel use that pixel alone; the vertical pass is
//     uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
//   * cubic: A = -0.75 in float32; borders replicate; the vertical pass of the first (width / 8) * 8 columns runs in float32
//     (t = S3 b3; t = S2 b2 + t; t = S1 b1 + t; t = S0 b0 + t with b = beta / 2^22, rounded half to even, saturated), the
//     remaining columns in integers ((sum + 2^21) >> 22, saturated);
//   * source rows outside the image are clipped to its first / last row.
// The axis tables (offset + coefficients per destination index) are built on the host for both uses; the device kernel
// (k_resize_cubic in kernels.h) only runs the two passes.  Parity with OpenCV itself is unpinned (see oracle/oracle_resize.c).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

struct CvAxis {           // one axis of a resize: where each destination index reads and with which weights
    int ksize = 0;        // 2 linear, 4 cubic
    std::vector<int> ofs; // source index of tap (ksize / 2 - 1), unclipped for cubic
    std::vector<short> coef;
};

static inline short cv_coef(float c) {  // saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE)
    long r = lrintf(c * 2048.0f);
    return (short)(r > 32767 ? 32767 : (r < -32768 ? -32768 : r));
}

static inline CvAxis cv_axis(int ssize, int dsize, bool cubic, bool clamp_linear_columns) {
    CvAxis a;
    a.ksize = cubic ? 4 : 2;
    a.ofs.resize(dsize);
    a.coef.resize((size_t)dsize * a.ksize);
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (!cubic && clamp_linear_columns) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        a.ofs[d] = s;
        short* c = &a.coef[(size_t)d * a.ksize];
        if (cubic) {
            const float A = -0.75f;
            const float c0 = ((A * (f + 1) - 5 * A) * (f + 1) + 8 * A) * (f + 1) - 4 * A;
            const float c1 = ((A + 2) * f - (A + 3)) * f * f + 1;
            const float c2 = ((A + 2) * (1 - f) - (A + 3)) * (1 - f) * (1 - f) + 1;
            const float c3 = 1.f - c0 - c1 - c2;
            c[0] = cv_coef(c0); c[1] = cv_coef(c1); c[2] = cv_coef(c2); c[3] = cv_coef(c3);
        } else {
            c[0] = cv_coef(1.f - f);
            c[1] = cv_coef(f);
        }
    }
    return a;
}

static inline int cv_clip(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// horizontal pass of one source row into 32-bit sums
static inline void cv_hpass(const uint8_t* S, int sw, const CvAxis& ax, int dw, int* D) {
    const int first = ax.ksize / 2 - 1;
    for (int dx = 0; dx < dw; dx++) {
        const short* c = &ax.coef[(size_t)dx * ax.ksize];
        int v = 0;
        for (int j = 0; j < ax.ksize; j++) v += S[cv_clip(ax.ofs[dx] - first + j, sw)] * c[j];
        D[dx] = v;
    }
}

static inline uint8_t cv_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// the vertical cubic pass for one output value (shared by the host routine below and the device kernel's restatement)
static inline uint8_t cv_vcubic(const int s[4], const short b[4], bool vector_column) {
    if (vector_column) {
        const float scale = 1.f / (2048 * 2048);
        float t = (float)s[3] * (b[3] * scale);
        t = (float)s[2] * (b[2] * scale) + t;
        t = (float)s[1] * (b[1] * scale) + t;
        t = (float)s[0] * (b[0] * scale) + t;
        return cv_sat_u8((int)lrintf(t));
    }
    const int v = s[0] * b[0] + s[1] * b[1] + s[2] * b[2] + s[3] * b[3];
    return cv_sat_u8((v + (1 << 21)) >> 22);
}

static inline void cv_resize_u8(bool cubic, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    if (sh == dh && sw == dw) {
        memcpy(dst, src, (size_t)sh * sw);
        return;
    }
    const CvAxis ax = cv_axis(sw, dw, cubic, true), ay = cv_axis(sh, dh, cubic, false);
    const int k = ax.ksize;
    std::vector<int> rows((size_t)k * dw);
    for (int dy = 0; dy < dh; dy++) {
        for (int j = 0; j < k; j++) cv_hpass(src + (size_t)cv_clip(ay.ofs[dy] - (k / 2 - 1) + j, sh) * sw, sw, ax, dw, &rows[(size_t)j * dw]);
        const short* b = &ay.coef[(size_t)dy * k];
        for (int x = 0; x < dw; x++) {
            if (cubic) {
                const int s[4] = {rows[x], rows[dw + x], rows[2 * (size_t)dw + x], rows[3 * (size_t)dw + x]};
                dst[(size_t)dy * dw + x] = cv_vcubic(s, b, x < (dw / 8) * 8);
            } else {
                dst[(size_t)dy * dw + x] = (uint8_t)((((b[0] * (rows[x] >> 4)) >> 16) + ((b[1] * (rows[dw + x] >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
}
