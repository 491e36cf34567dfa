/*
 * oracle_resize.c -- TEST INFRASTRUCTURE (oracle), never linked into the product library.
 *
 * CPU restatement of the two OpenCV resizes on the reference's step() path, for single-channel 8-bit images:
 *   cv::resize(image, map_, Size(w, h))                      INTER_LINEAR   grid_map.cpp:28-38   (map load)
 *   cv2.resize(view, (48, 48), interpolation=INTER_CUBIC)    INTER_CUBIC    yaml_env.py:431-438  (sensor_map)
 *
 * OpenCV is a THIRD-PARTY dependency that is not under /root/reference and not in this image (Ubuntu 20.04 / ROS noetic
 * ship OpenCV 4.2.0).  What follows restates the published algorithm of 4.2.0's generic CPU path,
 * modules/imgproc/src/resize.cpp (no IPP, no OpenCL, no FMA: the Debian build's baseline is SSE3):
 *
 *   cv::resize            dsize given => inv_scale = (double)dsize / ssize; equal sizes => plain copy
 *   hal::resize           scale = 1. / inv_scale; per destination column  fx = (float)((dx + 0.5) * scale_x - 0.5),
 *                         sx = cvFloor(fx), fx -= sx; rows likewise; 8-bit images take the fixed-point path with
 *                         coefficients saturate_cast<short>(c * 2048)  (INTER_RESIZE_COEF_BITS = 11, cvRound = half to even)
 *   interpolateCubic      A = -0.75, float32 arithmetic
 *   linear                cbuf = {1 - fx, fx}; sx < 0 => (fx, sx) = (0, 0); sx >= width - 1 => (0, width - 1)
 *   resizeGeneric_Invoker source rows sy0 - ksize/2 + 1 + k clipped to [0, h - 1]
 *   HResizeLinear / HResizeCubic   int D[dx] = sum_j S[clamp(sx - (ksize/2 - 1) + j)] * alpha[j]   (border columns replicate)
 *   VResizeLinear<uchar>  dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)   (vector and scalar alike)
 *   VResizeCubicVec_32s8u columns x < (width / 8) * 8: float32, t = S3 * b3; t = S2 * b2 + t; t = S1 * b1 + t; t = S0 * b0 + t
 *                         with b_k = beta_k * (1.f / (2048 * 2048)), v_round (half to even), saturate to uchar
 *   VResizeCubic (tail)   remaining columns: saturate_cast<uchar>((S0 b0 + S1 b1 + S2 b2 + S3 b3 + (1 << 21)) >> 22)
 *
 * PARITY UNPINNED: neither the reference nor this image holds a vector for these two calls.  tests/test_oracle_resize.py
 * pins the restatement on hand-derived cases (constant images, identity sizes, a unit step, a single bright pixel, the
 * published coefficient values); the HIP / host implementations (img_env_amd/csrc/cv_resize.h, k_resize_cubic) are written
 * independently and compared with this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define COEF_BITS 11
#define COEF_SCALE (1 << COEF_BITS)

static int cv_floor(double v) { return (int)floor(v); }
static short sat_short_round(float v) { /* saturate_cast<short>(float): cvRound (nearest, ties to even), then clamp */
    long r = lrintf(v);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void interpolate_cubic(float x, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* per destination index: source offset and the ksize fixed-point coefficients */
static void axis_tables(int ssize, int dsize, int cubic, int* ofs, short* coef) {
    const int ksize = cubic ? 4 : 2;
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (!cubic) {
            if (s < 0) {
                f = 0;
                s = 0;
            }
            if (s >= ssize - 1) {
                f = 0;
                s = ssize - 1;
            }
        }
        ofs[d] = s;
        float cbuf[4];
        if (cubic)
            interpolate_cubic(f, cbuf);
        else {
            cbuf[0] = 1.f - f;
            cbuf[1] = f;
        }
        for (int k = 0; k < ksize; k++) coef[d * ksize + k] = sat_short_round(cbuf[k] * COEF_SCALE);
    }
}

static void hresize_row(const uint8_t* S, int swidth, int dwidth, const int* xofs, const short* alpha, int ksize, int* D) {
    for (int dx = 0; dx < dwidth; dx++) {
        int v = 0;
        for (int j = 0; j < ksize; j++) {
            const int sxj = clampi(xofs[dx] - (ksize / 2 - 1) + j, 0, swidth - 1);
            v += S[sxj] * alpha[dx * ksize + j];
        }
        D[dx] = v;
    }
}

/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) for CV_8UC1 */
void oracle_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    if (sh == dh && sw == dw) {
        memcpy(dst, src, (size_t)sh * sw);
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* alpha = (short*)malloc(sizeof(short) * 2 * dw);
    short* beta = (short*)malloc(sizeof(short) * 2 * dh);
    int* r0 = (int*)malloc(sizeof(int) * dw);
    int* r1 = (int*)malloc(sizeof(int) * dw);
    axis_tables(sw, dw, 0, xofs, alpha);
    axis_tables(sh, dh, 0, yofs, beta);
    /* the row offsets keep the unclamped floor for rows (hal::resize clamps fx / sx for columns only): redo them */
    {
        const double scale = 1. / ((double)dh / sh);
        for (int d = 0; d < dh; d++) {
            float f = (float)((d + 0.5) * scale - 0.5);
            int s = cv_floor(f);
            f -= s;
            yofs[d] = s;
            beta[2 * d] = sat_short_round((1.f - f) * COEF_SCALE);
            beta[2 * d + 1] = sat_short_round(f * COEF_SCALE);
        }
    }
    for (int dy = 0; dy < dh; dy++) {
        hresize_row(src + (size_t)clampi(yofs[dy], 0, sh - 1) * sw, sw, dw, xofs, alpha, 2, r0);
        hresize_row(src + (size_t)clampi(yofs[dy] + 1, 0, sh - 1) * sw, sw, dw, xofs, alpha, 2, r1);
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        for (int x = 0; x < dw; x++)
            dst[(size_t)dy * dw + x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(r0); free(r1);
}

static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* cv2.resize(src, (dw, dh), interpolation=cv2.INTER_CUBIC) for uint8, one channel */
void oracle_resize_cubic_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    if (sh == dh && sw == dw) {
        memcpy(dst, src, (size_t)sh * sw);
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* alpha = (short*)malloc(sizeof(short) * 4 * dw);
    short* beta = (short*)malloc(sizeof(short) * 4 * dh);
    int* rows[4];
    for (int k = 0; k < 4; k++) rows[k] = (int*)malloc(sizeof(int) * dw);
    axis_tables(sw, dw, 1, xofs, alpha);
    axis_tables(sh, dh, 1, yofs, beta);
    const float scale = 1.f / (COEF_SCALE * COEF_SCALE);
    const int vec_end = (dw / 8) * 8; /* v_uint16::nlanes = 8 (128-bit universal intrinsics) */
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 4; k++) hresize_row(src + (size_t)clampi(yofs[dy] - 1 + k, 0, sh - 1) * sw, sw, dw, xofs, alpha, 4, rows[k]);
        const short* b = beta + 4 * dy;
        const float b0 = b[0] * scale, b1 = b[1] * scale, b2 = b[2] * scale, b3 = b[3] * scale;
        for (int x = 0; x < dw; x++) {
            if (x < vec_end) {
                float t = (float)rows[3][x] * b3;
                t = (float)rows[2][x] * b2 + t;
                t = (float)rows[1][x] * b1 + t;
                t = (float)rows[0][x] * b0 + t;
                dst[(size_t)dy * dw + x] = sat_u8((int)lrintf(t));
            } else {
                const int v = rows[0][x] * b[0] + rows[1][x] * b[1] + rows[2][x] * b[2] + rows[3][x] * b[3];
                dst[(size_t)dy * dw + x] = sat_u8((v + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS));
            }
        }
    }
    free(xofs); free(yofs); free(alpha); free(beta);
    for (int k = 0; k < 4; k++) free(rows[k]);
}
