/*
 * oracle_resize.h -- TEST INFRASTRUCTURE (oracle).  The two OpenCV resizes of the reference's step() path restated for
 * one-channel 8-bit images (see oracle_resize.c for what is restated and from where).
 */
#ifndef ORACLE_RESIZE_H_
#define ORACLE_RESIZE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* cv::resize(src, dst, Size(dw, dh)) -- INTER_LINEAR (grid_map.cpp:28-38) */
void oracle_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
/* cv2.resize(src, (dw, dh), interpolation=cv2.INTER_CUBIC) (yaml_env.py:431-438) */
void oracle_resize_cubic_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
#ifdef __cplusplus
}
#endif
#endif
