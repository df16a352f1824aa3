/*
 * pais_pyramid.h -- C ABI of the camera pyramid construction on the MI355X: the load-time step right
 * before the hot path, PAIS::Camera's constructor (TMVS/mvs/camera.cpp:45-136):
 *   maxLOD  = min((int)(ln max(W,H) / ln(1/lodRatio)), cfg.maxLOD)                      (camera.cpp:63-64)
 *   level i = resize(level 0, Size(), s, s, INTER_AREA), s = lodRatio^i                  (camera.cpp:85)
 *   edge  i = sqrt(Sx^2 + Sy^2) of level i with Sobel(ksize = 1), min-max normalised     (camera.cpp:72-77, 87-91)
 * The arithmetic is the one of the host restatement pais_mvs_amd/camera.py (area weights in double, separable,
 * rows then columns, ascending source index; round-half-even to uchar): parity is bit for bit against it; against
 * OpenCV 2.4.2 itself it is unpinned (not in the image).
 */
#ifndef PAIS_PYRAMID_H
#define PAIS_PYRAMID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAIS_PYRAMID_MAX_LEVELS 16

typedef struct pais_pyramid {
    int      max_lod;                              /* levels 0 .. max_lod                         */
    int      width[PAIS_PYRAMID_MAX_LEVELS];
    int      height[PAIS_PYRAMID_MAX_LEVELS];
    uint8_t *image[PAIS_PYRAMID_MAX_LEVELS];       /* host, row-major, stride == width            */
    double  *edge[PAIS_PYRAMID_MAX_LEVELS];        /* host, NULL unless build_edges               */
    double   kernel_ms;                            /* device time of all kernels of this build    */
} pais_pyramid;

/* Camera::Camera's pyramid + edge maps for one image, computed on `device`.  level0: W x H uchar, `stride` bytes per
 * row.  Returns 0 and a pyramid to be released with pais_pyramid_free, or < 0 (pais_pyramid_last_error).  There is no
 * host path: without a HIP device the call fails. */
int  pais_pyramid_build(int device, const uint8_t *level0, int width, int height, int64_t stride, double lod_ratio,
                        int cfg_max_lod, int build_edges, pais_pyramid **out);
void pais_pyramid_free(pais_pyramid *p);
const char *pais_pyramid_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
