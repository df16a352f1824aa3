/* po_io.c -- TEST INFRASTRUCTURE (part of the oracle, never linked into the product).
 *
 * Independent C restatement of the reference's on-disk formats and camera preparation, used to cross-check the
 * product's pais_io / pais_pyramid.  Written from the reference sources only:
 *   MVS_V3 writer   io/filewriter.cpp:3-103        (header line, raw MvsConfig, CAMERAS n, PATCHES n)
 *   MVS_V3 reader   io/fileloader.cpp:167-231, 403-472
 *   NVM point line  io/fileloader.cpp:112-165      (pixel offsets are relative to the image centre: + cols/2, rows/2)
 *   MvsConfig       mvs/mvs.h:19-72                (the struct is written raw: MSVC x64 / gcc x86-64 natural alignment,
 *                                                   `bool` = 1 byte -> 160 bytes)
 *   pyramid levels  mvs/camera.cpp:63-64, 85       (maxLOD; resize(level0, s = ratio^i, INTER_AREA))
 *   edge maps       mvs/camera.cpp:72-77, 87-91    (Sobel ksize 1, magnitude, min-max normalisation per level)
 * cv::resize(INTER_AREA) and cv::Sobel live in OpenCV 2.4.2, which is not under /root/reference: their published
 * algorithms are restated (parity unpinned):
 *   po_resize_area      -- area interpolation as a separable operator in double precision: per axis the decimation
 *                          table of a fractional scale (first partial cell, whole cells, last partial cell, weights
 *                          normalised to the cell), rows reduced first, then columns, round-half-even, saturate.
 *   po_resize_area_f32  -- the same table evaluated the way OpenCV 2.4's resizeArea_<uchar, float> does it as far as it
 *                          is documented by its source: float weights with the 1/(sx*sy) area factor folded into the
 *                          horizontal taps, every source row reduced horizontally into a float buffer, buffers blended
 *                          vertically in source-row order, saturate_cast<uchar>(float).  Differs from the double
 *                          version by +-1 grey level on a small fraction of pixels.
 */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pais_oracle.h"

/* ---- mvs/mvs.h:19-72, field for field ---------------------------------------------------------------------------- */
typedef struct {
    int cellSize, patchRadius, patchSize, minCamNum;
    double textureVariation, visibleCorrelation, minCorrelation, maxFitness, lodRatio;
    int minLOD, maxLOD, maxCellPatchNum;
    double reduceNormalRange;
    bool adaptiveDistanceEnable, adaptiveDifferenceEnable, adaptiveGradientEnable;
    double distWeighting, diffWeighting, gradientWeighting, neighborRadius, neighborRadiusScalar, minRegionRatio, depthRangeScalar;
    int particleNum, maxIteration, expansionStrategy;
} po_mvsconfig_raw;

size_t po_io_sizeof_mvsconfig(void) { return sizeof(po_mvsconfig_raw); }

static void cfg_to_raw(const po_config *c, po_mvsconfig_raw *r)
{
    memset(r, 0, sizeof(*r)); /* the reference writes whatever sits in the padding; zeros here */
    r->cellSize = c->cellSize; r->patchRadius = c->patchRadius; r->patchSize = c->patchSize; r->minCamNum = c->minCamNum;
    r->textureVariation = c->textureVariation; r->visibleCorrelation = c->visibleCorrelation; r->minCorrelation = c->minCorrelation;
    r->maxFitness = c->maxFitness; r->lodRatio = c->lodRatio;
    r->minLOD = c->minLOD; r->maxLOD = c->maxLOD; r->maxCellPatchNum = c->maxCellPatchNum;
    r->reduceNormalRange = c->reduceNormalRange;
    r->adaptiveDistanceEnable = c->adaptiveDistanceEnable != 0; r->adaptiveDifferenceEnable = c->adaptiveDifferenceEnable != 0;
    r->adaptiveGradientEnable = c->adaptiveGradientEnable != 0;
    r->distWeighting = c->distWeighting; r->diffWeighting = c->diffWeighting; r->gradientWeighting = c->gradientWeighting;
    r->neighborRadius = c->neighborRadius; r->neighborRadiusScalar = c->neighborRadiusScalar; r->minRegionRatio = c->minRegionRatio;
    r->depthRangeScalar = c->depthRangeScalar;
    r->particleNum = c->particleNum; r->maxIteration = c->maxIteration; r->expansionStrategy = c->expansionStrategy;
}
static void raw_to_cfg(const po_mvsconfig_raw *r, po_config *c)
{
    c->cellSize = r->cellSize; c->patchRadius = r->patchRadius; c->patchSize = r->patchSize; c->minCamNum = r->minCamNum;
    c->textureVariation = r->textureVariation; c->visibleCorrelation = r->visibleCorrelation; c->minCorrelation = r->minCorrelation;
    c->maxFitness = r->maxFitness; c->lodRatio = r->lodRatio;
    c->minLOD = r->minLOD; c->maxLOD = r->maxLOD; c->maxCellPatchNum = r->maxCellPatchNum;
    c->reduceNormalRange = r->reduceNormalRange;
    c->adaptiveDistanceEnable = r->adaptiveDistanceEnable; c->adaptiveDifferenceEnable = r->adaptiveDifferenceEnable;
    c->adaptiveGradientEnable = r->adaptiveGradientEnable;
    c->distWeighting = r->distWeighting; c->diffWeighting = r->diffWeighting; c->gradientWeighting = r->gradientWeighting;
    c->neighborRadius = r->neighborRadius; c->neighborRadiusScalar = r->neighborRadiusScalar; c->minRegionRatio = r->minRegionRatio;
    c->depthRangeScalar = r->depthRangeScalar;
    c->particleNum = r->particleNum; c->maxIteration = r->maxIteration; c->expansionStrategy = r->expansionStrategy;
}

/* flat records the Python tests exchange with this file */
typedef struct { char name[256]; double center[3], focal[2], pp[2], quaternion[4], radial; } po_io_camera;
typedef struct { double center[3], normalS[2]; int numCam; int camIdx[PO_MAX_VIS]; double fitness, correlation; } po_io_patch;

/* io/filewriter.cpp:70-103 */
int po_io_write_mvs_v3(const char *path, const po_config *cfg, int ncam, const po_io_camera *cams, int npatch, const po_io_patch *patches)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fputs("MVS_V3\n", f);
    po_mvsconfig_raw raw;
    cfg_to_raw(cfg, &raw);
    fwrite(&raw, sizeof(raw), 1, f);
    fprintf(f, "CAMERAS %d\n", ncam);
    for (int i = 0; i < ncam; ++i) { /* writeCamera :26-47 */
        const po_io_camera *c = &cams[i];
        const int len = (int)strlen(c->name);
        fwrite(&len, sizeof(int), 1, f);
        fwrite(c->name, 1, (size_t)len, f);
        fwrite(c->center, sizeof(double), 3, f);
        fwrite(c->focal, sizeof(double), 2, f);
        fwrite(c->pp, sizeof(double), 2, f);
        fwrite(c->quaternion, sizeof(double), 4, f);
        fwrite(&c->radial, sizeof(double), 1, f);
    }
    fprintf(f, "PATCHES %d\n", npatch);
    for (int i = 0; i < npatch; ++i) { /* writePatch :49-68 */
        const po_io_patch *p = &patches[i];
        fwrite(p->center, sizeof(double), 3, f);
        fwrite(p->normalS, sizeof(double), 2, f);
        fwrite(&p->numCam, sizeof(int), 1, f);
        fwrite(p->camIdx, sizeof(int), (size_t)p->numCam, f);
        fwrite(&p->fitness, sizeof(double), 1, f);
        fwrite(&p->correlation, sizeof(double), 1, f);
    }
    fclose(f);
    return 0;
}

/* one text line (without the newline) as ifstream::getline + strtok(" \t") see it; returns the first token or NULL */
static char *read_line_token(FILE *f, char *buf, size_t cap)
{
    size_t n = 0;
    int ch;
    while ((ch = fgetc(f)) != EOF && ch != '\n')
        if (n + 1 < cap) buf[n++] = (char)ch;
    buf[n] = 0;
    if (ch == EOF && n == 0) return NULL;
    return strtok(buf, " \t\r");
}

/* io/fileloader.cpp:403-472 (MVS_V3 only).  Arrays are caller-allocated with the given capacities. */
int po_io_read_mvs_v3(const char *path, po_config *cfg, int *hasCfg, int capCam, po_io_camera *cams, int *ncam, int capPatch,
                      po_io_patch *patches, int *npatch)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char buf[4096];
    int loadCamera = 0, loadPatch = 0;
    *hasCfg = 0; *ncam = 0; *npatch = 0;
    while (!feof(f)) {
        char *tok = read_line_token(f, buf, sizeof(buf));
        if (!tok) { if (feof(f)) break; continue; }
        if (strcmp(tok, "MVS_V3") == 0) {
            po_mvsconfig_raw raw;
            if (fread(&raw, sizeof(raw), 1, f) != 1) break;
            raw_to_cfg(&raw, cfg);
            *hasCfg = 1;
            loadCamera = 1;
            continue;
        }
        if (loadCamera) {
            char *q = strtok(NULL, " \t\r");
            const int num = q ? atoi(q) : 0;
            for (int i = 0; i < num; ++i) { /* loadMvsCamera :173-204 */
                po_io_camera c;
                memset(&c, 0, sizeof(c));
                int len = 0;
                if (fread(&len, sizeof(int), 1, f) != 1 || len < 0 || len > 255) { fclose(f); return -2; }
                if (fread(c.name, 1, (size_t)len, f) != (size_t)len) { fclose(f); return -2; }
                if (fread(c.center, sizeof(double), 3, f) != 3 || fread(c.focal, sizeof(double), 2, f) != 2 ||
                    fread(c.pp, sizeof(double), 2, f) != 2 || fread(c.quaternion, sizeof(double), 4, f) != 4 ||
                    fread(&c.radial, sizeof(double), 1, f) != 1) { fclose(f); return -2; }
                if (*ncam < capCam) cams[*ncam] = c;
                ++*ncam;
            }
            loadCamera = 0;
            loadPatch = 1;
            continue;
        }
        if (loadPatch) {
            char *q = strtok(NULL, " \t\r");
            const int num = q ? atoi(q) : 0;
            for (int i = 0; i < num; ++i) { /* loadMvsPatch :206-231 */
                po_io_patch p;
                memset(&p, 0, sizeof(p));
                int n = 0;
                if (fread(p.center, sizeof(double), 3, f) != 3 || fread(p.normalS, sizeof(double), 2, f) != 2 ||
                    fread(&n, sizeof(int), 1, f) != 1 || n < 0 || n > PO_MAX_VIS) { fclose(f); return -3; }
                p.numCam = n;
                if (fread(p.camIdx, sizeof(int), (size_t)n, f) != (size_t)n || fread(&p.fitness, sizeof(double), 1, f) != 1 ||
                    fread(&p.correlation, sizeof(double), 1, f) != 1) { fclose(f); return -3; }
                if (*npatch < capPatch) patches[*npatch] = p;
                ++*npatch;
            }
            loadPatch = 0;
        }
    }
    fclose(f);
    return 0;
}

/* io/fileloader.cpp:112-165: "x y z r g b n  (camIdx featIdx dx dy) x n"; image points = offsets + (cols/2, rows/2).
 * widths / heights: per camera index.  Returns the number of measurements or < 0. */
int po_io_parse_nvm_point(const char *line, int ncam, const int *widths, const int *heights, double center[3], int rgb[3],
                          int capMeas, int *camIdx, double *imgPoints)
{
    const size_t len = strlen(line);
    char *buf = (char *)malloc(len + 1);
    memcpy(buf, line, len + 1);
    char *t = strtok(buf, " \t\r\n");
    int rc = -1;
    double v[3];
    for (int i = 0; i < 3; ++i) { if (!t) goto done; v[i] = atof(t); t = strtok(NULL, " \t\r\n"); }
    center[0] = v[0]; center[1] = v[1]; center[2] = v[2];
    for (int i = 0; i < 3; ++i) { if (!t) goto done; rgb[i] = atoi(t); t = strtok(NULL, " \t\r\n"); }
    if (!t) goto done;
    {
        const int n = atoi(t);
        for (int i = 0; i < n; ++i) {
            t = strtok(NULL, " \t\r\n"); if (!t) goto done;
            const int idx = atoi(t);
            t = strtok(NULL, " \t\r\n"); if (!t) goto done; /* feature index */
            t = strtok(NULL, " \t\r\n"); if (!t) goto done;
            const double dx = atof(t);
            t = strtok(NULL, " \t\r\n"); if (!t) goto done;
            const double dy = atof(t);
            if (idx < 0 || idx >= ncam) goto done;
            if (i < capMeas) {
                camIdx[i] = idx;
                imgPoints[2 * i] = dx + widths[idx] / 2;      /* integer division, as `cols / 2` */
                imgPoints[2 * i + 1] = dy + heights[idx] / 2;
            }
        }
        rc = n;
    }
done:
    free(buf);
    return rc;
}

/* ---- camera preparation ------------------------------------------------------------------------------------------- */
typedef struct { int dsize, maxTaps; int *first, *count; double *w; } axis_table;

/* decimation table of one axis for a scale fx < 1: destination cell [dx*s, (dx+1)*s) with s = 1/fx */
static axis_table area_table(int ssize, double fx)
{
    axis_table t;
    int dsize = (int)lrint((double)ssize * fx); /* Size(cvRound(w*fx), ...) */
    if (dsize < 1) dsize = 1;
    const double scale = 1.0 / fx;
    t.dsize = dsize;
    t.maxTaps = (int)ceil(scale) + 2;
    t.first = (int *)calloc((size_t)dsize, sizeof(int));
    t.count = (int *)calloc((size_t)dsize, sizeof(int));
    t.w = (double *)calloc((size_t)dsize * t.maxTaps, sizeof(double));
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        double *w = t.w + (size_t)dx * t.maxTaps;
        int n = 0, first = sx1;
        if (sx1 - fsx1 > 1e-3) { first = sx1 - 1; w[n++] = (sx1 - fsx1) / cell; }
        for (int sx = sx1; sx < sx2; ++sx) w[n++] = 1.0 / cell;
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            if (a > 1.0) a = 1.0;
            if (a > cell) a = cell;
            w[n++] = a / cell;
        }
        double rs = 0;
        for (int k = 0; k < n; ++k) rs += w[k];
        if (rs == 0) rs = 1.0;
        const double inv = 1.0 / rs;
        for (int k = 0; k < n; ++k) w[k] = inv * w[k];
        t.first[dx] = first;
        t.count[dx] = n;
    }
    return t;
}
static void free_table(axis_table *t) { free(t->first); free(t->count); free(t->w); }

void po_resize_dims(int w, int h, double fx, int *dw, int *dh)
{
    int a = (int)lrint((double)w * fx), b = (int)lrint((double)h * fx);
    *dw = a < 1 ? 1 : a;
    *dh = b < 1 ? 1 : b;
}

/* double precision, rows then columns (what the product's kernels and bench use) */
void po_resize_area(const uint8_t *src, int w, int h, double fx, uint8_t *dst)
{
    axis_table ty = area_table(h, fx), tx = area_table(w, fx);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)ty.dsize * w);
    for (int dy = 0; dy < ty.dsize; ++dy)
        for (int x = 0; x < w; ++x) {
            double s = 0;
            for (int k = 0; k < ty.count[dy]; ++k) s += ty.w[(size_t)dy * ty.maxTaps + k] * (double)src[(size_t)(ty.first[dy] + k) * w + x];
            tmp[(size_t)dy * w + x] = s;
        }
    for (int dy = 0; dy < ty.dsize; ++dy)
        for (int dx = 0; dx < tx.dsize; ++dx) {
            double s = 0;
            for (int j = 0; j < tx.count[dx]; ++j) s += tx.w[(size_t)dx * tx.maxTaps + j] * tmp[(size_t)dy * w + tx.first[dx] + j];
            double r = rint(s);
            r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
            dst[(size_t)dy * tx.dsize + dx] = (uint8_t)r;
        }
    free(tmp);
    free_table(&ty);
    free_table(&tx);
}

/* OpenCV 2.4 resizeArea_<uchar, float> order (see the header): float taps normalised to the (possibly clipped) cell
 * width, horizontal reduction of every source row into a float buffer, vertical blending in source-row order divided by
 * the (possibly clipped) cell height, saturate_cast<uchar>(float) = round-half-even + clip */
void po_resize_area_f32(const uint8_t *src, int w, int h, double fx, uint8_t *dst)
{
    int dw, dh;
    po_resize_dims(w, h, fx, &dw, &dh);
    const double scale_x = 1.0 / fx, scale_y = 1.0 / fx;
    int cap = w * 2 + 2, nk = 0;
    int *di = (int *)malloc(sizeof(int) * (size_t)cap), *si = (int *)malloc(sizeof(int) * (size_t)cap);
    float *al = (float *)malloc(sizeof(float) * (size_t)cap);
    for (int dx = 0; dx < dw; ++dx) {
        const double fsx1 = dx * scale_x, fsx2 = fsx1 + scale_x;
        const double cellWidth = scale_x < w - fsx1 ? scale_x : w - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > w - 1) sx2 = w - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { di[nk] = dx; si[nk] = sx1 - 1; al[nk++] = (float)((sx1 - fsx1) / cellWidth); }
        for (int sx = sx1; sx < sx2; ++sx) { di[nk] = dx; si[nk] = sx; al[nk++] = (float)(1.0 / cellWidth); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            if (a > 1.0) a = 1.0;
            if (a > cellWidth) a = cellWidth;
            di[nk] = dx; si[nk] = sx2; al[nk++] = (float)(a / cellWidth);
        }
    }
    float *buf = (float *)calloc((size_t)dw, sizeof(float)), *sum = (float *)calloc((size_t)dw, sizeof(float));
    int cur_dy = 0;
    const float sy_f = (float)scale_y;
    for (int sy = 0; sy < h; ++sy) {
        const uint8_t *S = src + (size_t)sy * w;
        for (int k = 0; k < nk; ++k) buf[di[k]] += S[si[k]] * al[k];
        if ((cur_dy + 1) * sy_f <= sy + 1 || sy == h - 1) {
            float beta = sy + 1 - (cur_dy + 1) * sy_f;
            if (beta < 0) beta = 0;
            const float beta1 = 1 - beta;
            if (cur_dy >= dh) break;
            const float cellH = sy_f < h - cur_dy * sy_f ? sy_f : h - cur_dy * sy_f;
            uint8_t *D = dst + (size_t)cur_dy * dw;
            for (int dx = 0; dx < dw; ++dx) {
                const float v = (fabsf(beta) < 1e-3f ? (sum[dx] + buf[dx]) : (sum[dx] + buf[dx] * beta1)) / cellH;
                float r = rintf(v);
                r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
                D[dx] = (uint8_t)r;
                sum[dx] = fabsf(beta) < 1e-3f ? 0.f : buf[dx] * beta;
                buf[dx] = 0.f;
            }
            cur_dy++;
        } else {
            for (int dx = 0; dx < dw; ++dx) { sum[dx] += buf[dx]; buf[dx] = 0.f; }
        }
    }
    if (cur_dy < dh) { /* the last (clipped) cell when its first source row was also the image's last */
        const float cellH = sy_f < h - cur_dy * sy_f ? sy_f : h - cur_dy * sy_f;
        uint8_t *D = dst + (size_t)cur_dy * dw;
        for (int dx = 0; dx < dw; ++dx) {
            float r = rintf(sum[dx] / cellH);
            r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
            D[dx] = (uint8_t)r;
        }
    }
    free(di); free(si); free(al); free(buf); free(sum);
}

/* camera.cpp:72-77, 87-91: Sobel(CV_64F, 1, 0, ksize 1) / (0, 1): central differences, BORDER_REFLECT_101; magnitude;
 * (m - min) / (max - min) over the level (0 when the level is flat) */
void po_sobel_magnitude_normalised(const uint8_t *img, int w, int h, double *out)
{
    double mn = INFINITY, mx = -INFINITY;
    for (int y = 0; y < h; ++y) {
        const int yu = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yd = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
        for (int x = 0; x < w; ++x) {
            const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
            const double gx = (double)img[(size_t)y * w + xr] - (double)img[(size_t)y * w + xl];
            const double gy = (double)img[(size_t)yd * w + x] - (double)img[(size_t)yu * w + x];
            const double m = sqrt(gx * gx + gy * gy);
            out[(size_t)y * w + x] = m;
            if (m < mn) mn = m;
            if (m > mx) mx = m;
        }
    }
    for (size_t i = 0; i < (size_t)w * h; ++i) out[i] = (mx > mn) ? (out[i] - mn) / (mx - mn) : 0.0;
}
