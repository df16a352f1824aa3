/*
 * pais_io.h -- C ABI of the on-disk formats around the hot path.
 * Host-only code (no GPU involved); mirrors the reference's FileLoader / FileWriter:
 *
 *   config.txt            FileLoader::loadConfig      io/fileloader.cpp:474-564
 *   NVM_V3 (.nvm/.nvm2)   FileLoader::loadNVM/NVM2    io/fileloader.cpp:15-165, 251-401
 *   MVS_V2 / MVS_V3       FileLoader::loadMVS         io/fileloader.cpp:167-249, 403-472
 *                         FileWriter::writeMVS        io/filewriter.cpp:3-102
 *   PLY (ascii)           FileWriter::writePLY        io/filewriter.cpp:104-139
 *   PSR (raw floats)      FileWriter::wirtePSR        io/filewriter.cpp:141-171
 *
 * Files written here are byte-compatible with what the reference writes for the
 * same content, and files the reference wrote load here.
 */
#ifndef PAIS_IO_H
#define PAIS_IO_H

#include "pais_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One camera line of an NVM file / one camera record of an MVS file. */
typedef struct pais_io_camera {
    char   file_name[256];        /* MAX_FILE_NAME_LENGTH, camera.h:4 */
    double focal[2];
    double principle_point[2];    /* (-1,-1): "image centre" (NVM has no principal point, fileloader.cpp:59) */
    double quaternion[4];         /* w x y z */
    double center[3];
    double radial_distortion;
} pais_io_camera;

/* One point line of an NVM file (fileloader.cpp:112-165): the seed a Patch is built from. */
typedef struct pais_io_point {
    double  center[3];
    uint8_t rgb[3];
    uint8_t _pad;
    int32_t num_meas;
    int32_t cam_idx[PAIS_MAX_VIS];
    int32_t feat_idx[PAIS_MAX_VIS];
    double  xy[PAIS_MAX_VIS][2];  /* as in the file: relative to the image centre */
} pais_io_point;

/* One patch record of an MVS file (filewriter.cpp:49-69). */
typedef struct pais_io_patch {
    double  center[3];
    double  normalS[2];
    int32_t num_cam;
    int32_t cam_idx[PAIS_MAX_VIS];
    double  fitness;
    double  correlation;
} pais_io_patch;

typedef struct pais_io_scene pais_io_scene;  /* parsed file content */

/* config.txt: updates the keys present in the file, leaves the others (22 keys; `gradientWeighting`
 * and `neighborRadius` are NOT keys of the reference's parser -- SURVEY D6). 0 ok, <0 cannot open. */
int  pais_io_load_config(const char *path, pais_config *inout);

/* nvm2 != 0: the NVM2 dialect (fx fy px py after the name).  Returns NULL on failure. */
pais_io_scene *pais_io_load_nvm(const char *path, int nvm2);
/* MVS_V2 / MVS_V3; for V3 *has_config is set and *cfg filled from the embedded MvsConfig blob. */
pais_io_scene *pais_io_load_mvs(const char *path, pais_config *cfg, int *has_config);
void pais_io_free(pais_io_scene *s);
/* Lists longer than PAIS_MAX_VIS (include/pais_hip.h) are cut when loading; this is how many were.  A non-zero count
 * means the scene exceeds what the refine path tracks per patch: results would differ from the reference's. */
int  pais_io_num_truncated(const pais_io_scene *s);
int  pais_io_num_cameras(const pais_io_scene *s);
int  pais_io_num_points(const pais_io_scene *s);    /* NVM points */
int  pais_io_num_patches(const pais_io_scene *s);   /* MVS patches */
int  pais_io_get_camera(const pais_io_scene *s, int i, pais_io_camera *out);
int  pais_io_get_point(const pais_io_scene *s, int i, pais_io_point *out);
int  pais_io_get_patch(const pais_io_scene *s, int i, pais_io_patch *out);

/* writers; normals / colours are per patch (colours BGR as the reference's Vec3b, may be NULL = black) */
int  pais_io_write_mvs(const char *path, const pais_config *cfg, int num_cams, const pais_io_camera *cams,
                       int num_patches, const pais_io_patch *patches);
int  pais_io_write_ply(const char *path, int n, const double *centers /* n x 3 */, const double *normals /* n x 3 */,
                       const uint8_t *bgr /* n x 3 or NULL */);
int  pais_io_write_psr(const char *path, int n, const double *centers, const double *normals);

size_t pais_io_sizeof_mvsconfig_disk(void);   /* 160: the reference's sizeof(MvsConfig) */

#ifdef __cplusplus
}
#endif
#endif
