// pais_io.hip -- host-only: the reference's file formats (include/pais_io.h).  No device code.
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

#include "../../include/pais_io.h"

namespace {

// PAIS::MvsConfig exactly as the reference declares it (mvs/mvs.h:19-72): written raw into MVS_V3 files
// (filewriter.cpp:3-6) and read back raw (fileloader.cpp:167-171).  Natural alignment, 160 bytes.
struct MvsConfigDisk {
    int cellSize;
    int patchRadius;
    int patchSize;
    int minCamNum;
    double textureVariation;
    double visibleCorrelation;
    double minCorrelation;
    double maxFitness;
    double lodRatio;
    int minLOD;
    int maxLOD;
    int maxCellPatchNum;
    double reduceNormalRange;
    bool adaptiveDistanceEnable;
    bool adaptiveDifferenceEnable;
    bool adaptiveGradientEnable;
    double distWeighting;
    double diffWeighting;
    double gradientWeighting;
    double neighborRadius;
    double neighborRadiusScalar;
    double minRegionRatio;
    double depthRangeScalar;
    int particleNum;
    int maxIteration;
    int expansionStrategy;
};
static_assert(sizeof(MvsConfigDisk) == 160, "MvsConfig on-disk layout");
static_assert(offsetof(MvsConfigDisk, textureVariation) == 16, "layout");
static_assert(offsetof(MvsConfigDisk, minLOD) == 56, "layout");
static_assert(offsetof(MvsConfigDisk, reduceNormalRange) == 72, "layout");
static_assert(offsetof(MvsConfigDisk, adaptiveDistanceEnable) == 80, "layout");
static_assert(offsetof(MvsConfigDisk, distWeighting) == 88, "layout");
static_assert(offsetof(MvsConfigDisk, particleNum) == 144, "layout");
static_assert(offsetof(MvsConfigDisk, expansionStrategy) == 152, "layout");

void toDisk(const pais_config &c, MvsConfigDisk &d)
{
    memset(&d, 0, sizeof(d));
    d.cellSize = c.cellSize; d.patchRadius = c.patchRadius; d.patchSize = (c.patchRadius << 1) + 1; d.minCamNum = c.minCamNum;
    d.textureVariation = c.textureVariation; d.visibleCorrelation = c.visibleCorrelation; d.minCorrelation = c.minCorrelation;
    d.maxFitness = c.maxFitness; d.lodRatio = c.lodRatio; d.minLOD = c.minLOD; d.maxLOD = c.maxLOD;
    d.maxCellPatchNum = c.maxCellPatchNum; d.reduceNormalRange = c.reduceNormalRange;
    d.adaptiveDistanceEnable = c.adaptiveDistanceEnable != 0; d.adaptiveDifferenceEnable = c.adaptiveDifferenceEnable != 0;
    d.adaptiveGradientEnable = c.adaptiveGradientEnable != 0;
    d.distWeighting = c.distWeighting; d.diffWeighting = c.diffWeighting; d.gradientWeighting = c.gradientWeighting;
    d.neighborRadius = c.neighborRadius; d.neighborRadiusScalar = c.neighborRadiusScalar; d.minRegionRatio = c.minRegionRatio;
    d.depthRangeScalar = c.depthRangeScalar; d.particleNum = c.particleNum; d.maxIteration = c.maxIteration;
    d.expansionStrategy = c.expansionStrategy;
}
void fromDisk(const MvsConfigDisk &d, pais_config &c)
{
    memset(&c, 0, sizeof(c));
    c.cellSize = d.cellSize; c.patchRadius = d.patchRadius; c.patchSize = d.patchSize; c.minCamNum = d.minCamNum;
    c.textureVariation = d.textureVariation; c.visibleCorrelation = d.visibleCorrelation; c.minCorrelation = d.minCorrelation;
    c.maxFitness = d.maxFitness; c.lodRatio = d.lodRatio; c.minLOD = d.minLOD; c.maxLOD = d.maxLOD;
    c.maxCellPatchNum = d.maxCellPatchNum; c.reduceNormalRange = d.reduceNormalRange;
    c.adaptiveDistanceEnable = d.adaptiveDistanceEnable; c.adaptiveDifferenceEnable = d.adaptiveDifferenceEnable;
    c.adaptiveGradientEnable = d.adaptiveGradientEnable;
    c.distWeighting = d.distWeighting; c.diffWeighting = d.diffWeighting; c.gradientWeighting = d.gradientWeighting;
    c.neighborRadius = d.neighborRadius; c.neighborRadiusScalar = d.neighborRadiusScalar; c.minRegionRatio = d.minRegionRatio;
    c.depthRangeScalar = d.depthRangeScalar; c.particleNum = d.particleNum; c.maxIteration = d.maxIteration;
    c.expansionStrategy = d.expansionStrategy;
}

const char *DELIM = " \t";
const int BUF = 10240; // STRING_BUFFER_LENGTH, fileloader.cpp:1

} // namespace

struct pais_io_scene {
    std::vector<pais_io_camera> cams;
    std::vector<pais_io_point> points;
    std::vector<pais_io_patch> patches;
    int truncated = 0; // camera / measurement lists longer than PAIS_MAX_VIS that were cut
};

extern "C" size_t pais_io_sizeof_mvsconfig_disk(void) { return sizeof(MvsConfigDisk); }

// FileLoader::loadConfig, fileloader.cpp:474-564
extern "C" int pais_io_load_config(const char *path, pais_config *config)
{
    if (!path || !config) return -1;
    std::ifstream file(path, std::ifstream::in);
    if (!file.is_open()) return -2;
    std::vector<char> buf(BUF);
    while (!file.eof()) {
        file.getline(buf.data(), BUF);
        if (file.fail() && !file.eof()) file.clear();
        char *strbuf = buf.data();
        if (strbuf[0] == '#') continue; // skip comment
        char *key = strtok(strbuf, DELIM);
        if (key == NULL) continue; // blank line
        char *val = strtok(NULL, " \t\r");
        if (val == NULL) continue; // (the reference would dereference NULL here)
#define KEY_I(name) if (strcmp(key, #name) == 0) { config->name = atoi(val); continue; }
#define KEY_D(name) if (strcmp(key, #name) == 0) { config->name = atof(val); continue; }
        if (strcmp(key, "patchRadius") == 0) {
            config->patchRadius = atoi(val);
            config->patchSize = (config->patchRadius << 1) + 1;
            continue;
        }
        KEY_D(reduceNormalRange) KEY_I(adaptiveDistanceEnable) KEY_I(adaptiveDifferenceEnable) KEY_I(adaptiveGradientEnable)
        KEY_D(distWeighting) KEY_D(diffWeighting) KEY_D(visibleCorrelation) KEY_D(depthRangeScalar)
        KEY_I(particleNum) KEY_I(maxIteration) KEY_I(cellSize) KEY_I(maxCellPatchNum) KEY_I(expansionStrategy)
        KEY_D(textureVariation) KEY_I(minLOD) KEY_I(maxLOD) KEY_D(lodRatio) KEY_I(minCamNum) KEY_D(minCorrelation)
        KEY_D(minRegionRatio) KEY_D(maxFitness) KEY_D(neighborRadiusScalar)
#undef KEY_I
#undef KEY_D
        // unknown keys (including gradientWeighting, README.md:142) are ignored, as in the reference
    }
    return 0;
}

// loadNvmCamera / loadNvm2Camera, fileloader.cpp:15-110
static bool parse_nvm_camera(char *line, bool nvm2, pais_io_camera &c)
{
    memset(&c, 0, sizeof(c));
    char *t = strtok(line, DELIM);
    if (!t) return false;
    strncpy(c.file_name, t, sizeof(c.file_name) - 1);
    auto next = [&](double &v) { char *q = strtok(NULL, " \t\r"); if (!q) return false; v = atof(q); return true; };
    if (!nvm2) {
        double f;
        if (!next(f)) return false;
        c.focal[0] = f; c.focal[1] = f;
        c.principle_point[0] = -1; c.principle_point[1] = -1; // Vec2d(-1,-1): image centre (camera.cpp:101-104)
    } else {
        if (!next(c.focal[0]) || !next(c.focal[1]) || !next(c.principle_point[0]) || !next(c.principle_point[1])) return false;
    }
    for (int i = 0; i < 4; ++i) if (!next(c.quaternion[i])) return false;
    for (int i = 0; i < 3; ++i) if (!next(c.center[i])) return false;
    if (!nvm2) { if (!next(c.radial_distortion)) return false; } else c.radial_distortion = 0;
    return true;
}

// loadNvmPatch, fileloader.cpp:112-165 (image-centre offset is applied by the caller, who knows the image size)
static bool parse_nvm_point(char *line, pais_io_point &p, int *truncated = nullptr)
{
    memset(&p, 0, sizeof(p));
    char *t = strtok(line, DELIM);
    if (!t) return false;
    p.center[0] = atof(t);
    auto nd = [&](double &v) { char *q = strtok(NULL, " \t\r"); if (!q) return false; v = atof(q); return true; };
    auto ni = [&](int &v) { char *q = strtok(NULL, " \t\r"); if (!q) return false; v = atoi(q); return true; };
    if (!nd(p.center[1]) || !nd(p.center[2])) return false;
    int r, g, b, n;
    if (!ni(r) || !ni(g) || !ni(b) || !ni(n)) return false;
    p.rgb[0] = (uint8_t)r; p.rgb[1] = (uint8_t)g; p.rgb[2] = (uint8_t)b;
    if (n < 0) n = 0;
    p.num_meas = n > PAIS_MAX_VIS ? PAIS_MAX_VIS : n;
    if (n > PAIS_MAX_VIS && truncated) ++*truncated;
    for (int i = 0; i < n; ++i) {
        int ci, fi; double x, y;
        if (!ni(ci) || !ni(fi) || !nd(x) || !nd(y)) return false;
        if (i < PAIS_MAX_VIS) { p.cam_idx[i] = ci; p.feat_idx[i] = fi; p.xy[i][0] = x; p.xy[i][1] = y; }
    }
    return true;
}

// FileLoader::loadNVM / loadNVM2, fileloader.cpp:251-401
extern "C" pais_io_scene *pais_io_load_nvm(const char *path, int nvm2)
{
    if (!path) return NULL;
    std::ifstream file(path, std::ifstream::in);
    if (!file.is_open()) return NULL;
    pais_io_scene *s = new pais_io_scene();
    std::vector<char> buf(BUF), copy(BUF);
    bool loadCamera = false, loadPatch = false;
    while (!file.eof()) {
        file.getline(buf.data(), BUF);
        if (file.fail() && !file.eof()) file.clear();
        memcpy(copy.data(), buf.data(), BUF);
        char *tok = strtok(copy.data(), " \t\r");
        if (tok == NULL) continue;
        if (strcmp(tok, "NVM_V3") == 0) { loadCamera = true; continue; }
        if (loadCamera) {
            int num = atoi(tok);
            for (int i = 0; i < num && !file.eof(); ++i) {
                file.getline(buf.data(), BUF);
                pais_io_camera c;
                if (parse_nvm_camera(buf.data(), nvm2 != 0, c)) s->cams.push_back(c);
            }
            loadCamera = false;
            loadPatch = true;
            continue;
        }
        if (loadPatch) {
            int num = atoi(tok);
            for (int i = 0; i < num && !file.eof(); ++i) {
                file.getline(buf.data(), BUF);
                pais_io_point p;
                if (parse_nvm_point(buf.data(), p, &s->truncated)) s->points.push_back(p);
            }
            break;
        }
    }
    return s;
}

// FileLoader::loadMVS, fileloader.cpp:403-472 (+ loadMvsCamera :173-204, loadMvsPatch :206-231)
extern "C" pais_io_scene *pais_io_load_mvs(const char *path, pais_config *cfg, int *has_config)
{
    if (has_config) *has_config = 0;
    if (!path) return NULL;
    std::ifstream file(path, std::ifstream::in | std::ifstream::binary);
    if (!file.is_open()) return NULL;
    pais_io_scene *s = new pais_io_scene();
    std::vector<char> buf(BUF);
    bool loadCamera = false, loadPatch = false;
    auto rd = [&](void *p, size_t n) { file.read((char *)p, (std::streamsize)n); return (bool)file; };
    while (!file.eof()) {
        file.getline(buf.data(), BUF);
        if (!file) break;
        char *tok = strtok(buf.data(), DELIM);
        if (tok == NULL) continue;
        if (strcmp(tok, "MVS_V2") == 0) { loadCamera = true; continue; }
        if (strcmp(tok, "MVS_V3") == 0) {
            MvsConfigDisk d;
            if (!rd(&d, sizeof(d))) break;
            if (cfg) fromDisk(d, *cfg);
            if (has_config) *has_config = 1;
            loadCamera = true;
            continue;
        }
        if (loadCamera) { // "CAMERAS n"
            char *q = strtok(NULL, DELIM);
            int num = q ? atoi(q) : 0;
            for (int i = 0; i < num; ++i) {
                pais_io_camera c;
                memset(&c, 0, sizeof(c));
                int len = 0;
                if (!rd(&len, sizeof(int)) || len < 0 || len > 100000) { num = 0; break; }
                std::string name((size_t)len, '\0');
                if (len && !rd(&name[0], (size_t)len)) break;
                strncpy(c.file_name, name.c_str(), sizeof(c.file_name) - 1);
                if (!rd(c.center, 24) || !rd(c.focal, 16) || !rd(c.principle_point, 16) || !rd(c.quaternion, 32) ||
                    !rd(&c.radial_distortion, 8)) break;
                s->cams.push_back(c);
            }
            loadCamera = false;
            loadPatch = true;
            continue;
        }
        if (loadPatch) { // "PATCHES n"
            char *q = strtok(NULL, DELIM);
            int num = q ? atoi(q) : 0;
            for (int i = 0; i < num; ++i) {
                pais_io_patch p;
                memset(&p, 0, sizeof(p));
                int camNum = 0;
                if (!rd(p.center, 24) || !rd(p.normalS, 16) || !rd(&camNum, sizeof(int)) || camNum < 0 || camNum > 100000) break;
                p.num_cam = camNum > PAIS_MAX_VIS ? PAIS_MAX_VIS : camNum;
                if (camNum > PAIS_MAX_VIS) s->truncated++;
                bool ok = true;
                for (int k = 0; k < camNum; ++k) {
                    int idx;
                    if (!rd(&idx, sizeof(int))) { ok = false; break; }
                    if (k < PAIS_MAX_VIS) p.cam_idx[k] = idx;
                }
                if (!ok || !rd(&p.fitness, 8) || !rd(&p.correlation, 8)) break;
                s->patches.push_back(p);
            }
            loadPatch = false;
        }
    }
    return s;
}

extern "C" void pais_io_free(pais_io_scene *s) { delete s; }
extern "C" int pais_io_num_truncated(const pais_io_scene *s) { return s ? s->truncated : 0; }
extern "C" int pais_io_num_cameras(const pais_io_scene *s) { return s ? (int)s->cams.size() : 0; }
extern "C" int pais_io_num_points(const pais_io_scene *s) { return s ? (int)s->points.size() : 0; }
extern "C" int pais_io_num_patches(const pais_io_scene *s) { return s ? (int)s->patches.size() : 0; }
extern "C" int pais_io_get_camera(const pais_io_scene *s, int i, pais_io_camera *out)
{
    if (!s || !out || i < 0 || i >= (int)s->cams.size()) return -1;
    *out = s->cams[i];
    return 0;
}
extern "C" int pais_io_get_point(const pais_io_scene *s, int i, pais_io_point *out)
{
    if (!s || !out || i < 0 || i >= (int)s->points.size()) return -1;
    *out = s->points[i];
    return 0;
}
extern "C" int pais_io_get_patch(const pais_io_scene *s, int i, pais_io_patch *out)
{
    if (!s || !out || i < 0 || i >= (int)s->patches.size()) return -1;
    *out = s->patches[i];
    return 0;
}

// FileWriter::writeMVS, filewriter.cpp:71-102 (+ writeCamera :26-47, writePatch :49-69)
extern "C" int pais_io_write_mvs(const char *path, const pais_config *cfg, int num_cams, const pais_io_camera *cams,
                                 int num_patches, const pais_io_patch *patches)
{
    if (!path || !cfg || num_cams < 0 || num_patches < 0 || (num_cams && !cams) || (num_patches && !patches)) return -1;
    std::fstream file;
    file.open(path, std::fstream::out | std::fstream::binary);
    if (!file.is_open()) return -2;
    file << "MVS_V3" << std::endl;
    MvsConfigDisk d;
    toDisk(*cfg, d);
    file.write((const char *)&d, sizeof(d));
    file << "CAMERAS " << num_cams << std::endl;
    for (int i = 0; i < num_cams; ++i) {
        const pais_io_camera &c = cams[i];
        const int len = (int)strlen(c.file_name);
        file.write((const char *)&len, sizeof(int));
        file.write(c.file_name, len);
        file.write((const char *)c.center, 24);
        file.write((const char *)c.focal, 16);
        file.write((const char *)c.principle_point, 16);
        file.write((const char *)c.quaternion, 32);
        file.write((const char *)&c.radial_distortion, 8);
    }
    file << "PATCHES " << num_patches << std::endl;
    for (int i = 0; i < num_patches; ++i) {
        const pais_io_patch &p = patches[i];
        file.write((const char *)p.center, 24);
        file.write((const char *)p.normalS, 16);
        file.write((const char *)&p.num_cam, sizeof(int));
        for (int k = 0; k < p.num_cam; ++k) file.write((const char *)&p.cam_idx[k], sizeof(int));
        file.write((const char *)&p.fitness, 8);
        file.write((const char *)&p.correlation, 8);
    }
    file.close();
    return file.fail() ? -3 : 0;
}

// FileWriter::writePLY, filewriter.cpp:104-139 (default ostream formatting, colours written R G B from BGR)
extern "C" int pais_io_write_ply(const char *path, int n, const double *centers, const double *normals, const uint8_t *bgr)
{
    if (!path || n < 0 || (n && (!centers || !normals))) return -1;
    std::ofstream file;
    file.open(path, std::ofstream::out);
    if (!file.is_open()) return -2;
    file << "ply" << std::endl;
    file << "format ascii 1.0" << std::endl;
    file << "element vertex " << n << std::endl;
    file << "property float x" << std::endl;
    file << "property float y" << std::endl;
    file << "property float z" << std::endl;
    file << "property float nx" << std::endl;
    file << "property float ny" << std::endl;
    file << "property float nz" << std::endl;
    file << "property uchar diffuse_red" << std::endl;
    file << "property uchar diffuse_green" << std::endl;
    file << "property uchar diffuse_blue" << std::endl;
    file << "end_header" << std::endl;
    for (int i = 0; i < n; ++i) {
        const double *p = centers + 3 * i, *q = normals + 3 * i;
        const uint8_t c0 = bgr ? bgr[3 * i] : 0, c1 = bgr ? bgr[3 * i + 1] : 0, c2 = bgr ? bgr[3 * i + 2] : 0;
        file << p[0] << " " << p[1] << " " << p[2] << " ";
        file << q[0] << " " << q[1] << " " << q[2] << " ";
        file << int(c2) << " " << int(c1) << " " << int(c0) << std::endl;
    }
    file.close();
    return 0;
}

// FileWriter::wirtePSR, filewriter.cpp:141-171
extern "C" int pais_io_write_psr(const char *path, int n, const double *centers, const double *normals)
{
    if (!path || n < 0 || (n && (!centers || !normals))) return -1;
    std::ofstream file;
    file.open(path, std::ofstream::binary);
    if (!file.is_open()) return -2;
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { float v = (float)centers[3 * i + k]; file.write((const char *)&v, sizeof(float)); }
        for (int k = 0; k < 3; ++k) { float v = (float)normals[3 * i + k]; file.write((const char *)&v, sizeof(float)); }
    }
    file.close();
    return 0;
}
