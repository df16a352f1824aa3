/* pais_seed.h -- seed patches from image features: FeatureManager::setSeedPatches after its SIFT call
 * (mvs/featuremanager.cpp:28-99, 118-287).  The step before the path: brute-force cross-checked descriptor
 * matching between every ordered pair of cameras (the one heavy part: a HIP kernel), epipolar-line filtering with the
 * pair's fundamental matrix, removal of non-cross matches and of weakly matched views, union of the pairwise matches into
 * n-view features, one seed per feature with at least minCamNum views through Patch::reCentering.
 * The keypoints and descriptors themselves (cv::SIFT, OpenCV non-free) are the caller's. */
#ifndef PAIS_SEED_H
#define PAIS_SEED_H

#include "pais_mvs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* keypoints[i].pt and the descriptor rows of one camera (featuremanager.cpp:9-11, 24-25) */
typedef struct pais_keypoints {
    int32_t      n;
    int32_t      _pad;
    const float *xy;    /* n x 2: KeyPoint::pt (pixels)                  */
    const float *desc;  /* n x dim, row-major (cv::SIFT: dim = 128)      */
} pais_keypoints;

/* FeatureManager::getFundamental (featuremanager.cpp:245-262): F = [eT]x * P_to * pinv(P_from), eT = P_to * (C_from, 1);
 * x_to' F x_from = 0.  Host only. */
int  pais_seed_fundamental(const pais_camera_desc *from, const pais_camera_desc *to, double F[9]);

/* BFMatcher(NORM_L2, crossCheck = true).match(query, train) (featuremanager.cpp:32-39) on HIP device `device`:
 * train_of_query[q] = index of the train descriptor matched to query q, or -1; dist[q] = L2 distance of q to its nearest
 * train descriptor (float).  Host pointers. */
int  pais_seed_match(int device, int nq, const float *query_desc, int nt, const float *train_desc, int dim,
                     int32_t *train_of_query, float *dist);

/* The nearest searches of a whole rig in one call: every camera's descriptors are uploaded ONCE and nearest(i -> j) is
 * evaluated once per ORDERED camera pair (pais_seed_match per pair would upload both sets and search both directions for
 * (i, j) and again for (j, i)).  nearest: sum over i of kp[i].n * (num_cams - 1) entries, pair major -- for i ascending,
 * for j ascending without i, the kp[i].n indices into camera j's descriptors (-1 when camera j has none); first minimum
 * of the float L2 distance as in pais_seed_match.  The cross-check of a pair is `nearest(j -> i)[nearest(i -> j)[q]] == q`. */
int  pais_seed_nearest_all(int device, int num_cams, const pais_keypoints *kp, int dim, int32_t *nearest);

/* One pairwise match as the reference's match table holds it: matchTable[cam_q][cam_t] gets DMatch(q, t). */
typedef struct pais_pair_match { int32_t cam_q, cam_t, q, t; } pais_pair_match;

/* The rest of setSeedPatches from a filled match table (the matches of pais_seed_match for every ordered camera pair, in
 * the reference's order: cam_q outer, cam_t inner, q ascending): epipolarLineFiltering (:158-196) with distance bound
 * `max_dist`, filteroutNonMatches (:198-243), the union of :56-82 / setNVMatch (:118-156), and one seed per n-view
 * feature with >= minCamNum views: Patch(0, grey, camIdx, imgPoint) + reCentering (:84-99), appended to `m` as by
 * pais_mvs_add_seed_measured(..., recenter = 1).  *num_seeds: seeds added.  Host only (m may be a GPU-less driver).
 * A (cam_q, cam_t, q, t) tuple given twice is an error: the reference's matcher yields each query at most once per
 * pair, and its erase-while-scanning lists treat duplicates in a way no caller should rely on. */
int  pais_mvs_seeds_from_matches(pais_mvs *m, int num_cams, const pais_keypoints *kp, int num_matches,
                                 const pais_pair_match *matches, double max_dist, int *num_seeds);

/* FeatureManager::setSeedPatches(cameras, maxDist, mvs) from the keypoints on: pais_seed_nearest_all on the driver's GPU,
 * the cross-check of every ordered pair (BFMatcher's crossCheck), then pais_mvs_seeds_from_matches. */
int  pais_mvs_set_seed_patches(pais_mvs *m, int num_cams, const pais_keypoints *kp, int dim, double max_dist, int *num_seeds);

const char *pais_seed_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
