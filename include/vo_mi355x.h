/*
 * vo_mi355x.h -- C ABI of libvo_mi355x.so: the MI355X (gfx950) visual-odometry inner loop.
 *
 * The reference (JonasFrey96/Visual-Odom-Pipeline) is pure Python and has no FFI; its hot path
 * sits behind two Python classes and bottoms out in OpenCV / SciPy calls.  This header declares
 * the flat entry points a ctypes binding uses to replace exactly those calls.  Every export cites
 * the reference call site it replaces (paths relative to /root/reference).  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int32_t: VO_OK (0) or a negative VO_E_* code; vo_last_error(ctx)
 *     gives the message.  No C++ exception crosses the boundary.
 *   - the caller owns all host buffers (C-contiguous numpy arrays); the library owns only the
 *     device memory inside the opaque vo_ctx.  One vo_ctx = one GPU + one HIP stream; calls on
 *     one ctx must be serialised by the caller; different ctxs are independent.
 *   - synchronous entry points return with the outputs written.  The *_async / *_resident forms
 *     only enqueue work on the ctx's stream; vo_sync() waits for it.
 *   - there is NO CPU fallback: without a HIP device vo_ctx_create fails with VO_E_HIP.
 */
#ifndef VO_MI355X_H
#define VO_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_ABI_VERSION 4

enum {
  VO_OK = 0,
  VO_E_INVALID = -1,     /* bad argument */
  VO_E_HIP = -2,         /* HIP runtime error / no device */
  VO_E_NOMEM = -3,
  VO_E_STATE = -4,       /* call sequence error (e.g. tracking before two frames were pushed) */
  VO_E_CAPACITY = -5,    /* a fixed-capacity device buffer would overflow */
  VO_E_NUMERIC = -6      /* solver breakdown (non-positive pivot, non-finite cost) */
};

typedef struct vo_ctx vo_ctx;

/* ---- parameters (defaults = the reference's hard-coded values) ------------------------------ */

/* cv2.calcOpticalFlowPyrLK parameters, src/extractor/extractor.py:16-19 */
typedef struct {
  int32_t win;               /* 31  (winSize 31x31; odd, <= 31)                     */
  int32_t max_level;         /* 3   (=> 4 pyramid levels)                            */
  int32_t max_count;         /* 30  TERM_CRITERIA_COUNT                              */
  double  epsilon;           /* 0.03 TERM_CRITERIA_EPS (squared internally)          */
  float   min_eig_threshold; /* 1e-4 (OpenCV default)                                */
  int32_t _pad;
} vo_klt_params;

/* cv2.goodFeaturesToTrack parameters, src/extractor/extractor.py:21-24 */
typedef struct {
  int32_t max_corners;       /* 1000 */
  int32_t block_size;        /* 31   */
  double  quality_level;     /* 0.03 */
  double  min_distance;      /* min_kp_dist (7 in src/pipeline/pipeline.py:21,27) */
  int32_t use_harris;        /* 0: cv2.goodFeaturesToTrack's default response, the minimum eigenvalue (what the reference runs: its parameter
                                dict extractor.py:21-24 leaves useHarrisDetector at False); 1: useHarrisDetector=True -- the response becomes
                                a c - b^2 - harris_k (a + c)^2 over the same box-filtered Sobel products without the 1/2 factors
                                (imgproc/corner.cpp calcHarris, its SCALAR form: float a c - b^2, the k term in double; a SIMD build of
                                cv2 evaluates all but the last pixels of a row in float throughout and differs by ~1 ulp -- parity is with
                                oracle/vo_oracle.c's restatement, not with cv2 bit for bit), everything after
                                the response map is unchanged.
                                A frame whose largest masked response is not positive yields no corners here (OpenCV would rank negative
                                responses). */
  int32_t _pad;
  double  harris_k;          /* 0.04 (OpenCV default) */
} vo_st_params;

/* BundleAdjuster configuration, src/bundle_adjuster/bundle_adjuster.py:8-16 and
 * src/pipeline/pipeline.py:28-29 (xtol = ftol = 1e-3, loss 'huber', f_scale 1) */
typedef struct {
  int32_t max_iters;         /* LM iterations (linearise + solve + evaluate) cap, e.g. 50 */
  int32_t _pad;
  double  ftol;              /* 1e-3 */
  double  xtol;              /* 1e-3 */
  double  gtol;              /* 1e-8 (scipy default) */
  double  lambda0;           /* initial Marquardt damping, 1e-4 */
  double  huber_delta;       /* 1.0 px */
  double  lambda_min;        /* floor of the damping, 1e-3.  The reference fixes no gauge, so the normal equations
                                have a 7-dimensional near-null space; without a floor the Nielsen schedule drives
                                lambda to ~1e-5 after two good steps and then spends 4-6 iterations on rejected
                                steps along the gauge directions before it has doubled its way back. */
} vo_ba_params;

typedef struct {
  double  cost0;             /* 0.5 * sum rho(|e|^2) at the input */
  double  cost;              /* at the output */
  double  lambda;            /* final damping */
  int32_t iters;             /* LM iterations executed */
  int32_t accepted;          /* accepted steps */
  int32_t status;            /* 1 gtol, 2 ftol, 3 xtol, 0 max_iters, 4 damping overflow, <0 VO_E_* */
  int32_t n_obs;
} vo_ba_stats;

/* cv2.solvePnPRansac parameters as Extractor.camera_pose passes them (src/extractor/extractor.py:182-185;
 * reprojectionError = max_err_reproj = 2.0 from src/pipeline/pipeline.py:124-125) */
typedef struct {
  double  reproj_err;        /* 2.0 px: consensus threshold on the reprojection error */
  double  confidence;        /* 0.9999 */
  int32_t max_iters;         /* 1000000 */
  int32_t seed;              /* of the counter-based sample generator (OpenCV: fixed RNG state) */
} vo_pnp_params;

typedef struct {
  double  cost;              /* sum of squared pixel errors over the consensus set after refinement */
  int32_t n_inliers;
  int32_t hypotheses;        /* evaluated */
  int32_t best;              /* index of the winning hypothesis */
  int32_t status;            /* 0, or VO_E_NUMERIC when no pose with >= 4 inliers was found */
} vo_pnp_stats;

/* cv2.findEssentialMat(p1, p2, K, prob=0.9999, method=RANSAC, threshold=1.0) + cv2.recoverPose(E, p1, p2, K) as
 * Extractor.camera_pose(corr='2D-2D') calls them (src/extractor/extractor.py:162-172) */
typedef struct {
  double  threshold;         /* 1.0 px: consensus threshold on the Sampson distance (divided by (fx + fy) / 2 internally) */
  double  prob;              /* 0.9999 */
  double  distance_thresh;   /* 50: recoverPose keeps triangulated depths in (0, distance_thresh) baselines */
  int32_t max_iters;         /* 1000 (OpenCV's default bound for findEssentialMat) */
  int32_t seed;              /* of the counter-based sample generator */
} vo_ess_params;

typedef struct {
  int32_t n_inliers;         /* consensus set of the returned E */
  int32_t n_good;            /* inliers in front of both cameras for the returned (R, t) */
  int32_t hypotheses;        /* five-point samples evaluated */
  int32_t best;              /* index of the winning sample */
  int32_t status;            /* 0, or VO_E_NUMERIC when no model with >= 5 inliers was found */
  int32_t pad;
} vo_ess_stats;

/* the fields of cv2.KeyPoint that cv2.SIFT fills (the reference reads pt through cv2.KeyPoint_convert and hands the
 * objects back to compute(), src/extractor/extractor.py:114-122) */
typedef struct {
  float   x, y;              /* pt, pixels of the input image */
  float   size;              /* diameter of the meaningful neighbourhood */
  float   angle;             /* degrees, [0, 360) */
  float   response;          /* |contrast| of the refined extremum */
  int32_t octave;            /* packed: octave (low byte, signed), layer (second byte), sub-layer offset (third byte) */
} vo_sift_kp;

/* ---- context -------------------------------------------------------------------------------- */
int32_t vo_abi_version(void);
int32_t vo_device_count(int32_t* n);
/* width/height: image size; max_pts: capacity of the tracked point set (and of DLT batches);
 * max_level / win: pyramid depth and LK window the frame store is built for (3 / 31 in the
 * reference, extractor.py:16-19).  Levels stop early when the next one would be <= win in either
 * dimension, as cv2.buildOpticalFlowPyramid does. */
int32_t vo_ctx_create(int32_t device, int32_t width, int32_t height, int32_t max_pts,
                      int32_t max_level, int32_t win, vo_ctx** out);
/* BATCHED CONTEXT: `batch` independent sequences of identical shape advance in lockstep; every launch serves all of
 * them (the path is launch / latency bound per sequence, batching is what fills the 256 CUs).  With batch > 1 EVERY
 * array argument of every entry point below gains a leading dimension of length `batch` (images [batch][h][w],
 * points [batch][n][2], status [batch][n], X4 [batch][4][n], P0 [batch][12], K [batch][9], obs [batch][W][N][2],
 * out_pts [batch][max_corners][2], n_out [batch], stats [batch], ...); scalars (n, parameters) are shared.
 * vo_ctx_create(...) == vo_ctx_create_batched(..., 1, ...).
 * STREAMS: a context owns its streams (up to five: front end, side, bundle adjustment, copy-in, and a CU-masked form of the first in the gated
 * layout); the library is compiled with -fgpu-default-stream=per-thread, so its SYNCHRONOUS copies (uploads, read-backs, probes) run on the calling
 * thread's own default stream and never join the process's legacy null stream -- a caller that relies on null-stream ordering with its own HIP work
 * must synchronise explicitly (vo_sync). */
int32_t vo_ctx_create_batched(int32_t device, int32_t width, int32_t height, int32_t max_pts,
                              int32_t max_level, int32_t win, int32_t batch, vo_ctx** out);
int32_t vo_ctx_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);
int32_t vo_sync(vo_ctx* ctx);

/* ---- frames: pyramid + Scharr derivatives ---------------------------------------------------
 * Replaces the pyramid / derivative construction inside cv2.calcOpticalFlowPyrLK
 * (extractor.py:44-45,65-66 rebuild it on each of 4 calls per frame; here once per frame).
 * Pushing a frame rotates cur -> prev.  `stride` in bytes (>= width). */
int32_t vo_frame_push(vo_ctx* ctx, const uint8_t* img, int32_t stride);
/* Loader pre-filter (SURVEY.md 8f "next" row 2): cv2.bilateralFilter(img, d=5, sigmaColor=1.5, sigmaSpace=1.5) that
 * Loader.getImage applies to every frame (src/loader/loader.py:16-20,86), fused into the kernel that writes pyramid
 * level 0, so a raw frame can be pushed as read from disk.  d = 0: off (default; the drop-in classes receive frames the
 * reference's loader has already filtered); d < 0: diameter from sigma_space as OpenCV does; diameter <= 7. */
int32_t vo_set_prefilter(vo_ctx* ctx, int32_t d, double sigma_color, double sigma_space);
/* frames preloaded into HBM (bench: inputs resident before the timed region); frames: [batch][n_frames][h][w] */
int32_t vo_seq_upload(vo_ctx* ctx, const uint8_t* frames, int32_t n_frames);
int32_t vo_frame_push_resident(vo_ctx* ctx, int32_t frame_index);           /* async */
/* parity probes: which = 0 prev / 1 cur; img_out (h_l x w_l u8), deriv_out (h_l x w_l x 2 i16), either may be NULL */
int32_t vo_pyramid_level_size(vo_ctx* ctx, int32_t level, int32_t* w, int32_t* h);
int32_t vo_pyramid_read(vo_ctx* ctx, int32_t which, int32_t level, uint8_t* img_out, int16_t* deriv_out);   /* sequence 0 */
int32_t vo_pyramid_read_seq(vo_ctx* ctx, int32_t seq, int32_t which, int32_t level, uint8_t* img_out, int16_t* deriv_out);

/* ---- KLT ------------------------------------------------------------------------------------
 * Replaces cv2.calcOpticalFlowPyrLK(prev, cur, p0, None, **lk_params) at extractor.py:44,65
 * (and, being deterministic, the redundant second call at :45,:66).
 * p0: n x 2 f32.  Outputs: p1 n x 2 f32, status n u8, err n f32, iters n x (max_level+1) i32
 * (iterations run per pyramid level, -1 = level skipped; may be NULL). */
int32_t vo_klt_default_params(vo_klt_params* p);
int32_t vo_klt_track(vo_ctx* ctx, const float* p0, int32_t n, const vo_klt_params* prm,
                     float* p1, uint8_t* status, float* err, int32_t* iters);
/* resident form: the point set lives in HBM and is advanced in place (p <- tracked p) */
int32_t vo_points_upload(vo_ctx* ctx, const float* p, int32_t n);
/* iters (may be NULL): n x (max_level+1) of the last resident track, as in vo_klt_track */
int32_t vo_points_download(vo_ctx* ctx, float* p, uint8_t* status, float* err, int32_t* iters, int32_t n);
int32_t vo_klt_track_resident(vo_ctx* ctx, int32_t n, const vo_klt_params* prm);  /* async */

/* ---- Shi-Tomasi re-detection ----------------------------------------------------------------
 * Replaces the exclusion-mask loop + cv2.goodFeaturesToTrack(img, mask=mask, **shitomasi_params)
 * at extractor.py:103-112 on the CURRENT frame.  cur_pts (n_cur x 2 f32, may be NULL) are the
 * tracked keypoints; discs of `mask_radius` at int32-truncated coordinates are excluded
 * (cv2.circle fill semantics).  `mask` (h x w u8, may be NULL) is an optional explicit mask that
 * is AND-ed with the discs.  out_pts: max_corners x 2 f32 (integer-valued x, y); *n_out set. */
int32_t vo_st_default_params(vo_st_params* p);
int32_t vo_shi_tomasi(vo_ctx* ctx, const float* cur_pts, int32_t n_cur, int32_t mask_radius,
                      const uint8_t* mask, const vo_st_params* prm, float* out_pts, int32_t* n_out);
int32_t vo_shi_tomasi_resident(vo_ctx* ctx, int32_t n_cur, int32_t mask_radius,
                               const vo_st_params* prm);                      /* async; uses the resident points */
int32_t vo_shi_tomasi_fetch(vo_ctx* ctx, float* out_pts, int32_t* n_out);     /* sync + copy out */
/* parity probes: min-eigenvalue map (h x w f32) and the mask actually used (h x w u8) of the last SYNCHRONOUS call
 * (vo_shi_tomasi).  The resident forms keep neither -- the map never leaves the kernel that forms it and the mask is handed
 * back clean for the next frame -- and eig_out / mask_out then return VO_E_STATE (vo_tuning.st_keep_eig makes the
 * resident forms keep them); n_candidates (local maxima above the quality threshold) is always available.
 * More than 16384 candidates are consumed in rank-ordered chunks, like OpenCV's scan; VO_E_CAPACITY only beyond 262144. */
int32_t vo_shi_tomasi_read(vo_ctx* ctx, float* eig_out, uint8_t* mask_out, int32_t* n_candidates);

/* ---- DLT triangulation ----------------------------------------------------------------------
 * Replaces cv2.triangulatePoints(P0, P1, uv0, uv1) at extractor.py:270 and the reprojection
 * statistics TriangulatorNL.refine filters on (src/extractor/triangulate.py:87-111,139).
 * P0, P1: 3x4 f32 row-major (already rounded to f32 as extractor.py:268-269 does);
 * uv0, uv1: n x 2 f32.  X4: 4 x n f32 (OpenCV layout, homogeneous, unit norm).
 * Optional filter statistics (computed in f64 on the f32-rounded, dehomogenised point exactly as
 * the reference does): K 3x3 f64, H0/H1 4x4 f64 row-major -> depth1 n f64 ((H1 [X;1])_z) and
 * reproj n f64 ((|e0| + |e1|) / 2).  Pass K = NULL to skip them. */
int32_t vo_triangulate_dlt(vo_ctx* ctx, const float* P0, const float* P1, const float* uv0,
                           const float* uv1, int32_t n, float* X4,
                           const double* K, const double* H0, const double* H1,
                           double* depth1, double* reproj);

/* resident form for the bench (inputs uploaded once, kernel enqueued per frame, results fetched at the frame's end) */
int32_t vo_dlt_upload(vo_ctx* ctx, const float* P0, const float* P1, const float* uv0, const float* uv1,
                      int32_t n, const double* K, const double* H0, const double* H1);
int32_t vo_dlt_resident(vo_ctx* ctx);                                          /* async */
int32_t vo_dlt_fetch(vo_ctx* ctx, float* X4, double* depth1, double* reproj);

/* ---- sliding-window bundle adjustment -------------------------------------------------------
 * Replaces the scipy.optimize.least_squares call in BundleAdjuster.adjust
 * (src/bundle_adjuster/bundle_adjuster.py:189-194) and its objective (:18-65).
 * Problem layout (float64, C-contiguous):
 *   K       3x3
 *   poses   W x 6   (rvec, tvec) world->camera; slot 0 = newest frame (as :169-176)
 *   points  N x 3
 *   obs     W x N x 2 pixel observations; NaN in obs[i][j][0] = landmark j not seen in slot i
 * Same cost as the reference (Huber on the per-observation pixel-error norm, f_scale 1, no gauge
 * fixing); solver = analytic Jacobian, IRLS-weighted J^T J, landmark Schur complement (f64 MFMA
 * SYRK), dense Cholesky, Levenberg-Marquardt -- see DESIGN.md. */
int32_t vo_ba_default_params(vo_ba_params* p);
int32_t vo_ba_adjust(vo_ctx* ctx, const double* K, const double* poses, const double* points,
                     const double* obs, int32_t n_slots, int32_t n_pts, const vo_ba_params* prm,
                     double* poses_out, double* points_out, vo_ba_stats* stats);
/* resident form for the bench: upload once, solve repeatedly from the same x0 */
int32_t vo_ba_upload(vo_ctx* ctx, const double* K, const double* poses, const double* points,
                     const double* obs, int32_t n_slots, int32_t n_pts);
int32_t vo_ba_solve_resident(vo_ctx* ctx, const vo_ba_params* prm);          /* async */
/* a BANK of n_problems resident problems of one shape, one selected per solve (host-side pointer switch, nothing enqueued): a sliding
 * window never presents the same problem twice, so the bench's sequences cycle through distinct problems without an upload in the loop.
 * poses [n_problems][batch][W][6], points [n_problems][batch][N][3], obs [n_problems][batch][W][N][2]; K [batch][9] */
int32_t vo_ba_upload_bank(vo_ctx* ctx, const double* K, const double* poses, const double* points, const double* obs,
                          int32_t n_slots, int32_t n_pts, int32_t n_problems);
int32_t vo_ba_select_problem(vo_ctx* ctx, int32_t k);
int32_t vo_ba_fetch(vo_ctx* ctx, double* poses_out, double* points_out, vo_ba_stats* stats);
/* parity probes at the uploaded x0 (no iteration):
 *   residual: m f64 in the reference's order (slot-major, ascending landmark; :18-65)
 *   normal equations with Huber IRLS weights: Hpp W x 6 x 6, gp W x 6, Hll N x 3 x 3, gl N x 3,
 *   reduced camera system at damping lambda: S 6W x 6W, rhs 6W; one LM step dposes W x 6, dpoints N x 3.
 *   Any output pointer may be NULL. */
int32_t vo_ba_probe(vo_ctx* ctx, double lambda, double huber_delta, double* residual, int32_t* n_obs,
                    double* cost, double* Hpp, double* gp, double* Hll, double* gl, double* S,
                    double* rhs, double* dposes, double* dpoints);

/* ---- landmark-sharded bundle adjustment of ONE problem (BASELINE config 5) -------
 * The reference has no multi-device path (single Python process, src/pipeline/pipeline.py:155-156 calls adjust once
 * per frame); this is the exchange step north_star names ("RCCL over xGMI only for the shared-landmark BA
 * reduction").  The landmarks of one window are dealt round-robin to S = n_ranks x batch shards: shard s = rank *
 * batch + b owns landmarks j with j % S == s together with ALL their observations; the W poses are replicated.
 * Every shard eliminates its own landmarks; the shared quantity is the reduced camera system every landmark
 * contributes to.  Per LM iteration: ONE in-place RCCL all-reduce (sum, f64) of the packed
 * [Gram tiles of Y^ (E and r) | camera sums H_pp, g_p, cost | per-rank max|g_l| slots] (23 KB at W = 10, 78 KB at
 * W = 20) and one of the 4 step statistics; every rank then solves the 6W x 6W system redundantly and takes the
 * same accept/reject decision.  The batch dimension of a batched context acts as `batch` shards on one GPU (summed
 * by a kernel) -- useful on its own for tests and for filling one GPU with a single large problem.
 *   vo_comm_unique_id: 128-byte RCCL id made by rank 0; distribute it to the other ranks by any means.
 *   vo_comm_init:      one communicator per context (one process per GPU); n_ranks = 1 is valid.
 *   vo_ba_set_sharded: on = 1: the `batch` problems of vo_ba_upload / vo_ba_adjust are shards of one problem
 *                      (same K and poses in every entry, shard-local points / obs, all shards padded to the same
 *                      N with unobserved landmarks at the origin).
 *   vo_ba_gather_points: points of every shard of every rank after a solve, [n_ranks][batch][N][3]. */
#define VO_COMM_ID_BYTES 128
#define VO_COMM_MAX_RANKS 16
int32_t vo_comm_unique_id(uint8_t* id_out /* VO_COMM_ID_BYTES */);
int32_t vo_comm_init(vo_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* id);
int32_t vo_comm_destroy(vo_ctx* ctx);
int32_t vo_ba_set_sharded(vo_ctx* ctx, int32_t on);
int32_t vo_ba_gather_points(vo_ctx* ctx, double* points_all);

/* ---- 3D-2D pose (SURVEY.md 8f "next" row 1) ---------------------------------------------------
 * Replaces cv2.solvePnPRansac(pts3d, pts2d, K, None, reprojectionError, iterationsCount, confidence) in
 * Extractor.camera_pose(corr='3D-2D') (src/extractor/extractor.py:174-191): RANSAC over P3P hypotheses (one wave per
 * hypothesis, 256 per batch and sequence, iteration bound updated like OpenCV's RANSACUpdateNumIters), consensus set
 * = squared reprojection error <= reproj_err^2, Gauss-Newton refinement of (rvec, tvec) over the consensus set.
 * OpenCV's sample sequence cannot be reproduced (own RNG, EPnP on 5 points): parity is statistical -- same consensus
 * set and minimiser whenever the inlier set is unambiguous; the algorithm itself is defined by oracle/pnp_oracle.py.
 * K [batch][9]; pts3d [batch][n][3], pts2d [batch][n][2] f32 (NaN rows are never inliers);
 * rvec, tvec [batch][3] f64 (x_cam = R(rvec) X + tvec); inlier_mask [batch][n] u8 (may be NULL); stats [batch]. */
int32_t vo_pnp_default_params(vo_pnp_params* p);
int32_t vo_pnp_ransac(vo_ctx* ctx, const double* K, const float* pts3d, const float* pts2d, int32_t n,
                      const vo_pnp_params* prm, double* rvec, double* tvec, uint8_t* inlier_mask, vo_pnp_stats* stats);
/* resident form: correspondences uploaded once, a solve enqueued per frame with no host synchronisation (`blind_batches`
 * batches of 32, then 256 hypotheses each, early-exiting once the iteration bound is reached: 2 cover down to ~42 % inliers), results
 * fetched later; a sequence whose bound was not reached reports status VO_E_CAPACITY (pose = best so far). */
int32_t vo_pnp_upload(vo_ctx* ctx, const double* K, const float* pts3d, const float* pts2d, int32_t n);
int32_t vo_pnp_solve_resident(vo_ctx* ctx, const vo_pnp_params* prm, int32_t blind_batches);        /* async */
int32_t vo_pnp_fetch(vo_ctx* ctx, double* rvec, double* tvec, uint8_t* inlier_mask, vo_pnp_stats* stats);

/* ---- 2D-2D bootstrap pose (SURVEY.md 8f "next" row 4, pose part) ---------------------------------
 * Replaces cv2.findEssentialMat(..., method=RANSAC) + cv2.recoverPose in Extractor.camera_pose(corr='2D-2D')
 * (src/extractor/extractor.py:162-172; Pipeline._get_init_state, src/pipeline/pipeline.py:63): RANSAC over Nister
 * five-point essential matrices (a lane per sample, <= 10 models each, a wave per model for the consensus count),
 * consensus = squared Sampson distance <= (threshold / mean focal length)^2, iteration bound as RANSACUpdateNumIters
 * with 5 model points; the best model is returned as found (OpenCV does not re-fit); then the four (R, t)
 * candidates of E and OpenCV's cheirality vote by DLT triangulation of the inliers.  Sample draws differ from
 * OpenCV's: statistical parity; the algorithm is defined by oracle/essential_oracle.py.
 * K [batch][9]; pts1, pts2 [batch][n][2] f32 pixels (NaN rows are never inliers); E [batch][9] (unit Frobenius norm,
 * may be NULL), R [batch][9], t [batch][3] (|t| = 1, x2 ~ R x1 + t); inlier_mask [batch][n] u8 (may be NULL). */
int32_t vo_essential_default_params(vo_ess_params* p);
int32_t vo_essential_ransac(vo_ctx* ctx, const double* K, const float* pts1, const float* pts2, int32_t n,
                            const vo_ess_params* prm, double* E, double* R, double* t, uint8_t* inlier_mask,
                            vo_ess_stats* stats);

/* ---- SIFT features (SURVEY.md 8f "next" row 4, feature part) -----------------------------------
 * Replaces cv2.SIFT_create(nfeatures=1000).detect(img, mask) + .compute(img, kps) in
 * Extractor.extract(detector='custom', describe=True) (src/extractor/extractor.py:26-28, 114-122; the two bootstrap
 * frames of Pipeline._get_init_state, src/pipeline/pipeline.py:48-49): doubled base image, Gaussian / DoG scale space
 * (3 layers per octave, sigma 1.6), 26-neighbour extrema, quadratic refinement with contrast (0.04) and edge (10)
 * tests, orientation histogram peaks, removeDuplicatedSorted + retainBest(nfeatures), 4 x 4 x 8 descriptors scaled to
 * 0..255.  The float32 operation order is defined by oracle/sift_oracle.py (parity with OpenCV itself is unpinned).
 * img [batch][height][stride] u8 (the context's image size); mask the same layout or NULL (0 = drop the keypoint);
 * kps [batch][max_out]; desc [batch][max_out][128] f32; n_out [batch].  VO_E_CAPACITY if max_out is too small. */
int32_t vo_sift_detect_compute(vo_ctx* ctx, const uint8_t* img, int32_t stride, const uint8_t* mask, int32_t nfeatures,
                               int32_t max_out, vo_sift_kp* kps, float* desc, int32_t* n_out);

/* ---- descriptor matching (SURVEY.md 8f "next" row 4, matching part) -----------------------------
 * Replaces cv2.BFMatcher().knnMatch(desc_1, desc_2, k=2) in Extractor.match (src/extractor/extractor.py:134-145;
 * match_lists :147-154, Pipeline._get_init_state src/pipeline/pipeline.py:52): for every query descriptor the two
 * nearest train descriptors under the L2 norm, ordered by (distance, train index).  The ratio test stays with the caller.
 * desc1 [batch][n1][dim], desc2 [batch][n2][dim] f32; idx [batch][n1][2] (-1 = no such neighbour), dist [batch][n1][2] f32. */
int32_t vo_match_knn2(vo_ctx* ctx, const float* desc1, int32_t n1, const float* desc2, int32_t n2, int32_t dim,
                      int32_t* idx, float* dist);

/* ---- device-resident track table (SURVEY.md 8f "next" row 3) -----------------------------------
 * The bookkeeping Extractor.extend_tracks / extend_landmarks / extract do on Python lists of Keypoint objects
 * (src/extractor/extractor.py:38-88, 90-132; src/state/keypoint.py:4-21), as a structure of arrays in HBM: per
 * sequence an ORDERED list (survivors keep their order, detections are appended) of uv (the resident point set),
 * uv_first, t_first, t_total, a stable tag, and a ring of the last 32 positions by absolute frame index.
 *   vo_tracks_seed    initial tracks born at frame t (t_total = 1, history = [uv]);  pts [batch][n][2]
 *   vo_tracks_track   KLT prev -> cur of every live track, then the reference's rule: keep iff 0 <= x <= W and
 *                     0 <= y <= H (ends included; KLT status ignored, bidirectional test off: pipeline.py:98-100);
 *                     survivors: uv, t_total + 1, history append; the others go to the dead list        (async)
 *   vo_tracks_detect  exclusion discs at the live tracks + Shi-Tomasi on the current frame, up to max_new corners per
 *                     sequence appended as tracks born at t (extractor.py:103-132)                      (async)
 *   vo_tracks_read    synchronous read-back: n / n_dead [batch]; uv, uv_first [batch][max_pts][2]; t_first, t_total,
 *                     tag, dead_tag [batch][max_pts]; any pointer may be NULL
 *   vo_tracks_obs     the bundle adjuster's observation table of the live tracks (bundle_adjuster.py:150-158 with
 *                     t_latest = t_now): obs [batch][window][max_pts][2] f64, slot s <-> frame t_now - s, NaN = not
 *                     observed -- the layout vo_ba_upload / vo_ba_adjust take (N = max_pts)
 * Counts differ per sequence of a batch; they stay on the device and the KLT / disc kernels skip the rest. */
int32_t vo_tracks_seed(vo_ctx* ctx, const float* pts, int32_t n, int32_t t);
int32_t vo_tracks_track(vo_ctx* ctx, int32_t t, const vo_klt_params* prm);
int32_t vo_tracks_detect(vo_ctx* ctx, int32_t t, int32_t mask_radius, const vo_st_params* st, int32_t max_new);
int32_t vo_tracks_read(vo_ctx* ctx, int32_t* n, float* uv, float* uv_first, int32_t* t_first, int32_t* t_total,
                       int32_t* tag, int32_t* n_dead, int32_t* dead_tag);
int32_t vo_tracks_obs(vo_ctx* ctx, int32_t t_now, int32_t window, double* obs);
/* the same table written into the RESIDENT BA problem (uploaded with N = max_pts landmarks): no host round trip (async) */
int32_t vo_ba_obs_from_tracks(vo_ctx* ctx, int32_t t_now);

/* ---- closed-loop Pipeline.step on the device (SURVEY.md 8f "next" row 3, the State half) ---------------------------
 * The whole per-frame state the reference keeps in Python lists of objects -- State(landmarks, landmarks_kp, candidates_kp,
 * trajectory) (src/state/state.py:4-10) and Pipeline._landmarks_dead / _landmarks_kp_dead (src/pipeline/pipeline.py:31) -- as
 * device tables, and Pipeline.step (pipeline.py:92-167) as ONE enqueue per frame with every data dependence on the device:
 *   TRACK        pyramid + KLT of all landmark and candidate keypoints, the keep rule and bookkeeping of
 *                Extractor.extend_tracks / extend_landmarks (extractor.py:38-88), what dies goes to the dead lists (pipeline.py:98-103)
 *   POSE         RANSAC-P3P pose from the landmarks' 3-D points and tracked pixels (extractor.py:174-191), non-inliers to the dead
 *                lists (pipeline.py:124-137), pose appended to the trajectory (:140)
 *   TRIANGULATE  Extractor.triangulate_tracks (extractor.py:193-277): candidates with t_total >= min_track_length leave the
 *                candidate list, DLT between their first and newest observation, cheirality + reprojection filters
 *                (triangulate.py:87-111), the per-group gate of extractor.py:231-240, promotion to landmarks
 *   ADJUST       BundleAdjuster.adjust (bundle_adjuster.py:127-215): dead landmarks whose track lies inside the window are
 *                appended to the state's lists again AS THE SAME OBJECTS (:142-150), observation table from the keypoint
 *                histories (:153-158), x0 from the landmark positions and the window's poses (:165-176), LM solve, positions and
 *                poses written back (:197-213)
 *   DETECT       exclusion discs at every keypoint of the state + Shi-Tomasi, corners appended as candidates (pipeline.py:159-163)
 * Object identity matters in the reference (adjust appends without copying, extend_landmarks deep-copies the keypoint but keeps the
 * landmark object, Pipeline.step deep-copies what dies): the tables are OBJECT rows with stable indices -- K rows (Keypoint:
 * t_first, t_total, uv_first, uv, history length, ring of the last 32 history entries by index) and L rows (Landmark: t_latest, p) --
 * plus ordered lists of row indices (candidates [K]; landmarks [(L, K, keypoint-shared-with-a-dead-entry)]; dead [(L, K)]), so
 * several list entries can refer to one object exactly as in the reference.  Dead entries that can never be resurrected again
 * are dropped and counted.  oracle/pipe_oracle.py restates the algorithm; tests/test_gpu_pipe.py compares the tables with the
 * reference's loop over Python objects frame by frame.
 * Capacity: candidates + landmarks <= max_pts (they share the KLT point buffer), dead list <= max_pts, object rows 4 x max_pts.
 * When a list is full, detections / promotions / resurrections are cut in list order and the record's `overflow` bits say so
 * (the reference's lists are unbounded).
 * Frames come from the resident sequence (vo_seq_upload) by index, or frame_idx = -1: the caller has pushed the frame
 * (vo_frame_push).  Up to VO_PIPE_INFLIGHT steps may be enqueued before the oldest record is fetched; nothing else crosses
 * the bus per frame. */
typedef struct {
  int32_t ba_window;          /* 4    pipeline.py:19  (<= 20) */
  int32_t min_track_length;   /* 3    pipeline.py:147 */
  int32_t mask_radius;        /* 7    pipeline.py:162 (min_kp_dist) */
  int32_t max_new;            /* 1000 corners appended per frame (maxCorners, extractor.py:21) */
  int32_t pnp_blind_batches;  /* 4    batches of P3P hypotheses (32, then 256 each) enqueued per frame (they exit early once the RANSAC bound is reached) */
  int32_t ba_budget;          /* LM iterations enqueued per frame, <= ba.max_iters (they exit early once the LM has stopped) */
  int32_t resurrect;          /* 1: as the reference -- adjust appends dead landmarks whose track lies inside the window to the state's lists again
                                 (bundle_adjuster.py:142-150).  With the reference's window of 4 that is a handful per frame;
                                 with a window of 10 every young death comes back every frame, and again for every copy that dies again, until the
                                 table holds little else (measured: bench.py --workload pipeline).  0: dead landmarks stay dead. */
  int32_t reserved;
  double  max_reproj_err;     /* 2.0  pipeline.py:23 (PnP consensus and triangulation filter) */
  double  min_bearing_angle;  /* 0.5  pipeline.py:24 */
  vo_klt_params klt;
  vo_st_params  st;
  vo_ba_params  ba;
  vo_pnp_params pnp;
} vo_pipe_params;

/* what comes back per sequence and frame */
typedef struct {
  int32_t t;                  /* step index after this frame (Pipeline._t_step) */
  int32_t status;             /* 0, or VO_PIPE_* bits: the sequence stopped at the frame that set them */
  int32_t overflow;           /* capacity policy acted: 1 dead list, 2 promotion, 4 resurrection, 8 detection, 16 Shi-Tomasi candidate capacity */
  int32_t n_landmarks, n_candidates;      /* len(state._landmarks), len(state._candidates_kp) after the frame */
  int32_t n_dead, n_dead_total;           /* dead entries kept on the device / ever (= len(Pipeline._landmarks_dead)) */
  int32_t n_tracked;                      /* keypoints that went into the KLT of this frame */
  int32_t pnp_inliers, pnp_hypotheses, pnp_bound_reached;
  int32_t n_ripe, n_new, n_resurrected, n_detected;
  int32_t ba_landmarks, ba_observations, ba_iters, ba_accepted, ba_status, ba_done;   /* ba_done 0: the budget cut the solve */
  int32_t t_final;            /* step whose pose has just left the BA window (t - ba_window + 1, or -1): H_final will not change any more */
  double  ba_cost0, ba_cost;
  double  H[12];              /* pose of this frame AFTER the adjust: rows of [R | t], world -> camera (later adjusts refine it while it is in the window) */
  double  H_final[12];        /* pose of step t_final: collecting these gives the trajectory the reference ends up with (state._trajectory) */
} vo_pipe_record;

enum { VO_PIPE_LOST = 1, VO_PIPE_CAPACITY = 2, VO_PIPE_GROUPS = 4 };   /* LOST: the 3D-2D pose found no consensus (the reference crashes there); CAPACITY: object rows exhausted; GROUPS: a ripe candidate was born more than 32 frames ago (its pose has left the trajectory ring) */
enum { VO_PIPE_TRACK = 1, VO_PIPE_POSE = 2, VO_PIPE_TRIANGULATE = 4, VO_PIPE_ADJUST = 8, VO_PIPE_DETECT = 16, VO_PIPE_ALL = 31,
       /* the two halves of TRACK's bookkeeping as the reference calls them (extractor.py:38-59 then :61-88): VO_PIPE_TRACK | VO_PIPE_TRACK_CANDIDATES
          = pyramid + KLT of every keypoint + extend_tracks; then VO_PIPE_TRACK_LANDMARKS alone = extend_landmarks on the same tracked set (the
          step counter advances here).  Neither bit: both halves (the closed loop).  Used by the object boundary, vo_mi355x/lazy.py */
       VO_PIPE_TRACK_CANDIDATES = 32, VO_PIPE_TRACK_LANDMARKS = 64,
       /* a stage-wise caller: rows that left the lists in this call are NOT handed out again yet (the free lists are rebuilt by the first later call
          without this bit, normally the frame's DETECT) -- so their contents can still be read back after the call (vo_pipe_rows_read) */
       VO_PIPE_KEEP_FREE_LISTS = 128 };
#define VO_PIPE_INFLIGHT 4
#define VO_PIPE_HIST 32

/* the device tables, for seeding and read-back (all with a leading [batch] dimension; N = max_pts, R = 4 x max_pts rows):
 * K rows: K_TFIRST, K_TTOTAL, K_HISTLEN i32 [R]; K_UV, K_UVFIRST f32 [R][2]; K_HIST f32 [32][R][2] (entry idx in slot idx % 32)
 * L rows: L_TLATEST i32 [R]; L_P f64 [R][3]
 * lists:  CAND i32 [N]; LM_L, LM_K, LM_KSHARED i32 [N]; DEAD_L, DEAD_K i32 [N]
 * COUNTS i32 [32]: [0] n_cand [1] n_lm [2] n_dead [3] n_dead_dropped [4] status [5] t
 * POSES f64 [32][12]: trajectory ring, pose of step t in slot t % 32 */
enum { VO_PIPE_K_TFIRST = 0, VO_PIPE_K_TTOTAL, VO_PIPE_K_HISTLEN, VO_PIPE_K_UV, VO_PIPE_K_UVFIRST, VO_PIPE_K_HIST, VO_PIPE_L_TLATEST,
       VO_PIPE_L_P, VO_PIPE_CAND, VO_PIPE_LM_L, VO_PIPE_LM_K, VO_PIPE_LM_KSHARED, VO_PIPE_DEAD_L, VO_PIPE_DEAD_K, VO_PIPE_COUNTS,
       VO_PIPE_POSES, VO_PIPE_N_TABLES };

int32_t vo_pipe_default_params(vo_pipe_params* p);
/* K [batch][9].  Allocates the tables (empty state, t = 0) and the BA / PnP / DLT workspaces for max_pts landmark slots. */
int32_t vo_pipe_create(vo_ctx* ctx, const double* K, const vo_pipe_params* prm);
int32_t vo_pipe_table_bytes(vo_ctx* ctx, int32_t which, uint64_t* bytes);
int32_t vo_pipe_table_write(vo_ctx* ctx, int32_t which, const void* src);     /* synchronous; needs no steps in flight */
int32_t vo_pipe_table_read(vo_ctx* ctx, int32_t which, void* dst);            /* synchronous */
/* after the tables were written: free rows and the resident point set are rebuilt from the lists */
int32_t vo_pipe_commit(vo_ctx* ctx);
/* one frame; stages = VO_PIPE_ALL, or a subset for stage-wise parity tests (the step counter advances in TRACK)   (async).
 * frame_idx >= 0: frame of the uploaded sequence (vo_seq_upload); < 0: the caller has pushed the frame (vo_frame_push*).
 * Up to VO_PIPE_INFLIGHT steps may be enqueued before the oldest is fetched.  With a side stream (vo_set_side_stream != 0, the
 * default) the re-detection and spawn of frame t and the pyramid + KLT of frame t + 1 (frame_idx >= 0) run beside the bundle
 * adjustment of frame t; results are bit-identical to the one-stream order.  Many sequences: ONE context with a large batch. */
int32_t vo_pipe_step(vo_ctx* ctx, int32_t frame_idx, int32_t stages);
/* the same step with the frame handed over by the host, as the reference's loop does (Pipeline.step(img): src/pipeline/pipeline.py:98,171-172):
 * frames[b] = the new image of sequence b (rows of `stride` bytes).  Upload on the copy stream (see vo_frame_step_host: one gather launch for
 * page-locked images), pyramid + tracking on the side stream behind it -- the overlap of a resident frame.  `stages` must contain VO_PIPE_TRACK.
 * Page-locked images must stay untouched until the step has been fetched. */
int32_t vo_pipe_step_host(vo_ctx* ctx, const uint8_t* const* frames, int32_t stride, int32_t stages);
int32_t vo_pipe_fetch(vo_ctx* ctx, vo_pipe_record* rec /* [batch] */);        /* waits for the OLDEST step not fetched yet */
int32_t vo_pipe_set_ba_budget(vo_ctx* ctx, int32_t budget);
/* Read-backs for the object boundary (vo_mi355x/lazy.py: the reference's Extractor / BundleAdjuster interface, src/extractor/extractor.py:38-277 and
 * src/bundle_adjuster/bundle_adjuster.py:127-215, as views of these tables).  All synchronous, no step may be in flight.
 * lists_read: tables VO_PIPE_CAND .. VO_PIPE_POSES in ONE copy, as they lie on the device -- each [batch][...], each padded to a multiple of 256 bytes.
 * rows_read: object rows by index, kind 0 = K rows -> {i32 t_first, t_total, hist_len, 0; f32 uv[2], uv_first[2]; f32 hist[32][2]} (288 bytes, ring
 * slot order), kind 1 = L rows -> {i32 t_latest, 0; f64 p[3]} (32 bytes); rows [batch][n], n <= max_pts.
 * inliers_read: the consensus mask of the last POSE stage over the landmark list as it was before the pruning, [batch][n]. */
int32_t vo_pipe_lists_bytes(vo_ctx* ctx, uint64_t* bytes);
int32_t vo_pipe_lists_read(vo_ctx* ctx, void* dst);
int32_t vo_pipe_rows_read(vo_ctx* ctx, int32_t kind, const int32_t* rows, int32_t n, void* out);
int32_t vo_pipe_inliers_read(vo_ctx* ctx, uint8_t* mask, int32_t n);

/* ---- forced forms (parity tests, A/B measurements) ---------------------------------------------
 * Every kernel choice the library makes by a rule (which bundle-adjustment family, how a problem is spread over workgroups, which form of the
 * eigenvalue pass, the stream layout's gate ...) can be forced per context; 0 in a field = the rule.  This replaces the process-wide VO_* environment
 * switches of rounds 1-5: the library reads no tuning from the environment any more.  Results do not depend on any of these fields except in the
 * last bits of a bundle adjustment (its partial sums are folded in another order: ~1e-12).  Set it with nothing in flight; the bundle-adjustment
 * fields marked (upload) shape the workspace and take effect at the next vo_ba_upload / vo_ba_upload_bank / vo_pipe_create. */
typedef struct {
  int32_t ba_kernels;          /* (upload) 1: lane-per-observation kernels (k_ba_build<>); 2: wave-private kernels (k_ba_build_w<>, windows <= 10) */
  int32_t ba_lanes;            /* (upload) lanes per landmark: 8 = wave-private kernels at windows of 9-10 slots (rule: 5); 16 = lane-per-observation
                                  kernels at windows <= 8 (rule: 8) */
  int32_t ba_threads;          /* (upload) lane-per-observation kernels: workgroup size 256 / 512 / 1024 */
  int32_t ba_pitch_pad;        /* (upload) lane-per-observation kernels: panel pitch = rows + (ba_pitch_pad - 1) doubles */
  int32_t ba_chunks;           /* lane-per-observation kernels: landmark chunks per workgroup */
  int32_t ba_workgroups;       /* wave-private kernels: workgroups per problem in a full launch */
  int32_t ba_workgroup_cap;    /* wave-private kernels: most workgroups a running problem gets once others have finished (rule: 16) */
  int32_t ba_fold;             /* 1: a reduce kernel folds the partial sets; 2: k_ba_solve folds them itself */
  int32_t klt_waves;           /* occupancy bound the tracker is compiled for: 4 / 5 / 6 waves per SIMD (rule: 6) */
  int32_t klt_pair;            /* builds with -DVO_EXPERIMENTS only: 3 / 4 / 5 = two keypoints per wave (k_klt_track2) */
  int32_t st_two_kernels;      /* 1: Sobel + row sums and column sums + eigenvalue as two kernels at block size 31 too (rule: fused) */
  int32_t st_band_rows;        /* rows per band of the fused eigenvalue kernel */
  int32_t st_separate_nms;     /* 1: eigenvalue map through HBM to a separate non-maximum-suppression kernel */
  int32_t st_keep_eig;         /* 1: resident launches keep the eigenvalue map and the mask (vo_shi_tomasi_read after them) */
  int32_t st_host_limit;       /* 1: closed loop: the corner limit of a sequence is not read on the device */
  int32_t xcd_remap_off;       /* 1: plain (block, sequence) order instead of one sequence per XCD */
  int32_t gate_groups;         /* stream layout 2: LM launch groups of frame t ahead of the tracker launch of frame t + 1; -1: none */
  int32_t reserve_cus;         /* stream layout 2: compute units the front-end stream leaves free; -1: none */
  int32_t gather_workgroups;   /* frames from the host: workgroups of k_gather_frames (rule: 2 per image in the gated layout of a batch, else 32) */
  int32_t reserved[13];
} vo_tuning;
int32_t vo_get_tuning(vo_ctx* ctx, vo_tuning* out);
int32_t vo_set_tuning(vo_ctx* ctx, const vo_tuning* t);

/* ---- fused per-frame step on resident data ---------------------------------------------------
 * One call enqueues the hot path of one frame in the order of Pipeline.step (src/pipeline/pipeline.py:92-167):
 * frame `frame_idx` of the uploaded sequence -> pyramid/Scharr -> KLT of the resident points -> [DLT of the
 * uploaded pairs] -> [BA of the uploaded problem] -> [Shi-Tomasi re-detection around the tracked points] ->
 * result copies.  With vo_set_graph_mode(ctx, 1) the launch sequence is captured once per buffer parity and replayed
 * as a hipGraph (bit-identical results; off by default: on ROCm 7.2 the replay costs more than ~45 plain launches).
 * Up to TWO steps may be in flight (in graph mode too: a capture is keyed by the mirror half it bakes in): step t + 1 can be enqueued before step t is fetched, the
 * results of consecutive steps land in alternating pinned mirrors, so the GPU queue never drains while the host
 * unpacks.  vo_frame_fetch waits for the OLDEST step not fetched yet (an event, not the whole stream) and unpacks it;
 * with nothing in flight it returns the last step's results again. */
int32_t vo_frame_step_resident(vo_ctx* ctx, int32_t frame_idx, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                               int32_t do_st, int32_t mask_radius, const vo_klt_params* klt,
                               const vo_st_params* st, const vo_ba_params* ba);               /* async */
int32_t vo_frame_fetch(vo_ctx* ctx, int32_t n_pts, float* p, uint8_t* status, float* err, float* X4,
                       double* depth1, double* reproj, double* poses, double* points, vo_ba_stats* stats,
                       float* corners, int32_t* n_corners);
/* The same step with the frames handed over BY THE HOST, as the reference's loop does (Loader.next -> Pipeline.step(img): src/loader/loader.py:86,
 * src/pipeline/pipeline.py:98,171-172): frames[b] = this step's image of sequence b, `height` rows of `stride` bytes (>= width), uint8.  The
 * upload runs on a copy stream of its own into a device buffer double-buffered by step parity, the pyramid of the step waits for it by an event:
 * with two steps in flight the upload of frame t + 1 overlaps the bundle adjustment of frame t.  Images that follow each other in host memory (a
 * [batch][height][width] array) travel as one copy.  Pinned memory (vo_host_alloc, or registered by the caller) is read by DMA asynchronously and
 * must stay untouched until vo_frame_fetch has returned this step; pageable memory is consumed before the call returns (staged by the runtime:
 * slower).  Results are bit-identical to vo_frame_step_resident on the same frames.  Plain launches (a captured graph bakes its source in). */
int32_t vo_frame_step_host(vo_ctx* ctx, const uint8_t* const* frames, int32_t stride, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                           int32_t do_st, int32_t mask_radius, const vo_klt_params* klt,
                           const vo_st_params* st, const vo_ba_params* ba);                    /* async */
/* page-locked host memory for frames a loader decodes into (numpy arrays over it: VoContext.host_alloc) */
int32_t vo_host_alloc(uint64_t bytes, void** out);
int32_t vo_host_free(void* p);
int32_t vo_set_graph_mode(vo_ctx* ctx, int32_t on);
/* Stream layout of vo_frame_step_resident.  0: one stream.  1 (default): re-detection +
 * triangulation on a side stream beside the bundle adjustment (+10-20 % for one context, +1-2 % with three).  2: pipelined, three
 * streams -- pyramid + KLT | re-detection + triangulation | bundle adjustment -- so that the bundle adjustment of frame t also runs
 * beside the front end of frame t + 1 when two steps are in flight (ONE sequence: 3 600 -> 4 500 frames/s; three batched contexts lose
 * 2 %).  Results are identical in every layout.  Fetch the steps in flight first.
 * Layout 2 with a batch of >= 8 sequences: the tracker's launch (one wave per keypoint, 512 000 workgroups at a batch of 256) takes every free
 * wave slot for its whole duration and the previous frame's LM chain would stand still beside it, so (a) the tracker of frame t + 1 is
 * enqueued behind the first `gate_groups` LM launch groups of frame t -- the groups in which (nearly) all problems still run get the whole
 * chip -- and (b) the ctx stream is re-created as a queue that leaves `reserved_cus` compute units free (one per shader engine and XCD), on
 * which the chain's narrow tail groups run beside the tracker.  +4 % at 256 sequences, +10-14 % at 8-192; a longer LM budget costs (almost)
 * nothing more.  vo_step_layout reports what is in effect; vo_tuning.gate_groups / reserve_cus force other values.  Graph
 * replay and vo_pipe_step run on the plain, unmasked stream. */
int32_t vo_set_side_stream(vo_ctx* ctx, int32_t on);
int32_t vo_step_layout(vo_ctx* ctx, int32_t* layout, int32_t* gate_groups, int32_t* reserved_cus);

/* ---- in-stream timing (hipEvent pairs recorded on the ctx stream around a region's launches) -----
 * Used by bench.py for the roofline figure: region VO_PROF_KLT brackets exactly the k_klt_track launch.
 * vo_profile_read synchronises the stream and returns the summed elapsed time and the number of
 * recorded regions since vo_profile_enable(ctx, mask); mask = OR of (1 << region), 0 = off. */
enum { VO_PROF_FRAME = 0, VO_PROF_KLT = 1, VO_PROF_ST = 2, VO_PROF_DLT = 3, VO_PROF_BA = 4, VO_PROF_COUNT = 5 };
int32_t vo_profile_enable(vo_ctx* ctx, int32_t region_mask);
int32_t vo_profile_read(vo_ctx* ctx, int32_t region, double* total_ms, int32_t* count);
/* diagnostic: shader-clock stamps (s_memtime deltas, cycles) of the phases of the last launch of a
 * single-workgroup / critical-path kernel.  which: 0 = k_st_select, 1 = k_ba_solve, 2 = k_ba_build (workgroup 0), 3 = k_klt_track (one wave).
 * out8 receives 8 values (unused entries 0).  Synchronises the stream. */
int32_t vo_debug_cycles(vo_ctx* ctx, int32_t which, int64_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* VO_MI355X_H */
