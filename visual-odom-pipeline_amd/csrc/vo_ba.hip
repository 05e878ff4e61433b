// Sliding-window bundle adjustment on gfx950: analytic Jacobians, IRLS-weighted J^T J, landmark Schur
// complement as an f64-MFMA SYRK, 6x6-blocked Cholesky in LDS, Levenberg-Marquardt with the accept/reject
// logic on the device (no host round trip inside the solve).
//
// Replaces the scipy.optimize.least_squares(...) call of BundleAdjuster.adjust,
// /root/reference/src/bundle_adjuster/bundle_adjuster.py:189-194, and its objective :18-65 / :68-83.
// Same cost as the reference (Huber, f_scale 1, on the per-observation pixel-error norm; no gauge fixing);
// the solver is the one `north_star` asks for and oracle/ba_oracle.py defines (the tests compare against it
// step by step): residual e (2), blocks J_p (2x6), J_l (2x3), w = rho'(|e|^2),
//   H = sum w J^T J, g = sum w J^T e, (H + lambda diag H) d = -g via Schur on the 3x3 landmark blocks.
//
// Data in HBM (float64):
//   obs   [W][N][2]   slot-major (slot 0 = newest frame), NaN = not observed
//   x[2]  {poses [W][6], points [N][3]}  current / trial, selected by state.cur
//
// Two kernel families build and update (chosen by the window, ba_geometry):
//   * windows of <= 10 slots (BASELINE's 10, the reference's own 4): the WAVE-PRIVATE k_ba_build_w / k_ba_update_w of vo_ba_wave.h -- a wave walks
//     landmark chunks with all Gram tiles in its accumulator registers, 5 / 8 / 4 lanes per landmark, running-problem compaction of the tail groups;
//   * windows of 11 .. 20 slots (config 5), the sharded solve's default, one or two problems at windows of 9-10 slots, and vo_tuning.ba_kernels = 1: the kernels below.
// k_ba_reduce / k_ba_solve / k_ba_finalize serve both.
//
// Work mapping of the kernels in this file: ONE LANE PER OBSERVATION.  A landmark owns a group of LPP = 8 (W <= 8), 16 (W <= 16) or 32 lanes, lane s of
// the group handles the observation in window slot s; landmark sums are DPP row-rotate all-reduces inside the
// group, camera sums are cross-group shuffles + one LDS pass, so every observation is linearised exactly once
// per kernel and nothing is re-read from HBM.
//
// One LM iteration = 3 launches on the ctx stream (state double-buffered by iteration parity):
//   k_ba_build  : [decide previous step]; per workgroup (1024 lanes = 64 or 32 landmarks): linearise, H_ll/g_l,
//                 damped 3x3 inverse factor L_j, camera partial sums (H_pp, g_p, cost); the workgroup's slice of
//                 Y^ = [H_pl L | L^T g_l] is staged in LDS ((3 PPB) x RP panel, 120 KB) and its Gram matrix
//                 Y^ Y^T -- i.e. E = sum_j H_pl M_j H_pl^T and r = sum_j H_pl M_j g_l at once -- is accumulated
//                 with v_mfma_f64_16x16x4_f64 (upper 16x16 tiles) -> one partial tile set per workgroup
//   k_ba_solve  : one workgroup: fixed-order reduction of the partials, damped reduced camera system
//                 S = H_pp + lambda D - E, 6x6-blocked Cholesky of [S rhs] in LDS (diagonal blocks factorised
//                 redundantly in registers -> 2 barriers per block column), wave-level back substitution
//   k_ba_update : lane per observation again: d_point = -M_j (g_l + sum_i B_ij^T d_pose_i), trial x, trial cost,
//                 step statistics
// k_ba_finalize applies the last decision and publishes x / stats.  All reductions have a fixed order, so a
// solve is bitwise reproducible from run to run.
#include "vo_internal.h"

#include <math.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

// The library is built with -ffp-contract=off because the float32 front end (KLT update, Shi-Tomasi, DLT output) must
// round exactly like the CPU oracle.  The bundle adjustment is float64 against tolerances (and bitwise reproducible
// against itself): here a*b+c may fuse -- one v_fma_f64 instead of v_mul_f64 + v_add_f64 in kernels that are bound by
// f64 VALU issue.
#pragma clang fp contract(fast)

#define BA_SOLVE_THREADS 1024
#define BA_AUX 18            // per landmark: Hll(6) gl(3) Cinv(6) z(3)
#define BA_POSE_VALS 28      // Hpp upper (21) + gp (6) + cost (1)
#define BA_MAX_SLOTS 20
#define BA_EVAL_VALS 4
#define BA_PITCH_PAD 8       // panel row pitch = RP + 8 doubles.  A pitch of RP + 16 keeps the 16-lane row groups of the MFMA operand reads on
                             // disjoint LDS banks, but its 30.7 KB panel lets only 4 workgroups share a CU; measured (vo_tuning.ba_pitch_pad): pads 0..12
                             // 54 us per launch, pad 16 57 us -- the bank conflicts cost nothing here, the fifth workgroup is worth 5 %
#define BA_CAM 21            // per-slot camera data staged in LDS: R(9) t(3) Jr(9)

typedef double d4 __attribute__((ext_vector_type(4)));

struct ba_info {            // written by k_ba_solve for the iteration
  double cost_cur, pred_pose, step2_pose, x2_pose, ginf;
  int chol_fail, pad;
};

struct ba_params_dev {
  double ftol, xtol, gtol, lambda0, lambda_min, delta;
  int max_iters;
};

struct ba_ptrs {
  const double* K; const double* obs; const double* x0;
  double* xa; double* xb;             // x[0] / x[1] (two named members: a dynamically indexed array would push the struct into scratch)
  double* aux; double* posepart; double* gmax; double* tiles; double* dp; double* evalpart;
  double* tilesum; double* posesum;   // k_ba_reduce outputs: n_tiles*256, W*28 + 1 (last = max |g_l|); ONE allocation per problem:
                                      // [tilesum | posesum | max|g_l| slot per rank] = the all-reduced packet of a sharded solve
  double* xstat;                      // sharded solve: [4] step statistics summed over all shards (every entry holds the total)
  double* cams;                       // [2][W][21]: R, t, Jr of the poses in x[0] / x[1] (written by k_ba_solve)
  ba_state* state; ba_info* info;
  unsigned long long* dbg;
  // per-problem strides (elements) of the batched buffers
  size_t s_obs, s_x, s_aux, s_posepart, s_gmax, s_tiles, s_dp, s_evalpart, s_tilesum, s_posesum, s_cams;
  int W, N, LPP, PPB, nblk, RP, RT, n_tiles, pitch;
  int nset;                           // partial sets (= workgroups of k_ba_build): nblk, or fewer when a workgroup walks several landmark chunks
  int cam_off;                        // k_ba_build: offset (doubles) of the staged cameras inside the dynamic LDS
  int sharded, rank, n_ranks, batch;  // sharded: the batch entries (x the ranks) are landmark shards of one problem
  int fold;                           // the solve sums the partial sets itself, no k_ba_reduce launch (small windows, unsharded: ba_make_ptrs)
  int n_eval;                         // entries of evalpart (= workgroups of k_ba_update)
  int* gdyn;                          // wave-private kernels: [2][batch] partial sets / statistics entries of a problem in iteration it & 1 (null: nset, n_eval)
  const int32_t* n_live; int s_nlive; // optional (closed loop): landmark slots [0, *n_live) of the problem are in use, the workgroups of the rest only zero
                                      // their partial sums (the tables of vo_pipeline.hip are sized for max_pts landmarks, a scene fills a part of them)
};

__device__ __forceinline__ double* ba_x(const ba_ptrs& P, int k) { return k ? P.xb : P.xa; }

// pointers of problem b of the batch
__device__ __forceinline__ ba_ptrs ba_select(ba_ptrs P, int b) {
  const size_t sb = (size_t)b;
  P.K += sb * 9; P.obs += sb * P.s_obs; P.x0 += sb * P.s_x; P.xa += sb * P.s_x; P.xb += sb * P.s_x;
  P.aux += sb * P.s_aux; P.posepart += sb * P.s_posepart; P.gmax += sb * P.s_gmax; P.tiles += sb * P.s_tiles;
  P.dp += sb * P.s_dp; P.evalpart += sb * P.s_evalpart; P.tilesum += sb * P.s_tilesum; P.posesum += sb * P.s_posesum; P.cams += sb * P.s_cams;
  P.xstat += sb * BA_EVAL_VALS;
  P.state += 2 * sb; P.info += sb;
  if (P.n_live) P.n_live += sb * P.s_nlive;
  if (b != 0) P.dbg = nullptr;
  return P;
}

struct vo_ba_ws {
  int W = 0, N = 0, LPP = 0, PPB = 0, nblk = 0, RP = 0, RT = 0, n_tiles = 0, pitch = 0, tpb = 0;
  int cap_W = 0, cap_N = 0;
  int cap_nblk = 0;             // partial sets the nblk-sized buffers (posepart, gmax, tiles, evalpart) were allocated for
  size_t build_lds = 0, solve_lds = 0;
  int cam_off = 0;
  double* d_K = nullptr;        // 9
  double* d_obs = nullptr;      // W*N*2
  double* d_x0 = nullptr;       // W*6 + N*3
  double* d_x[2] = {nullptr, nullptr};
  double* d_aux = nullptr;      // N * BA_AUX
  double* d_posepart = nullptr; // nblk * W * 28
  double* d_gmax = nullptr;     // nblk
  double* d_tiles = nullptr;    // nblk * n_tiles * 256
  double* d_dp = nullptr;       // 6W
  double* d_evalpart = nullptr; // nblk * 4
  double* d_tilesum = nullptr;  // n_tiles * 256                  } one allocation, red_stride doubles per problem:
  double* d_posesum = nullptr;  // W * 28 + 1 (+ 16 rank slots)    } d_posesum = d_tilesum + n_tiles * 256
  size_t red_stride = 0;
  double* d_xstat = nullptr;    // [batch][4] sharded solve: summed step statistics
  double* d_gather = nullptr;   // vo_ba_gather_points: send [batch][N][3] | recv [n_ranks][batch][N][3]
  double* h_gather = nullptr;   // pinned mirror of recv
  size_t gather_cap = 0;
  double* d_cams = nullptr;     // 2 * W * 21
  double* d_S = nullptr;        // probe: (6W)^2 + 6W
  double* d_Hpp = nullptr;      // W*28 reduced pose values (probe)
  double* d_res = nullptr;      // probe residual W*N
  double* d_dl = nullptr;       // probe d_points 3N
  double* d_xout = nullptr;     // published solution 6W + 3N (points into d_pub)
  uint8_t* d_pub = nullptr;     // [ba_state (64 B) | x]: what a fetch copies back in one go
  uint8_t* h_pub = nullptr;     // pinned mirror
  size_t pub_bytes = 0;
  ba_state* d_state = nullptr;  // [2]
  ba_info* d_info = nullptr;
  ba_state* h_state = nullptr;  // pinned
  bool uploaded = false;
  // bank of resident problems (vo_ba_upload_bank): d_x0 / d_obs point at the selected one; the workspace's own buffers are kept for the free
  double* d_x0_own = nullptr; double* d_obs_own = nullptr;
  double* d_bank_x0 = nullptr; double* d_bank_obs = nullptr;
  int bank_n = 0, bank_sel = 0;
  const int32_t* d_nlive = nullptr; int nlive_stride = 0;     // vo_ba_set_live
  // wave-private kernels (vo_ba_wave.h): windows of <= 10 slots
  int v2 = 0, v2_rt = 0, v2_spl = 0, v2_lpp = 8;
  size_t v2_lds = 0;
  int v2_g0 = 1, v2_gcap = 1;       // workgroups per problem of a launch; the most a running problem is given once others have finished
  int* d_gdyn = nullptr;            // [2][batch]
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// R (Rodrigues), t and the right Jacobian of SO(3) (R(r + dr) ~ R(r) Exp(Jr dr)) of one pose: 21 doubles, ONE sincos -- the trial
// cameras are on k_ba_solve's critical path, computed by W lanes
__device__ __forceinline__ void d_camera(const double* p, double* cam) {
  const double x = p[0], y = p[1], z = p[2];
  const double th2 = x * x + y * y + z * z;
  const double th = sqrt(th2);
  double sn, cs;
  sincos(th, &sn, &cs);
  double* R = cam;
  if (th < 2.220446049250313e-16) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  } else {
    const double kx = x / th, ky = y / th, kz = z / th, c1 = 1.0 - cs;
    R[0] = cs + c1 * kx * kx;      R[1] = c1 * kx * ky - sn * kz; R[2] = c1 * kx * kz + sn * ky;
    R[3] = c1 * ky * kx + sn * kz; R[4] = cs + c1 * ky * ky;      R[5] = c1 * ky * kz - sn * kx;
    R[6] = c1 * kz * kx - sn * ky; R[7] = c1 * kz * ky + sn * kx; R[8] = cs + c1 * kz * kz;
  }
  cam[9] = p[3]; cam[10] = p[4]; cam[11] = p[5];
  double a, b;
  if (th2 < 1e-8) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
  else { a = (1.0 - cs) / th2; b = (th - sn) / (th2 * th); }
  double* J = cam + 12;
  J[0] = 1.0 + b * (x * x - th2); J[1] = a * z + b * x * y;       J[2] = -a * y + b * x * z;
  J[3] = -a * z + b * y * x;      J[4] = 1.0 + b * (y * y - th2); J[5] = a * x + b * y * z;
  J[6] = a * y + b * z * x;       J[7] = -a * x + b * z * y;      J[8] = 1.0 + b * (z * z - th2);
}

__device__ __forceinline__ void stage_cameras(const double* poses, int W, double* cam, int tid, int nthreads) {
  for (int i = tid; i < W; i += nthreads) {
    const double* p = poses + 6 * i;
    double* c = cam + BA_CAM * i;
    d_camera(p, c);
  }
}

struct ba_obs_lin {
  double e0, e1, w, rho;
  double Jl[2][3];
  double Jp[2][6];
};

// residual + Jacobian blocks of one observation.  returns false if unobserved.
// Aout (optional, with WANT_JP = false): the 2 x 3 projection Jacobian d(u, v) / d(camera-frame point) -- with it the caller forms
// Jp d for a given step d without the 2 x 6 block (k_ba_update)
template <bool WANT_JP>
__device__ __forceinline__ bool ba_linearize_obs(const double* __restrict__ K, const double* __restrict__ cam,
                                                 const double X[3], double uo, double vo, double delta, ba_obs_lin& o,
                                                 double (*Aout)[3] = nullptr) {
  if (uo != uo) return false;
  const double* R = cam; const double* t = cam + 9; const double* Jr = cam + 12;
  const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  const double p0 = K[0] * xc + K[1] * yc + K[2] * zc;
  const double p1 = K[3] * xc + K[4] * yc + K[5] * zc;
  const double p2 = K[6] * xc + K[7] * yc + K[8] * zc;
  const double ip2 = 1.0 / p2;
  const double u = p0 * ip2, v = p1 * ip2;
  o.e0 = u - uo; o.e1 = v - vo;
  const double s = o.e0 * o.e0 + o.e1 * o.e1;
  const double d2 = delta * delta;
  if (s <= d2) { o.w = 1.0; o.rho = s; }
  else { const double rs = sqrt(s); o.w = delta / rs; o.rho = 2.0 * delta * rs - d2; }
  double A[2][3];
#pragma unroll
  for (int c = 0; c < 3; c++) { A[0][c] = (K[c] - u * K[6 + c]) * ip2; A[1][c] = (K[3 + c] - v * K[6 + c]) * ip2; }
  if (Aout) {
#pragma unroll
    for (int c = 0; c < 3; c++) { Aout[0][c] = A[0][c]; Aout[1][c] = A[1][c]; }
  }
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) o.Jl[k][c] = A[k][0] * R[c] + A[k][1] * R[3 + c] + A[k][2] * R[6 + c];
  if (WANT_JP) {
    // M3 = [X]x Jr  (column c = X x Jr[:,c]);  Jp_rot = -Jl M3 ;  Jp_trans = A
    double M3[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double a0 = Jr[c], a1 = Jr[3 + c], a2 = Jr[6 + c];
      M3[0][c] = X[1] * a2 - X[2] * a1;
      M3[1][c] = X[2] * a0 - X[0] * a2;
      M3[2][c] = X[0] * a1 - X[1] * a0;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
      for (int c = 0; c < 3; c++)
        o.Jp[k][c] = -(o.Jl[k][0] * M3[0][c] + o.Jl[k][1] * M3[1][c] + o.Jl[k][2] * M3[2][c]);
#pragma unroll
      for (int c = 0; c < 3; c++) o.Jp[k][3 + c] = A[k][c];
    }
  }
  return true;
}

// ---- cross-lane helpers on doubles ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// v[l] + v[l ^ 16] in every lane: v_permlane16_swap (VALU, no LDS traffic).  swap(a, b) with a = b = v leaves
// a' = {row0, row0, row2, row2}, b' = {row1, row1, row3, row3}, so a' + b' is the pair sum everywhere.
__device__ __forceinline__ double xor16_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
// v[l] + v[l ^ 32] in every lane: v_permlane32_swap
__device__ __forceinline__ double xor32_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}

// Reduce-scatter forms of the two swaps: x and y are summed over lane pairs at once, half of the lanes keep x's sum, the other
// half y's -- 3 instructions for two values instead of 3 per value (the all-reduce forms above duplicate every sum).
//   rs16(x, y): rows 0 and 2 of the wave receive x[row] + x[row + 1], rows 1 and 3 receive y[row - 1] + y[row]
//   rs32(x, y): lanes 0..31 receive x[l] + x[l + 32], lanes 32..63 receive y[l - 32] + y[l]
__device__ __forceinline__ double rs16_sum(double x, double y) {
  const auto l2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
__device__ __forceinline__ double rs32_sum(double x, double y) {
  const auto l2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}

// sum over the LPP (16 or 32) lanes of a landmark group; every lane of the group gets the total
__device__ __forceinline__ double group_allreduce(double v, int lpp) {
  v += dpp_f64<0x128>(v);   // row_ror:8
  v += dpp_f64<0x124>(v);   // row_ror:4
  v += dpp_f64<0x122>(v);   // row_ror:2
  v += dpp_f64<0x121>(v);   // row_ror:1
  if (lpp == 32) v = xor16_sum(v);
  return v;
}

// the same over a group of 8 lanes (windows of <= 8 slots: twice the landmarks per wave): x[l] + x[7 - l] by row_half_mirror, then the two
// quad exchanges -- lanes 0..3 and 4..7 of the group hold the same four pair sums (mirrored), so both quads finish with the total
__device__ __forceinline__ double group8_allreduce(double v) {
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  return v;
}
template <int LPPC>
__device__ __forceinline__ double group_allreduce_t(double v, int lpp) { return LPPC == 8 ? group8_allreduce(v) : group_allreduce(v, lpp); }

//   rs8(x, y): lanes with bit 3 clear receive x[l] + x[l ^ 8], lanes with bit 3 set receive y[l ^ 8] + y[l]  (row_ror:8 = lane ^ 8 in a row)
__device__ __forceinline__ double rs8_sum(double x, double y, int lane) {
  const bool hi = (lane & 8) != 0;
  const double keep = hi ? y : x, send = hi ? x : y;
  return keep + dpp_f64<0x128>(send);
}

// sum over the 64 lanes of a wave, result in every lane
__device__ __forceinline__ double wave_allreduce(double v) {
  v += dpp_f64<0x128>(v);
  v += dpp_f64<0x124>(v);
  v += dpp_f64<0x122>(v);
  v += dpp_f64<0x121>(v);
  return xor32_sum(xor16_sum(v));
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(x) to double precision: v_rsq_f64 seed + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  return y;
}

// ------------------------------------------------------------------------------------------------
// LM decision (inputs are identical in every workgroup, so every workgroup derives the same state)
// ------------------------------------------------------------------------------------------------
// fixed-order block reduction of the per-workgroup step statistics written by k_ba_update (4 values per workgroup): wave k sums
// statistic k -- lane l the workgroups l, l + 64, ... in order, then the fixed cross-lane tree of wave_allreduce -- so every caller
// derives bit-identical sums without staging anything in LDS (one barrier instead of three).  Call from all threads (TPB >= 256).
template <int TPB>
__device__ __forceinline__ void ba_reduce_evalpart(const double* __restrict__ evalpart, int n_eblk, double* s_sum /* 4 */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < BA_EVAL_VALS) {
    double s = 0;
    for (int b = lane; b < n_eblk; b += 64) s += evalpart[b * BA_EVAL_VALS + wave];
    s = wave_allreduce(s);
    if (lane == 0) s_sum[wave] = s;
  }
  __syncthreads();
}

__device__ inline void ba_decide(const ba_state& in, const ba_info& info, const double* __restrict__ sums,
                                 const ba_params_dev& prm, ba_state& out) {
  out = in;
  if (in.done) return;
  const double F = info.cost_cur;
  if (in.iter == 0) out.cost0 = F;
  out.cost = F;
  if (info.ginf < prm.gtol) { out.done = 1; out.status = 1; return; }
  const double Ft = sums[0], predp = sums[1], step2 = sums[2], x2 = sums[3];
  const double pred = 0.5 * (predp + info.pred_pose);
  const double step = sqrt(step2 + info.step2_pose), xn = sqrt(x2 + info.x2_pose);
  const bool ok = !info.chol_fail;
  const double rho = (ok && pred > 0) ? (F - Ft) / pred : -1.0;
  out.iter = in.iter + 1;
  if (ok && Ft < F && rho > 0) {
    out.cur = in.cur ^ 1;
    out.cost = Ft;
    out.accepted = in.accepted + 1;
    const double q = 2.0 * rho - 1.0;
    double f = 1.0 - q * q * q;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    double lam = in.lambda * f;
    if (lam < prm.lambda_min) lam = prm.lambda_min;
    out.lambda = lam; out.nu = 2.0;
    if ((F - Ft) < prm.ftol * Ft) { out.done = 1; out.status = 2; }
    else if (step < prm.xtol * (prm.xtol + xn)) { out.done = 1; out.status = 3; }
  } else {
    if (ok && step < prm.xtol * (prm.xtol + xn)) { out.done = 1; out.status = 3; }
    else {
      out.lambda = in.lambda * in.nu; out.nu = in.nu * 2.0;
      if (out.lambda > 1e12) { out.done = 1; out.status = 4; }
    }
  }
  if (!out.done && out.iter >= prm.max_iters) { out.done = 1; out.status = 0; }
}

__device__ inline ba_state ba_init_state(const ba_params_dev& prm) {
  ba_state s;
  s.lambda = prm.lambda0; s.nu = 2.0; s.cost = 0; s.cost0 = 0;
  s.cur = 0; s.iter = 0; s.accepted = 0; s.status = 0; s.done = 0; s.n_obs = 0;
  return s;
}

#include "vo_ba_wave.h"

// ------------------------------------------------------------------------------------------------
// k_ba_build
// ------------------------------------------------------------------------------------------------
// LPPC = 8: a landmark owns 8 lanes (windows of <= 8 slots -- the reference's own window is 4: with 16 lanes three quarters of the lanes carried
// zeros through every phase); LPPC = 0: 16 or 32 lanes, taken from the problem (P.LPP)
template <int TPB, int LPPC = 0>
__global__ void __launch_bounds__(TPB) k_ba_build(ba_ptrs Pall, ba_params_dev prm, int it, double probe_lambda) {
  const ba_ptrs P = ba_select(Pall, blockIdx.y);
  extern __shared__ double dyn[];   // phase A: camera-sum scratch [wave][LPP][28]; phase B: Y^ panel [3 PPB][pitch]
  // the staged cameras live behind the panel, sized by the actual window (W x 21 doubles): with a static 20-slot array
  // the W = 10 kernel needed 34.3 KB of LDS -> 4 workgroups per CU; now 32.6 KB -> 5
  double* s_cam = dyn + P.cam_off;
  __shared__ double s_K[9];
  __shared__ double s_gmax[(TPB / 64)];
  __shared__ ba_state s_st;
  __shared__ double s_esum[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ---- a problem that finished in an earlier iteration only carries its state forward ----
  if (it > 0 && P.state[(it - 1) & 1].done) {
    if (blockIdx.x == 0 && tid == 0) P.state[it & 1] = P.state[(it - 1) & 1];
    return;
  }
  // ---- state for this iteration (every workgroup derives it from the same inputs).  Handing the decision to the workgroup of
  //      k_ba_update that arrives last (an arrival counter per problem) was measured: k_ba_build 60 -> 54 us, but 125 agent-scope
  //      atomics on one address serialise (k_ba_update 15 -> 43 us), and an agent-scope release fence writes the L2 back (245 us) ----
  if (it > 0) ba_reduce_evalpart<TPB>(P.sharded ? P.xstat : P.evalpart, P.sharded ? 1 : P.n_eval, s_esum);
  if (tid == 0) {
    ba_state st;
    if (it == 0) st = ba_init_state(prm);
    else ba_decide(P.state[(it - 1) & 1], *P.info, s_esum, prm, st);
    if (probe_lambda >= 0) st.lambda = probe_lambda;
    s_st = st;
    if (blockIdx.x == 0) P.state[it & 1] = st;
  }
  __syncthreads();
  const ba_state st = s_st;
  if (st.done) return;
  unsigned long long* dbgb = (blockIdx.x == 0 && P.dbg) ? P.dbg + 16 : nullptr;
  VO_STAMP(dbgb, 0);
  const int W = P.W, N = P.N, LPP = LPPC ? LPPC : P.LPP;
  // iteration 0 reads the uploaded x0 and seeds x[0] with it (each workgroup its own landmarks)
  const double* poses = (it == 0) ? P.x0 : ba_x(P, st.cur);
  const double* pts = poses + 6 * W;
  // this lane's landmark and observation: issued before the camera staging so that the two HBM round trips overlap
  // cameras and K into LDS, once per workgroup
  if (it == 0) {
    stage_cameras(poses, W, s_cam, tid, TPB);           // nobody has prepared the cameras of x0 yet
  } else {
    const double* cg = P.cams + (size_t)st.cur * W * BA_CAM;   // prepared by k_ba_solve of the previous iteration
    for (int i = tid; i < W * BA_CAM; i += TPB) s_cam[i] = cg[i];
  }
  if (tid < 9) s_K[tid] = P.K[tid];
  if (it == 0 && blockIdx.x == 0 && tid < 6 * W) P.xa[tid] = poses[tid];
  __syncthreads();
  if (it == 0 && blockIdx.x == 0) for (int i = tid; i < W * BA_CAM; i += TPB) P.cams[i] = s_cam[i];   // cams[0] <-> x[0]
  const int pl = tid / LPP, slot = tid - pl * LPP;       // landmark (local), window slot
  // A workgroup walks the landmark chunks blockIdx.x, blockIdx.x + gridDim.x, ... (PPB landmarks each) and owns ONE partial set:
  // the later chunks add their Gram tiles and camera sums to what the earlier ones stored (the workgroup's own 20 KB, still in L2).
  // Half as many partial sets with two chunks per workgroup: the tile stores are HBM-write bound (61 MB per launch, 15 of the
  // kernel's 58 us) and k_ba_reduce reads them all back.
  int seed_x = (it == 0) ? 1 : 0;                  // (opaque, so that the chunk loop is not versioned on it)
  asm volatile("" : "+v"(seed_x));
  const int n_live = P.n_live ? *P.n_live : N;
  // this lane's landmark and observation of a chunk; the NEXT chunk's are requested before the Gram phase of this one (they were the first
  // thing a chunk waited for: a full trip to HBM / L2 at the head of every chunk of the walk)
  double Xn[3] = {0, 0, 0}, uon = __builtin_nan(""), von = 0;
  auto fetch_obs = [&](const int ch) {
    const int jn = ch * P.PPB + pl;
    Xn[0] = Xn[1] = Xn[2] = 0; uon = __builtin_nan(""); von = 0;
    if (ch < P.nblk && ch * P.PPB < n_live && slot < W && jn < N) {
      Xn[0] = pts[3 * jn]; Xn[1] = pts[3 * jn + 1]; Xn[2] = pts[3 * jn + 2];
      const double* ob = P.obs + ((size_t)slot * N + jn) * 2;
      uon = ob[0]; von = ob[1];
    }
  };
  fetch_obs(blockIdx.x);
#pragma unroll 1
  for (int chunk = blockIdx.x; chunk < P.nblk; chunk += gridDim.x) {
  const bool later = chunk != (int)blockIdx.x;
  if (chunk * P.PPB >= n_live) {                   // (uniform) no landmark of this chunk -- nor of the later ones -- is in use
    if (!later) {
      for (int t = tid; t < W * BA_POSE_VALS; t += TPB) P.posepart[(size_t)blockIdx.x * W * BA_POSE_VALS + t] = 0.0;
      for (int t = tid; t < P.n_tiles * 256; t += TPB) P.tiles[(size_t)blockIdx.x * P.n_tiles * 256 + t] = 0.0;
      if (tid == 0) P.gmax[blockIdx.x] = 0.0;
    }
    break;
  }
  const int j = chunk * P.PPB + pl;
  const double X[3] = {Xn[0], Xn[1], Xn[2]};
  const double uo = uon, vo = von;

  ba_obs_lin o;
  bool have = false;
  if (seed_x && slot == 0 && j < N) {
    double* dst = P.xa + 6 * W + 3 * j;
    dst[0] = X[0]; dst[1] = X[1]; dst[2] = X[2];
  }
  if (slot < W && j < N) have = ba_linearize_obs<true>(s_K, s_cam + BA_CAM * slot, X, uo, vo, prm.delta, o);
  if (!have) {
    o.e0 = o.e1 = o.w = o.rho = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
      for (int c = 0; c < 3; c++) o.Jl[k][c] = 0;
#pragma unroll
      for (int c = 0; c < 6; c++) o.Jp[k][c] = 0;
    }
  }
  VO_STAMP(dbgb, 1);   // cameras staged + observation linearised
  // the Huber weight goes into one factor of every product once (12 multiplications instead of one per term: 39 fewer)
  double wJp[2][6];
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int a = 0; a < 6; a++) wJp[k][a] = o.w * o.Jp[k][a];
  // ---- landmark sums over the group ----
  const double h00 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][0] * o.Jl[0][0] + o.Jl[1][0] * o.Jl[1][0]), LPP);
  const double h10 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][1] * o.Jl[0][0] + o.Jl[1][1] * o.Jl[1][0]), LPP);
  const double h11 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][1] * o.Jl[0][1] + o.Jl[1][1] * o.Jl[1][1]), LPP);
  const double h20 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][2] * o.Jl[0][0] + o.Jl[1][2] * o.Jl[1][0]), LPP);
  const double h21 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][2] * o.Jl[0][1] + o.Jl[1][2] * o.Jl[1][1]), LPP);
  const double h22 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][2] * o.Jl[0][2] + o.Jl[1][2] * o.Jl[1][2]), LPP);
  const double g0 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][0] * o.e0 + o.Jl[1][0] * o.e1), LPP);
  const double g1 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][1] * o.e0 + o.Jl[1][1] * o.e1), LPP);
  const double g2 = group_allreduce_t<LPPC>(o.w * (o.Jl[0][2] * o.e0 + o.Jl[1][2] * o.e1), LPP);
  // ---- damped 3x3 block: Cholesky C C^T, Cinv = C^-1 (lower), y = Cinv g, z = Cinv^T y = M g ----
  const double lam = st.lambda;
  const double a00 = h00 + lam * fmax(h00, 1e-12), a11 = h11 + lam * fmax(h11, 1e-12), a22 = h22 + lam * fmax(h22, 1e-12);
  const double i00 = rsqrt_nr(a00);
  const double c10 = h10 * i00, c20 = h20 * i00;
  const double i11 = rsqrt_nr(a11 - c10 * c10);
  const double c21 = (h21 - c20 * c10) * i11;
  const double i22 = rsqrt_nr(a22 - c20 * c20 - c21 * c21);
  const double i10 = -c10 * i00 * i11;
  const double i21 = -c21 * i11 * i22;
  const double i20 = -(c20 * i00 + c21 * i10) * i22;
  const double y0 = i00 * g0, y1 = i10 * g0 + i11 * g1, y2 = i20 * g0 + i21 * g1 + i22 * g2;
  if (slot == 0 && j < N) {
    double* ax = P.aux + (size_t)j * BA_AUX;
    ax[0] = h00; ax[1] = h10; ax[2] = h11; ax[3] = h20; ax[4] = h21; ax[5] = h22;
    ax[6] = g0; ax[7] = g1; ax[8] = g2;
    ax[9] = i00; ax[10] = i10; ax[11] = i11; ax[12] = i20; ax[13] = i21; ax[14] = i22;
    ax[15] = i00 * y0 + i10 * y1 + i20 * y2; ax[16] = i11 * y1 + i21 * y2; ax[17] = i22 * y2;
  }
  // ---- max |g_l| of the workgroup ----
  {
    double gm = (j < N) ? fmax(fabs(g0), fmax(fabs(g1), fabs(g2))) : 0.0;
    for (int ofs = 32; ofs > 0; ofs >>= 1) gm = fmax(gm, __shfl_xor(gm, ofs));
    if (lane == 0) s_gmax[wave] = later ? fmax(s_gmax[wave], gm) : gm;
  }
  VO_STAMP(dbgb, 2);   // group sums + 3x3 factor
  // ---- camera sums: across the landmarks of the wave by shuffles, across waves through LDS ----
  // The 28 values of a slot (21 of the upper H_pp, 6 of g_p, the cost) go four at a time through a REDUCE-SCATTER over the wave's
  // landmarks: after the two swap stages row r of the wave holds the total of value 4 g + r, so every lane stores one value per
  // group (7 stores) -- 63 instead of 168 cross-lane instructions per wave, the same additions in the same order as the
  // all-reduce it replaces (bit-identical sums).  Four values are live at a time.
  {
    constexpr int QA[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
    constexpr int QC[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
    auto term = [&](int q) -> double {
      if (q < 21) return wJp[0][QA[q]] * o.Jp[0][QC[q]] + wJp[1][QA[q]] * o.Jp[1][QC[q]];
      if (q < 27) return wJp[0][q - 21] * o.e0 + wJp[1][q - 21] * o.e1;
      return 0.5 * o.rho;
    };
    double* dst = dyn + (size_t)(wave * LPP + (lane & (LPP - 1))) * BA_POSE_VALS;
    if (LPPC == 8) {
      // eight landmarks per wave: four values at a time through two reduce-scatter stages (lane ^ 8, lane ^ 16) -- the lane whose bits
      // 3 and 4 spell i then holds value 4 g + i summed over four of the wave's landmarks -- and one exchange with lane ^ 32 for the other
      // four; the lower half-wave stores.  (Eight values at a time through three scatter stages is the same instruction count at 130
      // instead of 125 registers: one workgroup per CU less.)
#pragma unroll
      for (int g = 0; g < BA_POSE_VALS / 4; g++) {
        const double a01 = rs8_sum(term(4 * g), term(4 * g + 1), lane), a23 = rs8_sum(term(4 * g + 2), term(4 * g + 3), lane);
        const double u = xor32_sum(rs16_sum(a01, a23));
        if (lane < 32) dst[4 * g + ((lane >> 3) & 1) + 2 * (lane >> 4)] = u;
      }
    } else
#pragma unroll
    for (int g = 0; g < BA_POSE_VALS / 4; g++) {
      const double v0 = term(4 * g), v1 = term(4 * g + 1), v2 = term(4 * g + 2), v3 = term(4 * g + 3);
      if (LPP == 16) {
        const double u = rs32_sum(rs16_sum(v0, v1), rs16_sum(v2, v3));      // row r: total of v_r over the wave's four landmarks
        dst[4 * g + (lane >> 4)] = u;
      } else {
        const double u01 = rs32_sum(v0, v1), u23 = rs32_sum(v2, v3);       // lanes 0..31: v0 / v2, lanes 32..63: v1 / v3
        dst[4 * g + (lane >> 5)] = u01;
        dst[4 * g + 2 + (lane >> 5)] = u23;
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < W * BA_POSE_VALS; t += TPB) {
    const int sl = t / BA_POSE_VALS, k = t - sl * BA_POSE_VALS;
    double s = 0;
#pragma unroll
    for (int wv = 0; wv < (TPB / 64); wv++) s += dyn[(size_t)(wv * LPP + sl) * BA_POSE_VALS + k];
    double* pp = P.posepart + ((size_t)blockIdx.x * W + sl) * BA_POSE_VALS + k;
    *pp = later ? *pp + s : s;
  }
  if (tid == 0) {
    double gm = 0;
    for (int wv = 0; wv < (TPB / 64); wv++) gm = fmax(gm, s_gmax[wv]);
    P.gmax[blockIdx.x] = gm;
  }
  __syncthreads();
  VO_STAMP(dbgb, 3);   // camera sums reduced and written
  // ---- Y^ panel of this workgroup in LDS: row 3 pl + c, columns 6 slot .. 6 slot + 5, column 6W = y ----
  const int pitch = P.pitch;
  double* const rw0 = dyn + (size_t)(3 * pl) * pitch;         // rows 3 pl, 3 pl + 1, 3 pl + 2 of the panel
  double* const rw1 = rw0 + pitch;
  double* const rw2 = rw1 + pitch;
  if (slot < W) {
#pragma unroll
    for (int a = 0; a < 6; a++) {
      const double b0 = wJp[0][a] * o.Jl[0][0] + wJp[1][a] * o.Jl[1][0];
      const double b1 = wJp[0][a] * o.Jl[0][1] + wJp[1][a] * o.Jl[1][1];
      const double b2 = wJp[0][a] * o.Jl[0][2] + wJp[1][a] * o.Jl[1][2];
      rw0[6 * slot + a] = b0 * i00;                                   // Y[a][0] = B[a][0] Cinv[0][0]
      rw1[6 * slot + a] = b0 * i10 + b1 * i11;                        // Y[a][1]
      rw2[6 * slot + a] = b0 * i20 + b1 * i21 + b2 * i22;             // Y[a][2]
    }
  }
  if (slot == 0) {
    const bool in = j < N;
    rw0[6 * W] = in ? y0 : 0.0; rw1[6 * W] = in ? y1 : 0.0; rw2[6 * W] = in ? y2 : 0.0;
    for (int cidx = 6 * W + 1; cidx < P.RP; cidx++) { rw0[cidx] = 0; rw1[cidx] = 0; rw2[cidx] = 0; }
  }
  __syncthreads();
  fetch_obs(chunk + (int)gridDim.x);
  // ---- Gram matrix of the panel: upper 16x16 tiles, one wave per tile, v_mfma_f64_16x16x4_f64 ----
  //   A[i][k] = panel[k0 + k][16 ta + i]  (lane: i = l & 15, k = l >> 4),  B[k][j] = panel[k0 + k][16 tb + j]
  //   D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  VO_STAMP(dbgb, 4);   // panel staged
  const int krows = 3 * P.PPB;
  // a wave walks its tiles three at a time: three independent accumulator chains hide the LDS and MFMA latency
  for (int tile0 = wave; tile0 < P.n_tiles; tile0 += 3 * (TPB / 64)) {
    int tas[3], tbs[3];
    bool on[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int tile = tile0 + u * (TPB / 64);
      on[u] = tile < P.n_tiles;
      int ta = 0, rem = on[u] ? tile : 0;
      while (rem >= P.RT - ta) { rem -= P.RT - ta; ta++; }
      tas[u] = ta; tbs[u] = ta + rem;
    }
    const double* base = dyn + (size_t)(lane >> 4) * pitch + (lane & 15);
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0;
    // krows = 3 PPB is a multiple of 8 (PPB = 8, 16 or 32): two k-steps per trip, their LDS reads issued before the MFMAs.  The
    // chains of one loop run unconditionally (a conditional MFMA makes the compiler shuttle accumulators between register files);
    // a wave that owns only one or two tiles in this trip (wave-uniform) takes the loop with that many chains instead of
    // recomputing tile 0 in the spare ones (10 tiles on 4 waves: waves 2 and 3 issued 36 MFMAs for 24)
    auto gram = [&](auto nch_tag) {
      constexpr int NCH = decltype(nch_tag)::value;
      for (int k0 = 0; k0 < krows; k0 += 8) {
        double av[2][3], bv[2][3];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const double* rowp = base + (size_t)(k0 + 4 * q) * pitch;
#pragma unroll
          for (int u = 0; u < NCH; u++) { av[q][u] = rowp[16 * tas[u]]; bv[q][u] = rowp[16 * tbs[u]]; }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][0], bv[q][0], acc0, 0, 0, 0);
          if (NCH > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][1], bv[q][1], acc1, 0, 0, 0);
          if (NCH > 2) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][2], bv[q][2], acc2, 0, 0, 0);
        }
      }
    };
    if (on[2]) gram(std::integral_constant<int, 3>{});
    else if (on[1]) gram(std::integral_constant<int, 2>{});
    else gram(std::integral_constant<int, 1>{});
#pragma unroll
    for (int u = 0; u < 3; u++) {
      if (!on[u]) continue;
      const d4 acc = (u == 0) ? acc0 : (u == 1) ? acc1 : acc2;
      double* out = P.tiles + ((size_t)blockIdx.x * P.n_tiles + tile0 + u * (TPB / 64)) * 256 + lane * 4;
      if (later) { out[0] += acc[0]; out[1] += acc[1]; out[2] += acc[2]; out[3] += acc[3]; }
      else { out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; out[3] = acc[3]; }
    }
  }
  __syncthreads();                       // the panel has been read: the next chunk's camera-sum scratch may overwrite it
  }   // chunk
  VO_STAMP(dbgb, 5);   // Gram tiles (wave 0)
}

// ------------------------------------------------------------------------------------------------
// k_ba_reduce : fixed-order sum of the per-workgroup partials (Gram tiles, camera sums, max |g_l|), one output
// element per thread so that the one-workgroup solve reads ~23 KB instead of nblk x 23 KB.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ba_reduce(ba_ptrs Pall, int it) {
  const ba_ptrs P = ba_select(Pall, blockIdx.y);
  __shared__ double s_part[4][64];
  const ba_state st = P.state[it & 1];
  if (st.done) return;
  const int nset = Pall.gdyn ? Pall.gdyn[(it & 1) * Pall.batch + blockIdx.y] : P.nset;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;       // output within the workgroup, partial group
  const int e = blockIdx.x * 64 + o;
  const int n_tile_el = P.n_tiles * 256, n_pose_el = P.W * BA_POSE_VALS;
  const double* src = nullptr; size_t stride = 0; double* dst = nullptr;
  bool is_max = false;
  if (e < n_tile_el) { src = P.tiles + e; stride = (size_t)n_tile_el; dst = P.tilesum + e; }
  else if (e < n_tile_el + n_pose_el) { src = P.posepart + (e - n_tile_el); stride = (size_t)n_pose_el; dst = P.posesum + (e - n_tile_el); }
  else if (e == n_tile_el + n_pose_el) { src = P.gmax; stride = 1; dst = P.posesum + n_pose_el; is_max = true; }
  double acc = 0;
  if (src) {
    // group g sums partials g, g + 4, g + 8, ... (8 loads in flight), groups are combined in fixed order below
    double s0 = 0, s1 = 0;
    int b = g;
    for (; b + 60 < nset; b += 64) {   // 16 loads in flight
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = src[(size_t)(b + 4 * u) * stride];
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        if (is_max) { s0 = fmax(s0, fmax(v[u], v[u + 1])); }
        else { s0 += v[u]; s1 += v[u + 1]; }
      }
    }
    for (; b + 28 < nset; b += 32) {
      const double v0 = src[(size_t)b * stride], v1 = src[(size_t)(b + 4) * stride], v2 = src[(size_t)(b + 8) * stride];
      const double v3 = src[(size_t)(b + 12) * stride], v4 = src[(size_t)(b + 16) * stride], v5 = src[(size_t)(b + 20) * stride];
      const double v6 = src[(size_t)(b + 24) * stride], v7 = src[(size_t)(b + 28) * stride];
      if (is_max) { s0 = fmax(fmax(fmax(s0, v0), fmax(v1, v2)), fmax(fmax(v3, v4), fmax(fmax(v5, v6), v7))); }
      else { s0 += v0; s1 += v1; s0 += v2; s1 += v3; s0 += v4; s1 += v5; s0 += v6; s1 += v7; }
    }
    for (; b < nset; b += 4) { const double v = src[(size_t)b * stride]; if (is_max) s0 = fmax(s0, v); else s0 += v; }
    acc = is_max ? s0 : (s0 + s1);
  }
  s_part[g][o] = acc;
  __syncthreads();
  if (g == 0 && dst) {
    const double a0 = s_part[0][o], a1 = s_part[1][o], a2 = s_part[2][o], a3 = s_part[3][o];
    *dst = is_max ? fmax(fmax(a0, a1), fmax(a2, a3)) : ((a0 + a1) + (a2 + a3));
  }
}

// ------------------------------------------------------------------------------------------------
// sharded solve: the local half of the exchange.  k_ba_xsum sums the reduced packets of the `batch` shards of this
// GPU in shard order and stores the total in every entry (element-wise: one thread owns element e of all entries);
// max |g_l| is a maximum, so it travels as one slot per rank (own slot = local maximum, others 0) and the all-reduce
// SUM that follows delivers every rank's value to everybody.  k_ba_xstat does the same for the 4 step statistics.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ba_xsum(ba_ptrs Pall, int it) {
  if (Pall.state[it & 1].done) return;       // all shards hold the same state
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int n_sum = Pall.n_tiles * 256 + Pall.W * BA_POSE_VALS;      // tilesum | posesum are contiguous
  const size_t stride = Pall.s_tilesum;
  if (e < n_sum) {
    double s = 0;
    for (int b = 0; b < Pall.batch; b++) s += Pall.tilesum[(size_t)b * stride + e];
    for (int b = 0; b < Pall.batch; b++) Pall.tilesum[(size_t)b * stride + e] = s;
  } else if (e > n_sum && e <= n_sum + Pall.n_ranks) {
    const int r = e - n_sum - 1;             // rank slot; element n_sum itself (the local maximum) is left alone
    double m = 0;
    for (int b = 0; b < Pall.batch; b++) m = fmax(m, Pall.tilesum[(size_t)b * stride + n_sum]);
    const double v = (r == Pall.rank) ? m : 0.0;
    for (int b = 0; b < Pall.batch; b++) Pall.tilesum[(size_t)b * stride + e] = v;
  }
}

__global__ void __launch_bounds__(256) k_ba_xstat(ba_ptrs Pall, int it) {
  if (Pall.state[it & 1].done) return;
  __shared__ double s_sum[4];
  __shared__ double s_tot[4];
  if (threadIdx.x < 4) s_tot[threadIdx.x] = 0;
  for (int b = 0; b < Pall.batch; b++) {
    __syncthreads();
    ba_reduce_evalpart<256>(Pall.evalpart + (size_t)b * Pall.s_evalpart, Pall.n_eval, s_sum);
    if (threadIdx.x < 4) s_tot[threadIdx.x] += s_sum[threadIdx.x];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * Pall.batch; i += 256) Pall.xstat[i] = s_tot[i & 3];
}

// ------------------------------------------------------------------------------------------------
// k_ba_solve : one workgroup
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BA_SOLVE_THREADS) k_ba_solve(ba_ptrs Pall, ba_params_dev prm, int it, double* __restrict__ probe_S,
                                                         double* __restrict__ hpp_out) {
  const ba_ptrs P = ba_select(Pall, blockIdx.x);      // one workgroup per problem of the batch
  if (blockIdx.x != 0) { probe_S = nullptr; hpp_out = nullptr; }
  extern __shared__ double sm[];
  __shared__ int s_fail;
  const ba_state st = P.state[it & 1];
  if (st.done) return;
  const int tid = threadIdx.x;
  const int W = P.W, n = 6 * W, n1 = n + 1;
  const int PT = n1 | 1;                       // odd column pitch
  double* A = sm;                              // column-major lower triangle of [S rhs; rhs^T .], A[col * PT + row]
  double* s_hpp = A + (size_t)n1 * PT;         // W * 28 reduced camera sums
  double* s_invd = s_hpp + W * BA_POSE_VALS;   // n inverse diagonal of L
  double* s_dp = s_invd + n1;                  // n (+ 24 scratch)
  // (row | col << 8) of the lower-triangle elements ordered by column DESCENDING (rows ascending inside a column):
  // the trailing submatrix of every block column is then a prefix of the table, and consecutive threads get
  // consecutive rows of one column (conflict-free LDS) without any index arithmetic in the factorisation loop
  double* s_linv = s_dp + n1 + 24;             // W x 21: inverse of every diagonal block of the factor (lower triangle, row-major packed)
  unsigned short* s_tab = reinterpret_cast<unsigned short*>(s_linv + 21 * W);
  for (int e = tid; e < (n1 * (n1 + 1)) / 2; e += BA_SOLVE_THREADS) {
    int k = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);     // column n - k holds k + 1 elements
    while (k > 0 && (k * (k + 1)) / 2 > e) k--;
    while (((k + 1) * (k + 2)) / 2 <= e) k++;
    const int col = n - k, row = col + (e - (k * (k + 1)) / 2);
    s_tab[e] = (unsigned short)(row | (col << 8));
  }
  if (tid == 0) s_fail = 0;
  unsigned long long* dbgs = P.dbg ? P.dbg + 8 : nullptr;
  VO_STAMP(dbgs, 0);
  // ---- reduced camera sums; -E (lower triangle) and +r (row n) from the reduced Gram tiles ----
  // FOLD (P.fold: small windows, one problem per GPU or a batch): the partial sets of k_ba_build are summed right here, every thread the
  // one or two values it is about to use -- a window of 4 needs 325 Gram entries + 112 camera sums from 32 sets, one trip of 32 loads in
  // flight --, and the k_ba_reduce launch between build and solve is gone: one kernel boundary less in every LM iteration, and one
  // early-exit launch less in every group a finished problem still sees.  Fixed order (two interleaved chains over the sets), so a solve
  // stays bitwise reproducible; the sum differs in the last bits from k_ba_reduce's four-group order.
  const bool fold = P.fold != 0;
  const int nset = Pall.gdyn ? Pall.gdyn[(it & 1) * Pall.batch + blockIdx.x] : P.nset;
  auto set_sum = [&](const double* __restrict__ src, const size_t stride) -> double {
    double s0 = 0, s1 = 0;
    int bset = 0;
#pragma unroll 8
    for (; bset + 1 < nset; bset += 2) { s0 += src[(size_t)bset * stride]; s1 += src[(size_t)(bset + 1) * stride]; }
    if (bset < nset) s0 += src[(size_t)bset * stride];
    return s0 + s1;
  };
  const size_t n_tile_el = (size_t)P.n_tiles * 256, n_pose_el = (size_t)W * BA_POSE_VALS;
  for (int q = tid; q < W * BA_POSE_VALS; q += BA_SOLVE_THREADS) s_hpp[q] = fold ? set_sum(P.posepart + q, n_pose_el) : P.posesum[q];
  for (int e = tid; e < n1 * n1; e += BA_SOLVE_THREADS) {
    const int col = e / n1, row = e - col * n1;       // consecutive threads -> consecutive rows
    if (row < col) continue;
    if (row == n && col == n) { A[(size_t)col * PT + row] = 1.0; continue; }
    // element (a = col, b = row), a <= b, lives in upper tile (ta, tb)
    const int ta = col >> 4, tb = row >> 4, ii = col & 15, jj = row & 15;
    const int tile = ta * P.RT - (ta * (ta - 1)) / 2 + (tb - ta);
    const size_t idx = (size_t)tile * 256 + ((ii & 3) * 16 + jj) * 4 + (ii >> 2);
    const double v = fold ? set_sum(P.tiles + idx, n_tile_el) : P.tilesum[idx];
    A[(size_t)col * PT + row] = (row == n) ? v : -v;
  }
  if (fold && tid == BA_SOLVE_THREADS - 1) {           // max |g_l| over the sets (what k_ba_reduce leaves behind the camera sums)
    double gm = 0;
    for (int bset = 0; bset < nset; bset++) gm = fmax(gm, P.gmax[bset]);
    s_dp[n + 16] = gm;                                 // (scratch behind dp: entries n .. n + 15 carry the step statistics below)
  }
  __syncthreads();
  VO_STAMP(dbgs, 1);   // partials reduced
  // ---- + damped Hpp blocks, rhs = -gp + r ----
  const double lam = st.lambda;
  for (int q = tid; q < W * 36; q += BA_SOLVE_THREADS) {
    const int sl = q / 36, rr = (q % 36) / 6, cc = q % 6;
    if (cc > rr) continue;
    const int idx = cc * 6 - (cc * (cc - 1)) / 2 + (rr - cc);   // upper-packed (cc, rr)
    double v = s_hpp[sl * BA_POSE_VALS + idx];
    if (rr == cc) v += lam * fmax(v, 1e-12);
    A[(size_t)(6 * sl + cc) * PT + 6 * sl + rr] += v;
  }
  for (int a = tid; a < n; a += BA_SOLVE_THREADS) A[(size_t)a * PT + n] -= s_hpp[(a / 6) * BA_POSE_VALS + 21 + a % 6];
  __syncthreads();
  if (probe_S) {   // reduced camera system before factorisation (parity probe)
    for (int e = tid; e < n * n; e += BA_SOLVE_THREADS) {
      const int a = e / n, b = e - a * n;
      probe_S[e] = (b <= a) ? A[(size_t)b * PT + a] : A[(size_t)a * PT + b];
    }
    for (int a = tid; a < n; a += BA_SOLVE_THREADS) probe_S[(size_t)n * n + a] = A[(size_t)a * PT + n];
  }
  if (hpp_out) for (int q = tid; q < W * BA_POSE_VALS; q += BA_SOLVE_THREADS) hpp_out[q] = s_hpp[q];
  if (probe_S) __syncthreads();   // the factorisation below overwrites A in place
  VO_STAMP(dbgs, 2);   // system assembled
  // ---- 6x6-blocked right-looking Cholesky of the augmented matrix (row n carries rhs -> y), with LOOKAHEAD: the rank-6 update of
  //      block column kb is applied to the columns of block kb + 1 first (one short pass of all threads); then wave 0 factorises
  //      panel kb + 1 -- a chain of dependent float64 operations, the critical path of the kernel -- WHILE the other 15 waves
  //      apply the update to the rest of the trailing matrix.  Same operations on every element in the same order as the plain
  //      loop (panel, barrier, whole trailing update, barrier), so the factor is bit-identical; the chain and the bulk update
  //      now overlap instead of alternating (2.4 k -> 1.4 k cycles per block column). ----
  auto panel = [&](const int kb) {
    const int c0 = 6 * kb;
    const int r = c0 + tid;
    if (r < n1) {
      double D[6][6], inv[6], x[6];
#pragma unroll
      for (int c = 0; c < 6; c++)
#pragma unroll
        for (int d = 0; d < 6; d++) D[c][d] = (d <= c) ? A[(size_t)(c0 + d) * PT + c0 + c] : 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) x[c] = (r >= c0 + c) ? A[(size_t)(c0 + c) * PT + r] : 0.0;
      // RIGHT-LOOKING inside the block: as soon as column c is known it is subtracted from everything to its right,
      // so every pivot waits for one FMA after the previous column instead of a c-term dot product (a dependent f64
      // op costs ~30 cycles for a lone wave; this chain is the critical path of the whole kernel).  The row x
      // (this thread's row of the panel below / inside the block) is eliminated the same way.
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double s = D[c][c];
        if (!(s > 0)) { bad = true; s = 1.0; }
        const double rs = rsqrt_nr(s);
        inv[c] = rs;
        x[c] *= rs;
#pragma unroll
        for (int e = c + 1; e < 6; e++) D[e][c] *= rs;
#pragma unroll
        for (int e = c + 1; e < 6; e++) {
          x[e] -= x[c] * D[e][c];
#pragma unroll
          for (int f = c + 1; f <= e; f++) D[e][f] -= D[e][c] * D[f][c];
        }
      }
      if (bad && tid == 0) s_fail = 1;
      if (tid == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) s_invd[c0 + c] = inv[c];
      }
      // store the row (for the rows of the diagonal block itself only the lower part: there x reproduces the factor's
      // own row, operation for operation)
#pragma unroll
      for (int c = 0; c < 6; c++)
        if (r >= c0 + c) A[(size_t)(c0 + c) * PT + r] = x[c];
    }
  };
  // rank-6 update by block column kb of the table entries [lo, hi), shared out over the threads t0, t0 + 1, ... (nt of them)
  auto trail = [&](const int kb, const int lo, const int hi, const int t0, const int nt) {
    const int c0 = 6 * kb;
    for (int idx = lo + (tid - t0); idx < hi; idx += nt) {
      const int ij = s_tab[idx];
      const int i = ij & 255, jcol = ij >> 8;
      double acc = A[(size_t)jcol * PT + i];
#pragma unroll
      for (int c = 0; c < 6; c++) acc -= A[(size_t)(c0 + c) * PT + i] * A[(size_t)(c0 + c) * PT + jcol];
      A[(size_t)jcol * PT + i] = acc;
    }
  };
  // inverse of the factored diagonal block L_kk (lower triangular) for the back substitution, where the 6 x 6 triangular solve is
  // otherwise a chain of 27 dependent float64 operations per block column; with the inverse it is six independent dot products.
  // Formed off the critical path: by the LAST wave while wave 0 factorises the next panel.
  auto linv = [&](const int kb) {
    const int c0 = 6 * kb;
    double Lk[6][6], iv[6], Li[6][6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
      iv[c] = s_invd[c0 + c];
#pragma unroll
      for (int e = c + 1; e < 6; e++) Lk[e][c] = A[(size_t)(c0 + c) * PT + c0 + e];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) {
      Li[c][c] = iv[c];
#pragma unroll
      for (int e = c + 1; e < 6; e++) {
        double acc = 0.0;
#pragma unroll
        for (int k = c; k < e; k++) acc += Lk[e][k] * Li[k][c];
        Li[e][c] = -iv[e] * acc;
      }
    }
    if (tid == BA_SOLVE_THREADS - 64) {
      int q = 0;
#pragma unroll
      for (int e = 0; e < 6; e++)
#pragma unroll
        for (int c = 0; c <= e; c++) s_linv[21 * kb + q++] = Li[e][c];
    }
  };
  panel(0);
  __syncthreads();
  for (int kb = 0; kb + 1 < W; kb++) {
    // the table lists the lower triangle by column DESCENDING: entries [0, rest) are the columns behind block kb + 1,
    // [rest, full) the columns of block kb + 1 (entry 0 is (n, n): not needed)
    const int m = n1 - (6 * kb + 6), mp = m - 6;
    const int full = (m * (m + 1)) / 2, rest = (mp * (mp + 1)) / 2;
    trail(kb, rest, full, 0, BA_SOLVE_THREADS);
    __syncthreads();
    const int pw = (m + 63) & ~63;                 // the panel of block kb + 1 has m rows: whole waves (one, two above W = 10)
    if (tid < pw) panel(kb + 1);
    else {
      if (tid >= BA_SOLVE_THREADS - 64) linv(kb);
      trail(kb, 1, rest, pw, BA_SOLVE_THREADS - pw);
    }
    __syncthreads();
  }
  if (tid >= BA_SOLVE_THREADS - 64) linv(W - 1);
  __syncthreads();
  VO_STAMP(dbgs, 3);   // factorised
  // ---- back substitution  L^T dp = y  by wave 0, 6x6 block at a time (lane i holds rows i and i + 64):
  //      every lane solves the 6x6 triangular block redundantly in registers (operands are wave-uniform LDS
  //      broadcasts), then the lanes above the block subtract their 6-term update -> W steps instead of 6W ----
  if (tid < 64) {
    const int lane = tid;
    // SMALL (n <= 64, W <= 10): every row lives in y0 -- no second row per lane, no choice of the source register per entry
    auto backsub = [&](auto small_tag) {
      constexpr bool SMALL = decltype(small_tag)::value;
      const int la = min(lane, n - 1), lb = min(lane + 64, n - 1);            // clamped rows: the loads are unconditional, masked after
      double y0 = (lane < n) ? A[(size_t)la * PT + n] : 0.0;
      double y1 = (!SMALL && lane + 64 < n) ? A[(size_t)lb * PT + n] : 0.0;
      // the LDS operands of a block do not depend on the running y.  This lane's column entries of the NEXT block are requested
      // before this block's arithmetic (prefetching the 21 entries of the next diagonal-block inverse as well needs 160 live VGPRs
      // in a 1024-lane workgroup: it spilled, 14.5 k -> 55 k cycles)
      double ca[6], cb[6];
      auto load_cols = [&](const int kb, double (&a)[6], double (&b)[6]) {
        const int c0 = 6 * kb;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double va = A[(size_t)la * PT + c0 + c];
          a[c] = (lane < c0) ? va : 0.0;
          if (!SMALL) { const double vb = A[(size_t)lb * PT + c0 + c]; b[c] = (lane + 64 < c0) ? vb : 0.0; } else b[c] = 0.0;
        }
      };
      load_cols(W - 1, ca, cb);
      for (int kb = W - 1; kb >= 0; kb--) {
        const int c0 = 6 * kb;
        double Li[21];
#pragma unroll
        for (int q = 0; q < 21; q++) Li[q] = s_linv[21 * kb + q];
        double nca[6], ncb[6];
        load_cols(max(kb - 1, 0), nca, ncb);
        double yb[6], d[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const int k = c0 + c;
          if (SMALL) yb[c] = readlane_f64(y0, k);
          else { const double r0 = readlane_f64(y0, k & 63), r1 = readlane_f64(y1, k & 63); yb[c] = (k < 64) ? r0 : r1; }
        }
        // d = L_kk^-T yb: six independent dot products (Li: lower triangle, row-major packed: entry (e, c) at e (e + 1) / 2 + c)
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double t = Li[(c * (c + 1)) / 2 + c] * yb[c];
#pragma unroll
          for (int e = c + 1; e < 6; e++) t += Li[(e * (e + 1)) / 2 + c] * yb[e];
          d[c] = t;
        }
        // rows above the block: y_i -= sum_c L[c0 + c][i] d_c
#pragma unroll
        for (int c = 0; c < 6; c++) { y0 -= ca[c] * d[c]; if (!SMALL) y1 -= cb[c] * d[c]; }
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const int k = c0 + c;
          if (SMALL) y0 = (lane == k) ? d[c] : y0;
          else { y0 = (lane == k) ? d[c] : y0; y1 = (lane + 64 == k) ? d[c] : y1; }
        }
#pragma unroll
        for (int c = 0; c < 6; c++) { ca[c] = nca[c]; cb[c] = ncb[c]; }
      }
      if (lane < n) s_dp[lane] = y0;
      if (!SMALL && lane + 64 < n) s_dp[lane + 64] = y1;
    };
    if (n <= 64) backsub(std::true_type{}); else backsub(std::false_type{});
  }
  __syncthreads();
  VO_STAMP(dbgs, 4);   // back substitution
  // ---- publish ----
  const int fail = s_fail;
  for (int a = tid; a < n; a += BA_SOLVE_THREADS) P.dp[a] = fail ? 0.0 : s_dp[a];
  // cameras of the trial poses for k_ba_update / the next k_ba_build: cams[cur ^ 1] <-> x[cur ^ 1]
  if (tid >= 128 && tid < 128 + W) {
    const int i = tid - 128;
    const double* pc = ba_x(P, st.cur) + 6 * i;
    double pt6[6];
#pragma unroll
    for (int a = 0; a < 6; a++) pt6[a] = pc[a] + (fail ? 0.0 : s_dp[6 * i + a]);
    double* cg = P.cams + ((size_t)(st.cur ^ 1) * W + i) * BA_CAM;
    d_camera(pt6, cg);
  }
  if (tid < 128) {
    // wave-parallel step statistics of the camera block (lanes = parameters)
    const double* poses = ba_x(P, st.cur);
    double pred = 0, step2 = 0, x2 = 0, gabs = 0, cost = 0;
    if (tid < n) {
      const int i = tid / 6, a = tid - 6 * i;
      const double g = s_hpp[i * BA_POSE_VALS + 21 + a];
      const double Dg = fmax(s_hpp[i * BA_POSE_VALS + a * 6 - (a * (a - 1)) / 2], 1e-12);
      const double d = fail ? 0.0 : s_dp[tid];
      const double xv = (it == 0) ? P.x0[tid] : poses[tid];
      pred = lam * Dg * d * d - g * d; step2 = d * d; x2 = xv * xv; gabs = fabs(g);
      if (a == 0) cost = s_hpp[i * BA_POSE_VALS + 27];
    }
    pred = wave_allreduce(pred); step2 = wave_allreduce(step2); x2 = wave_allreduce(x2); cost = wave_allreduce(cost);
    for (int ofs = 32; ofs > 0; ofs >>= 1) gabs = fmax(gabs, __shfl_xor(gabs, ofs));
    if ((tid & 63) == 0) {
      double* w = s_dp + n + 8 * (tid >> 6);   // scratch behind dp (s_dp has n1 + 8.. entries reserved)
      w[0] = pred; w[1] = step2; w[2] = x2; w[3] = cost; w[4] = gabs;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const double* w0 = s_dp + n; const double* w1 = s_dp + n + 8;
    ba_info inf;
    inf.cost_cur = w0[3] + w1[3]; inf.pred_pose = w0[0] + w1[0]; inf.step2_pose = w0[1] + w1[1]; inf.x2_pose = w0[2] + w1[2];
    double gl_max = 0;                       // max |g_l| over all landmarks
    if (P.sharded) { for (int r = 0; r < P.n_ranks; r++) gl_max = fmax(gl_max, P.posesum[W * BA_POSE_VALS + 1 + r]); }
    else gl_max = fold ? s_dp[n + 16] : P.posesum[W * BA_POSE_VALS];
    inf.ginf = fmax(fmax(w0[4], w1[4]), gl_max);
    inf.chol_fail = fail; inf.pad = 0;
    *P.info = inf;
  }
  VO_STAMP(dbgs, 5);
}

// ------------------------------------------------------------------------------------------------
// k_ba_update : back-substitute landmarks, form the trial x, evaluate the trial cost
// ------------------------------------------------------------------------------------------------
template <int TPB, int LPPC = 0>
__global__ void __launch_bounds__(TPB) k_ba_update(ba_ptrs Pall, ba_params_dev prm, int it, double* __restrict__ probe_dl) {
  const ba_ptrs P = ba_select(Pall, blockIdx.y);
  if (blockIdx.y != 0) probe_dl = nullptr;
  __shared__ double s_cam[BA_CAM * BA_MAX_SLOTS];    // current poses
  __shared__ double s_camt[BA_CAM * BA_MAX_SLOTS];   // trial poses
  __shared__ double s_K[9];
  __shared__ double s_dp[6 * BA_MAX_SLOTS];
  __shared__ double s_pose[6 * BA_MAX_SLOTS];
  __shared__ double s_red[(TPB / 64) * BA_EVAL_VALS];
  const ba_state st = P.state[it & 1];
  if (st.done) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, W = P.W, N = P.N, LPP = LPPC ? LPPC : P.LPP;
  if (P.n_live && blockIdx.x > 0 && (int)blockIdx.x * P.PPB >= *P.n_live) {     // an unused part of the table (workgroup 0 carries the trial poses)
    if (tid < BA_EVAL_VALS) P.evalpart[blockIdx.x * BA_EVAL_VALS + tid] = 0.0;
    return;
  }
  const double* poses = ba_x(P, st.cur);
  const double* pts = poses + 6 * W;
  double* tposes = ba_x(P, st.cur ^ 1);
  double* tpts = tposes + 6 * W;
  // this lane's landmark and observation first: their HBM latency overlaps the staging below
  const int pl = tid / LPP, slot = tid - pl * LPP;
  const int j = blockIdx.x * P.PPB + pl;
  double X[3] = {0, 0, 0};
  double uo = __builtin_nan(""), vo = 0;
  if (j < N) {
    X[0] = pts[3 * j]; X[1] = pts[3 * j + 1]; X[2] = pts[3 * j + 2];
    if (slot < W) {
      const double* ob = P.obs + ((size_t)slot * N + j) * 2;
      uo = ob[0]; vo = ob[1];
    }
  }
  for (int a = tid; a < 6 * W; a += TPB) {
    const double d = P.dp[a];
    s_dp[a] = d;
    s_pose[a] = poses[a] + d;
    if (blockIdx.x == 0) tposes[a] = poses[a] + d;
  }
  if (tid < 9) s_K[tid] = P.K[tid];
  {
    const double* cc = P.cams + (size_t)st.cur * W * BA_CAM;          // current poses (k_ba_build it == 0 / k_ba_solve)
    const double* ct = P.cams + (size_t)(st.cur ^ 1) * W * BA_CAM;    // trial poses (k_ba_solve of this iteration)
    for (int i = tid; i < W * BA_CAM; i += TPB) { s_cam[i] = cc[i]; s_camt[i] = ct[i]; }
  }
  __syncthreads();
  double v0 = 0, v1 = 0, v2 = 0;
  if (j < N) {
    if (slot < W) {
      ba_obs_lin o;
      double Aj[2][3];
      if (ba_linearize_obs<false>(s_K, s_cam + BA_CAM * slot, X, uo, vo, prm.delta, o, Aj)) {
        // Jp d without forming Jp = [-Jl [X]x Jr | A]:  Jp_rot d_rot = -Jl (X x (Jr d_rot)),  Jp_trans d_trans = A d_trans
        const double* d = s_dp + 6 * slot;
        const double* Jr = s_cam + BA_CAM * slot + 12;
        const double u0 = Jr[0] * d[0] + Jr[1] * d[1] + Jr[2] * d[2];
        const double u1 = Jr[3] * d[0] + Jr[4] * d[1] + Jr[5] * d[2];
        const double u2 = Jr[6] * d[0] + Jr[7] * d[1] + Jr[8] * d[2];
        const double t0 = X[1] * u2 - X[2] * u1, t1 = X[2] * u0 - X[0] * u2, t2 = X[0] * u1 - X[1] * u0;
        const double q0 = (Aj[0][0] * d[3] + Aj[0][1] * d[4] + Aj[0][2] * d[5]) - (o.Jl[0][0] * t0 + o.Jl[0][1] * t1 + o.Jl[0][2] * t2);
        const double q1 = (Aj[1][0] * d[3] + Aj[1][1] * d[4] + Aj[1][2] * d[5]) - (o.Jl[1][0] * t0 + o.Jl[1][1] * t1 + o.Jl[1][2] * t2);
        v0 = o.w * (o.Jl[0][0] * q0 + o.Jl[1][0] * q1);     // B^T d_pose = w Jl^T (Jp d_pose)
        v1 = o.w * (o.Jl[0][1] * q0 + o.Jl[1][1] * q1);
        v2 = o.w * (o.Jl[0][2] * q0 + o.Jl[1][2] * q1);
      }
    }
  }
  v0 = group_allreduce_t<LPPC>(v0, LPP); v1 = group_allreduce_t<LPPC>(v1, LPP); v2 = group_allreduce_t<LPPC>(v2, LPP);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  if (j < N) {
    const double* ax = P.aux + (size_t)j * BA_AUX;
    const double u0 = ax[6] + v0, u1 = ax[7] + v1, u2 = ax[8] + v2;
    const double i00 = ax[9], i10 = ax[10], i11 = ax[11], i20 = ax[12], i21 = ax[13], i22 = ax[14];
    const double t0 = i00 * u0, t1 = i10 * u0 + i11 * u1, t2 = i20 * u0 + i21 * u1 + i22 * u2;   // Cinv u
    const double dl0 = -(i00 * t0 + i10 * t1 + i20 * t2), dl1 = -(i11 * t1 + i21 * t2), dl2 = -(i22 * t2);
    const double Xt[3] = {X[0] + dl0, X[1] + dl1, X[2] + dl2};
    if (slot < W) {
      ba_obs_lin o;
      if (ba_linearize_obs<false>(s_K, s_camt + BA_CAM * slot, Xt, uo, vo, prm.delta, o)) e0 = 0.5 * o.rho;
    }
    if (slot == 0) {
      tpts[3 * j] = Xt[0]; tpts[3 * j + 1] = Xt[1]; tpts[3 * j + 2] = Xt[2];
      if (probe_dl) { probe_dl[3 * j] = dl0; probe_dl[3 * j + 1] = dl1; probe_dl[3 * j + 2] = dl2; }
      const double lam = st.lambda;
      e1 = lam * (fmax(ax[0], 1e-12) * dl0 * dl0 + fmax(ax[2], 1e-12) * dl1 * dl1 + fmax(ax[5], 1e-12) * dl2 * dl2)
           - (ax[6] * dl0 + ax[7] * dl1 + ax[8] * dl2);
      e2 = dl0 * dl0 + dl1 * dl1 + dl2 * dl2;
      e3 = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
    }
  }
  // four wave-wide sums as a reduce-scatter (21 cross-lane instructions instead of 4 x 18): after the two swap stages the 16-lane
  // row r of the wave holds, per lane, the column sums of value {e0, e2, e1, e3}[r]; one row all-reduce finishes all four at once
  {
    double u = rs16_sum(rs32_sum(e0, e1), rs32_sum(e2, e3));
    u += dpp_f64<0x128>(u); u += dpp_f64<0x124>(u); u += dpp_f64<0x122>(u); u += dpp_f64<0x121>(u);   // row_ror 8, 4, 2, 1
    if ((lane & 15) == 0) {
      const int r = lane >> 4;
      s_red[wave * 4 + ((r & 1) ? (r == 1 ? 2 : 3) : (r == 0 ? 0 : 1))] = u;
    }
  }
  __syncthreads();
  if (tid < BA_EVAL_VALS) {
    double s = 0;
    for (int wv = 0; wv < (TPB / 64); wv++) s += s_red[wv * 4 + tid];
    P.evalpart[blockIdx.x * BA_EVAL_VALS + tid] = s;
  }
}

// grid = batch; x_out / st_out of problem b: pub buffer b ([state 64 B | x]) resp. st_out + b * st_stride
__global__ void __launch_bounds__(256) k_ba_finalize(ba_ptrs Pall, ba_params_dev prm, int n_it, unsigned char* __restrict__ pub,
                                                     size_t pub_stride, ba_state* __restrict__ st_out, int st_stride) {
  const ba_ptrs P = ba_select(Pall, blockIdx.x);
  double* x_out = reinterpret_cast<double*>(pub + (size_t)blockIdx.x * pub_stride + VO_BA_PUB_HEADER);
  st_out += (size_t)blockIdx.x * st_stride;
  __shared__ ba_state s_st;
  __shared__ double s_esum[4];
  if (n_it > 0) ba_reduce_evalpart<256>(P.sharded ? P.xstat : P.evalpart,
                                        P.sharded ? 1 : (Pall.gdyn ? Pall.gdyn[((n_it - 1) & 1) * Pall.batch + blockIdx.x] : P.n_eval), s_esum);
  // gridDim.y workgroups per problem share the copy of x (one 256-thread workgroup took 10 us for 6 000 doubles, at the end of
  // the critical path of a step); each derives the final state itself (deterministic), the first one publishes it
  if (threadIdx.x == 0) {
    ba_state st;
    if (n_it == 0) st = ba_init_state(prm);
    else ba_decide(P.state[(n_it - 1) & 1], *P.info, s_esum, prm, st);
    s_st = st;
    if (blockIdx.y == 0) {
      *st_out = st;
      P.state[n_it & 1] = st;
      *reinterpret_cast<ba_state*>(reinterpret_cast<unsigned char*>(x_out) - VO_BA_PUB_HEADER) = st;   // pub header of this problem
    }
  }
  __syncthreads();
  const double* x = (n_it == 0) ? P.x0 : ba_x(P, s_st.cur);
  const int total = 6 * P.W + 3 * P.N;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < total; i += blockDim.x * gridDim.y) x_out[i] = x[i];
}

// per-observation residual norms at x (dense [W][N], NaN where unobserved) -- parity probe
__global__ void __launch_bounds__(128) k_ba_residual(ba_ptrs Pall, const double* __restrict__ x, double delta,
                                                     double* __restrict__ res) {
  const ba_ptrs P = ba_select(Pall, blockIdx.y);
  x += (size_t)blockIdx.y * Pall.s_x; res += (size_t)blockIdx.y * ((size_t)Pall.W * Pall.N);
  __shared__ double s_cam[BA_CAM * BA_MAX_SLOTS];
  __shared__ double s_K[9];
  const int tid = threadIdx.x;
  stage_cameras(x, P.W, s_cam, tid, 128);
  if (tid < 9) s_K[tid] = P.K[tid];
  __syncthreads();
  const int j = blockIdx.x * 128 + tid;
  if (j >= P.N) return;
  const double* pts = x + 6 * P.W;
  const double X[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
  for (int i = 0; i < P.W; i++) {
    const double* ob = P.obs + ((size_t)i * P.N + j) * 2;
    ba_obs_lin o;
    double r = __builtin_nan("");
    if (ba_linearize_obs<false>(s_K, s_cam + BA_CAM * i, X, ob[0], ob[1], delta, o)) r = sqrt(o.e0 * o.e0 + o.e1 * o.e1);
    res[(size_t)i * P.N + j] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void vo_ba_destroy(vo_ctx* c) {
  if (!c->ba) return;
  vo_ba_ws* b = c->ba;
  if (b->d_x0_own) { b->d_x0 = b->d_x0_own; b->d_obs = b->d_obs_own; }
  if (b->d_bank_x0) (void)hipFree(b->d_bank_x0);
  if (b->d_bank_obs) (void)hipFree(b->d_bank_obs);
  void* bufs[] = {b->d_K, b->d_obs, b->d_x0, b->d_x[0], b->d_x[1], b->d_aux, b->d_posepart, b->d_gmax,
                  b->d_tiles, b->d_dp, b->d_evalpart, b->d_tilesum, b->d_xstat, b->d_gather, b->d_cams, b->d_S, b->d_Hpp, b->d_res, b->d_dl, b->d_pub, b->d_state, b->d_info, b->d_gdyn};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (b->h_gather) (void)hipHostFree(b->h_gather);
  if (b->h_state) (void)hipHostFree(b->h_state);
  if (b->h_pub) (void)hipHostFree(b->h_pub);
  delete b;
  c->ba = nullptr;
}

bool vo_ba_ready(const vo_ctx* c) { return c->ba && c->ba->uploaded; }
double* vo_ba_obs_device(vo_ctx* c, int* n_slots, int* n_pts) {
  if (!vo_ba_ready(c)) return nullptr;
  *n_slots = c->ba->W; *n_pts = c->ba->N;
  return c->ba->d_obs;
}

extern "C" int32_t vo_ba_default_params(vo_ba_params* p) {
  if (!p) return VO_E_INVALID;
  p->max_iters = 50; p->_pad = 0; p->ftol = 1e-3; p->xtol = 1e-3; p->gtol = 1e-8; p->lambda0 = 1e-4; p->huber_delta = 1.0; p->lambda_min = 1e-3;
  return VO_OK;
}

static void ba_geometry(vo_ba_ws* b, int W, int N, const vo_tuning& tn) {
  b->W = W; b->N = N;
  b->LPP = (W <= 8) ? 8 : (W <= 16) ? 16 : 32;
  if (tn.ba_lanes == 16 && b->LPP == 8) b->LPP = 16;                                              // (vo_tuning: A/B)
  // workgroup size: 256 lanes (more workgroups -> more CUs, less contention on the f64 pipes) unless that would
  // produce more than 160 partial sets, then 1024
  // (8-lane groups -- windows of <= 8 slots -- keep 256 lanes whatever N: with 1 024-lane workgroups they fell back to 16 lanes per landmark,
  //  three quarters of them idle at the reference's window of 4; the chunk walk of k_ba_build bounds the partial sets instead.  Measured on the
  //  closed loop's 8 192-slot tables: k_ba_build 78 us -> see DESIGN section 4)
  b->tpb = (b->LPP == 8 || vo_div_up(N, 256 / b->LPP) <= 160) ? 256 : 1024;
  if (tn.ba_threads == 256 || tn.ba_threads == 512 || tn.ba_threads == 1024) b->tpb = tn.ba_threads;                    // (vo_tuning)
  if (b->LPP == 8 && b->tpb != 256) { b->LPP = 16; b->tpb = (vo_div_up(N, 256 / b->LPP) <= 160) ? 256 : 1024; }          // (8-lane groups exist for 256-lane workgroups only)
  b->PPB = b->tpb / b->LPP;
  b->nblk = vo_div_up(N, b->PPB);
  b->RP = ((6 * W + 1 + 15) / 16) * 16; b->RT = b->RP / 16; b->n_tiles = b->RT * (b->RT + 1) / 2;
  b->pitch = b->RP + BA_PITCH_PAD;
  if (tn.ba_pitch_pad > 0) b->pitch = b->RP + tn.ba_pitch_pad - 1;           // (vo_tuning)
  const size_t panel = sizeof(double) * (size_t)3 * b->PPB * b->pitch;
  const size_t scratch = sizeof(double) * (size_t)(b->tpb / 64) * b->LPP * BA_POSE_VALS;
  b->build_lds = panel > scratch ? panel : scratch;
  b->cam_off = (int)(b->build_lds / sizeof(double));
  b->build_lds += sizeof(double) * (size_t)BA_CAM * W;
  // wave-private kernels: RT column blocks (6 W + 1 <= 16 RT <= 64), a lane serves SPL = ceil(W / 8) slots.  vo_tuning.ba_kernels = 1: the older kernels
  b->v2 = (W <= 10 && tn.ba_kernels != 1) ? 1 : 0;
  // (one problem alone is the one shape the older kernels still win: build 12 against 16 us per iteration -- a wave-private workgroup spends
  //  ~7 us on its LM decision, camera staging and the four-wave fold of its tiles whatever it walks; 18 against 17 at four problems, 65 against
  //  49 at 32: ba_alloc switches contexts of one or two sequences with windows of 9-10 slots back to them)
  b->v2_rt = b->RT; b->v2_spl = (W + 7) / 8;
  // windows of 9 and 10 slots: 5 lanes per landmark (12 landmarks per wave), else 8 (vo_tuning.ba_lanes = 8: 8 for every window, A/B)
  // and 4 for windows of <= 4 slots (16 landmarks per wave)
  b->v2_lpp = (b->v2_spl == 2) ? 5 : (W <= 4) ? 4 : 8;
  if (tn.ba_lanes == 8 && b->v2_lpp == 5) b->v2_lpp = 8;
  b->v2_lds = b->v2_lpp == 5 ? sizeof(double) * ((size_t)4 * 36 * (16 * b->v2_rt) + (size_t)BA2_CAM * W)
                             : sizeof(double) * ((size_t)4 * 3 * (64 / b->v2_lpp) * (16 * b->v2_rt + 16) + (size_t)BA2_CAM * W);
  const int n1 = 6 * W + 1, PT = n1 | 1;
  b->solve_lds = sizeof(double) * ((size_t)n1 * PT + (size_t)W * BA_POSE_VALS + n1 + n1 + 24 + (size_t)21 * W) + sizeof(unsigned short) * ((size_t)n1 * (n1 + 1) / 2 + 8);
}

// every buffer: [batch] x per-problem size
static int32_t ba_alloc(vo_ctx* c, int W, int N) {
  VO_CHECK(c, W >= 1 && W <= BA_MAX_SLOTS, VO_E_CAPACITY, "window size must be 1..20");
  VO_CHECK(c, N >= 1, VO_E_INVALID, "no landmarks");
  if (c->ba && (c->ba->cap_W != W || c->ba->cap_N < N)) vo_ba_destroy(c);
  if (c->ba) {
    // the workgroup size switches with N (ba_geometry), so a SMALLER N can need MORE partial sets than the N the
    // workspace was sized for (W = 4: N = 2600 -> 41 sets of 64 landmarks, N = 2500 -> 157 sets of 16): rebuild then
    vo_ba_ws probe;
    ba_geometry(&probe, W, N, c->tune);
    if (probe.nblk > c->ba->cap_nblk) vo_ba_destroy(c);
  }
  const size_t B = (size_t)c->batch;
  if (!c->ba) {
    vo_ba_ws* b = new vo_ba_ws();
    c->ba = b;
    b->cap_W = W; b->cap_N = N;
    ba_geometry(b, W, N, c->tune);
    // any N' <= N runs with 256-lane workgroups when that gives <= 160 sets, else with the 1024-lane ones of N at most
    {
      const int small = vo_div_up(N, 256 / b->LPP);
      b->cap_nblk = b->nblk > (small < 160 ? small : 160) ? b->nblk : (small < 160 ? small : 160);
    }
    const int nblk_alloc = b->cap_nblk;
    const size_t nx = (size_t)6 * W + 3 * N;
    VO_HIP(c, hipMalloc((void**)&b->d_K, 9 * sizeof(double) * B));
    VO_HIP(c, hipMalloc((void**)&b->d_obs, sizeof(double) * 2 * W * N * B));
    VO_HIP(c, hipMalloc((void**)&b->d_x0, sizeof(double) * nx * B));
    VO_HIP(c, hipMalloc((void**)&b->d_x[0], sizeof(double) * nx * B));
    VO_HIP(c, hipMalloc((void**)&b->d_x[1], sizeof(double) * nx * B));
    VO_HIP(c, hipMalloc((void**)&b->d_aux, sizeof(double) * (size_t)N * BA_AUX * B));
    VO_HIP(c, hipMalloc((void**)&b->d_posepart, sizeof(double) * (size_t)nblk_alloc * W * BA_POSE_VALS * B));
    VO_HIP(c, hipMalloc((void**)&b->d_gmax, sizeof(double) * nblk_alloc * B));
    VO_HIP(c, hipMalloc((void**)&b->d_tiles, sizeof(double) * (size_t)nblk_alloc * b->n_tiles * 256 * B));
    VO_HIP(c, hipMalloc((void**)&b->d_dp, sizeof(double) * 6 * W * B));
    VO_HIP(c, hipMalloc((void**)&b->d_evalpart, sizeof(double) * nblk_alloc * BA_EVAL_VALS * B));
    b->red_stride = (size_t)b->n_tiles * 256 + (size_t)W * BA_POSE_VALS + 1 + VO_COMM_MAX_RANKS;
    VO_HIP(c, hipMalloc((void**)&b->d_tilesum, sizeof(double) * b->red_stride * B));
    VO_HIP(c, hipMemsetAsync(b->d_tilesum, 0, sizeof(double) * b->red_stride * B, c->stream));
    b->d_posesum = b->d_tilesum + (size_t)b->n_tiles * 256;
    VO_HIP(c, hipMalloc((void**)&b->d_xstat, sizeof(double) * BA_EVAL_VALS * B));
    VO_HIP(c, hipMalloc((void**)&b->d_cams, sizeof(double) * 2 * W * BA_CAM * B));
    VO_HIP(c, hipMalloc((void**)&b->d_S, sizeof(double) * ((size_t)36 * W * W + 6 * W)));          // probes: problem 0 only
    VO_HIP(c, hipMalloc((void**)&b->d_Hpp, sizeof(double) * (size_t)W * BA_POSE_VALS));
    VO_HIP(c, hipMalloc((void**)&b->d_res, sizeof(double) * (size_t)W * N * B));
    VO_HIP(c, hipMalloc((void**)&b->d_dl, sizeof(double) * (size_t)3 * N));
    b->pub_bytes = VO_BA_PUB_HEADER + sizeof(double) * nx;                                                        // per problem
    VO_HIP(c, hipMalloc((void**)&b->d_pub, b->pub_bytes * B));
    VO_HIP(c, hipHostMalloc((void**)&b->h_pub, 3 * b->pub_bytes * B, hipHostMallocDefault));   // two halves (pipelined frame steps) + one for vo_ba_fetch
    b->d_xout = reinterpret_cast<double*>(b->d_pub + VO_BA_PUB_HEADER);
    VO_HIP(c, hipMalloc((void**)&b->d_state, sizeof(ba_state) * 2 * B));
    VO_HIP(c, hipMalloc((void**)&b->d_info, sizeof(ba_info) * B));
    VO_HIP(c, hipMalloc((void**)&b->d_gdyn, sizeof(int) * 2 * B));
    VO_HIP(c, hipMemsetAsync(b->d_gdyn, 0, sizeof(int) * 2 * B, c->stream));
    VO_HIP(c, hipHostMalloc((void**)&b->h_state, sizeof(ba_state) * B, hipHostMallocDefault));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<256, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build_w<4, 2, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build_w<2, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  }
  ba_geometry(c->ba, W, N, c->tune);
  if (c->batch > 1024) c->ba->v2 = 0;             // (ba2_select_work keeps the running-problem flags of <= 1 024 problems in LDS)
  // one or two problems with BASELINE's window: the lane-per-observation kernels spread a problem over 125 workgroups of 1 024 lanes, the
  // wave-private ones over 42 of 256 -- 3 430 against 3 260 frames/s for ONE sequence, 6 180 against 5 930 for two, even at three, behind from
  // four on (tools/one_sequence_ba_knobs.sh, profiles/r05_small_batch_ba.txt).  Windows of <= 8 slots (the closed loop's 4) are faster
  // wave-private at every batch.  The two families agree to ~1e-12, not bit for bit: like a batch of 32 against its problems one by one.
  if (c->batch <= 2 && W >= 9 && c->tune.ba_kernels == 0) c->ba->v2 = 0;
  VO_CHECK(c, c->ba->build_lds <= 130 * 1024 && c->ba->solve_lds <= 150 * 1024, VO_E_CAPACITY, "window too large for LDS");
  VO_CHECK(c, c->ba->nblk <= 640, VO_E_CAPACITY, "too many landmarks for one adjust (raise the partial capacity)");
  return VO_OK;
}

// A bank of `n_problems` problems of one shape resident in HBM, one of them selected per solve: the bench's sequences see a different
// bundle-adjustment problem every frame (a sliding window never presents the same problem twice) without any upload in the loop.
// K [batch][9]; poses [n_problems][batch][W][6]; points [n_problems][batch][N][3]; obs [n_problems][batch][W][N][2]
extern "C" int32_t vo_ba_upload_bank(vo_ctx* c, const double* K, const double* poses, const double* points, const double* obs,
                                     int32_t n_slots, int32_t n_pts, int32_t n_problems) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, K && poses && points && obs && n_problems >= 1 && n_problems <= 64, VO_E_INVALID, "null buffer / bad bank size");
  VO_HIP(c, hipSetDevice(c->device));
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (c->ba) vo_ba_destroy(c);                       // a fresh workspace: the bank replaces its problem buffers
  int32_t r = ba_alloc(c, n_slots, n_pts);
  if (r != VO_OK) return r;
  vo_ba_ws* b = c->ba;
  const size_t W = b->W, N = b->N, B = c->batch, nx = 6 * W + 3 * N, no = 2 * W * N;
  VO_HIP(c, hipMalloc((void**)&b->d_bank_x0, sizeof(double) * nx * B * n_problems));
  VO_HIP(c, hipMalloc((void**)&b->d_bank_obs, sizeof(double) * no * B * n_problems));
  VO_HIP(c, hipMemcpyAsync(b->d_K, K, 9 * sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpyAsync(b->d_bank_obs, obs, sizeof(double) * no * B * n_problems, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(b->d_bank_x0, sizeof(double) * nx, poses, sizeof(double) * 6 * W, sizeof(double) * 6 * W, B * n_problems, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(b->d_bank_x0 + 6 * W, sizeof(double) * nx, points, sizeof(double) * 3 * N, sizeof(double) * 3 * N, B * n_problems, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  b->d_x0_own = b->d_x0; b->d_obs_own = b->d_obs;
  b->d_x0 = b->d_bank_x0; b->d_obs = b->d_bank_obs;
  b->bank_n = n_problems; b->bank_sel = 0;
  b->d_nlive = nullptr; b->nlive_stride = 0;      // a dense uploaded problem: every slot counts
  b->uploaded = true;
  return VO_OK;
}

// the problem the next vo_ba_solve_resident / vo_frame_step_resident solves (host-side pointer switch; nothing is enqueued)
extern "C" int32_t vo_ba_select_problem(vo_ctx* c, int32_t k) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->ba && c->ba->bank_n > 0, VO_E_STATE, "vo_ba_upload_bank first");
  VO_CHECK(c, k >= 0 && k < c->ba->bank_n, VO_E_INVALID, "no such problem in the bank");
  vo_ba_ws* b = c->ba;
  const size_t W = b->W, N = b->N, B = c->batch;
  b->d_x0 = b->d_bank_x0 + (size_t)k * B * (6 * W + 3 * N);
  b->d_obs = b->d_bank_obs + (size_t)k * B * (2 * W * N);
  b->bank_sel = k;
  return VO_OK;
}

static ba_ptrs ba_make_ptrs(const vo_ctx* c) {
  vo_ba_ws* b = c->ba;
  ba_ptrs P;
  P.K = b->d_K; P.obs = b->d_obs; P.x0 = b->d_x0; P.xa = b->d_x[0]; P.xb = b->d_x[1]; P.aux = b->d_aux;
  P.posepart = b->d_posepart; P.gmax = b->d_gmax; P.tiles = b->d_tiles; P.dp = b->d_dp; P.evalpart = b->d_evalpart;
  P.tilesum = b->d_tilesum; P.posesum = b->d_posesum; P.cams = b->d_cams; P.xstat = b->d_xstat;
  P.state = b->d_state; P.info = b->d_info; P.dbg = nullptr;
  P.W = b->W; P.N = b->N; P.LPP = b->LPP; P.PPB = b->PPB; P.nblk = b->nblk; P.cam_off = b->cam_off; P.RP = b->RP; P.RT = b->RT;
  P.n_tiles = b->n_tiles; P.pitch = b->pitch;
  {
    // landmark chunks per workgroup: as many as keep >= 1024 workgroups (4 per CU, what the kernel's registers allow) in the launch,
    // at most 4 -- a batch of 32 problems x 125 chunks -> 32 workgroups per problem; ONE sequence keeps a workgroup per chunk
    // vo_tuning.ba_chunks (read at every solve, so a test can switch it): chunks per workgroup, 0 = the rule.  NB the partial sums of a
    // workgroup's chunks are added in chunk order, so the last bits of a solution depend on this number -- and through the rule on the
    // batch size: a batch of 32 problems is not bit-identical to the same problems solved one by one (tests/test_gpu_ba.py pins 1e-12)
    const int cpw_env = c->tune.ba_chunks;
    int cpw = cpw_env > 0 ? cpw_env : (int)(((long long)c->batch * b->nblk + 512) / 1024);
    if (cpw < 1) cpw = 1;
    if (cpw > 4 && cpw_env <= 0) cpw = 4;
    P.nset = (cpw > 1 && b->tpb == 256) ? (b->nblk + cpw - 1) / cpw : b->nblk;
  }
  P.n_eval = b->nblk;
  if (b->v2) {
    // G workgroups (4 waves each) per problem: enough for BA2_TARGET_WAVES waves in the launch, at most one chunk (8 landmarks) per wave.
    // vo_tuning.ba_workgroups overrides.  The partial sums depend on G like the older kernels' on the chunks per workgroup.
    const int nchunk = vo_div_up(b->N, b->v2_lpp == 5 ? 12 : 64 / b->v2_lpp), gmax = vo_div_up(nchunk, 4);
    int G = vo_div_up(BA2_TARGET_WAVES, 4 * c->batch);
    if (c->tune.ba_workgroups > 0) G = c->tune.ba_workgroups;
    if (G > gmax) G = gmax;
    // small windows, whose partial sets k_ba_solve sums itself (one value per thread): more than 16 sets make that loop the longest kernel
    // of an iteration (ONE sequence, window 4: 32 sets 3 490, 16 sets 3 580, 8 sets 3 490 frames/s through the closed loop)
    { const int n1 = 6 * b->W + 1; if ((n1 * (n1 + 1)) / 2 + b->W * BA_POSE_VALS <= BA_SOLVE_THREADS && G > 16 && c->tune.ba_workgroups <= 0) G = 16; }
    if (G < 1) G = 1;
    P.nset = G; P.n_eval = G;
    // once problems of the batch have finished, a running one is given up to 16 workgroups (ba2_select_work)
    int cap = 16;
    if (c->tune.ba_workgroup_cap > 0) cap = c->tune.ba_workgroup_cap;
    if (cap > gmax) cap = gmax;
    if (cap < G) cap = G;
    b->v2_g0 = G; b->v2_gcap = cap;
  }
  P.gdyn = b->v2 ? b->d_gdyn : nullptr;
  // strides use the ALLOCATED capacity for N-dependent buffers? no: they are packed for the current problem size
  const size_t W = (size_t)b->W, N = (size_t)b->N;
  P.s_obs = 2 * W * N; P.s_x = 6 * W + 3 * N; P.s_aux = N * BA_AUX; P.s_posepart = (size_t)b->nblk * W * BA_POSE_VALS;
  P.s_gmax = (size_t)b->nblk; P.s_tiles = (size_t)b->nblk * b->n_tiles * 256; P.s_dp = 6 * W;
  P.s_evalpart = (size_t)b->nblk * BA_EVAL_VALS; P.s_tilesum = b->red_stride; P.s_posesum = b->red_stride;
  P.s_cams = 2 * W * BA_CAM;
  P.sharded = c->ba_sharded; P.rank = c->comm_rank; P.n_ranks = c->comm_ranks; P.batch = c->batch;
  P.n_live = b->d_nlive; P.s_nlive = b->nlive_stride;
  {
    // the solve sums the partial sets itself when that is ONE trip per thread (lower triangle of [S rhs] + camera sums <= 1 024 values: windows
    // of <= 6 slots; <= 32 sets) and no exchange between shards needs the reduced packet; vo_tuning.ba_fold = 1 / 2 overrides (A/B).  Measured
    // interleaved on one box (closed loop, window 4): 32 sequences (32 sets) 33 800 -> 35 000 frames/s, 96 sequences (16 sets) 42 200 ->
    // 42 500; ONE sequence (64 sets) 3 134 -> 3 077: worse, and at window 10 (2 171 values from 32 sets, three trips) the headline loses 4 %
    // and one sequence with 125 sets halves -- hence the rule
    const int n1 = 6 * b->W + 1;
    const bool small = ((n1 * (n1 + 1)) / 2 + b->W * BA_POSE_VALS <= BA_SOLVE_THREADS && P.nset <= 32) || (b->v2 && P.nset <= 4);
    // (wave-private kernels: the rule looks at the sets of a full launch; a tail launch hands a running problem up to v2_gcap of them)
    P.fold = (!c->ba_sharded && (c->tune.ba_fold ? c->tune.ba_fold == 2 : small)) ? 1 : 0;
  }
  return P;
}

// closed loop: counts[b * stride] landmark slots of problem b are in use (device memory, read by every launch; null: all N)
void vo_ba_set_live(vo_ctx* c, const int32_t* d_counts, int stride) {
  if (!c->ba) return;
  c->ba->d_nlive = d_counts; c->ba->nlive_stride = stride;
}
static ba_ptrs ba_make_ptrs_dbg(vo_ctx* c) { ba_ptrs P = ba_make_ptrs(c); P.dbg = c->d_dbg; return P; }

static ba_params_dev ba_dev_params(const vo_ba_params* p) {
  ba_params_dev d;
  d.ftol = p->ftol; d.xtol = p->xtol; d.gtol = p->gtol; d.lambda0 = p->lambda0; d.delta = p->huber_delta;
  d.lambda_min = (p->lambda_min > 1e-12) ? p->lambda_min : 1e-12;
  d.max_iters = p->max_iters;
  return d;
}

// K [batch][9], poses [batch][W][6], points [batch][N][3], obs [batch][W][N][2]
extern "C" int32_t vo_ba_upload(vo_ctx* c, const double* K, const double* poses, const double* points, const double* obs,
                                int32_t n_slots, int32_t n_pts) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, K && poses && points && obs, VO_E_INVALID, "null buffer");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));     // a pipelined frame step runs its bundle adjustment there
  if (c->ba && c->ba->bank_n > 0) { VO_HIP(c, hipStreamSynchronize(c->stream)); vo_ba_destroy(c); }      // a plain upload replaces a bank
  int32_t r = ba_alloc(c, n_slots, n_pts);
  if (r != VO_OK) return r;
  vo_ba_ws* b = c->ba;
  const size_t W = b->W, N = b->N, B = c->batch, nx = 6 * W + 3 * N;
  VO_HIP(c, hipMemcpyAsync(b->d_K, K, 9 * sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpyAsync(b->d_obs, obs, sizeof(double) * 2 * W * N * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(b->d_x0, sizeof(double) * nx, poses, sizeof(double) * 6 * W, sizeof(double) * 6 * W, B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(b->d_x0 + 6 * W, sizeof(double) * nx, points, sizeof(double) * 3 * N, sizeof(double) * 3 * N, B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  b->d_nlive = nullptr; b->nlive_stride = 0;      // a dense uploaded problem: every slot counts
  b->uploaded = true;
  return VO_OK;
}

static int32_t ba_launch_iter(vo_ctx* c, const ba_ptrs& P, const ba_params_dev& prm, int it, double probe_lambda,
                              double* probe_S, double* hpp_out, double* probe_dl) {
  vo_ba_ws* b = c->ba;
  const int B = c->batch;
  if (b->v2) {
    const dim3 g(b->v2_g0 * B);
    const int g0 = b->v2_g0, gc = b->v2_gcap;
    if (b->v2_spl == 2 && b->v2_lpp == 5) hipLaunchKernelGGL((k_ba_build_w<4, 2, 5>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else if (b->v2_spl == 2) hipLaunchKernelGGL((k_ba_build_w<4, 2, 8>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else if (b->v2_rt == 4) hipLaunchKernelGGL((k_ba_build_w<4, 1, 8>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else if (b->v2_rt == 3) hipLaunchKernelGGL((k_ba_build_w<3, 1, 8>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else if (b->v2_rt == 2 && b->v2_lpp == 4) hipLaunchKernelGGL((k_ba_build_w<2, 1, 4>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else if (b->v2_rt == 2) hipLaunchKernelGGL((k_ba_build_w<2, 1, 8>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else if (b->v2_lpp == 4) hipLaunchKernelGGL((k_ba_build_w<1, 1, 4>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
    else hipLaunchKernelGGL((k_ba_build_w<1, 1, 8>), g, dim3(256), b->v2_lds, c->stream, P, prm, it, probe_lambda, g0, gc);
  } else if (b->tpb == 256 && b->LPP == 8) hipLaunchKernelGGL((k_ba_build<256, 8>), dim3(P.nset, B), dim3(256), b->build_lds, c->stream, P, prm, it, probe_lambda);
  else if (b->tpb == 256) hipLaunchKernelGGL(k_ba_build<256>, dim3(P.nset, B), dim3(256), b->build_lds, c->stream, P, prm, it, probe_lambda);
  else if (b->tpb == 512) hipLaunchKernelGGL(k_ba_build<512>, dim3(b->nblk, B), dim3(512), b->build_lds, c->stream, P, prm, it, probe_lambda);
  else hipLaunchKernelGGL(k_ba_build<1024>, dim3(b->nblk, B), dim3(1024), b->build_lds, c->stream, P, prm, it, probe_lambda);
  if (!P.fold) hipLaunchKernelGGL(k_ba_reduce, dim3(vo_div_up(b->n_tiles * 256 + b->W * BA_POSE_VALS + 1, 64), B), dim3(256), 0, c->stream, P, it);
  if (P.sharded) {
    // exchange 1: every shard's reduced packet -> the sum over all shards of all ranks, in every entry
    hipLaunchKernelGGL(k_ba_xsum, dim3(vo_div_up((int)b->red_stride, 256)), dim3(256), 0, c->stream, P, it);
    const int32_t r = vo_comm_allreduce_f64(c, b->d_tilesum, b->red_stride * B);
    if (r != VO_OK) return r;
  }
  hipLaunchKernelGGL(k_ba_solve, dim3(B), dim3(BA_SOLVE_THREADS), b->solve_lds, c->stream, P, prm, it, probe_S, hpp_out);
  if (b->v2) {
    if (b->v2_spl == 2 && b->v2_lpp == 5) hipLaunchKernelGGL((k_ba_update_w<2, 5>), dim3(b->v2_g0 * B), dim3(256), 0, c->stream, P, prm, it, probe_dl, b->v2_g0, b->v2_gcap);
    else if (b->v2_spl == 2) hipLaunchKernelGGL((k_ba_update_w<2, 8>), dim3(b->v2_g0 * B), dim3(256), 0, c->stream, P, prm, it, probe_dl, b->v2_g0, b->v2_gcap);
    else if (b->v2_lpp == 4) hipLaunchKernelGGL((k_ba_update_w<1, 4>), dim3(b->v2_g0 * B), dim3(256), 0, c->stream, P, prm, it, probe_dl, b->v2_g0, b->v2_gcap);
    else hipLaunchKernelGGL((k_ba_update_w<1, 8>), dim3(b->v2_g0 * B), dim3(256), 0, c->stream, P, prm, it, probe_dl, b->v2_g0, b->v2_gcap);
  } else if (b->tpb == 256 && b->LPP == 8) hipLaunchKernelGGL((k_ba_update<256, 8>), dim3(b->nblk, B), dim3(256), 0, c->stream, P, prm, it, probe_dl);
  else if (b->tpb == 256) hipLaunchKernelGGL(k_ba_update<256>, dim3(b->nblk, B), dim3(256), 0, c->stream, P, prm, it, probe_dl);
  else if (b->tpb == 512) hipLaunchKernelGGL(k_ba_update<512>, dim3(b->nblk, B), dim3(512), 0, c->stream, P, prm, it, probe_dl);
  else hipLaunchKernelGGL(k_ba_update<1024>, dim3(b->nblk, B), dim3(1024), 0, c->stream, P, prm, it, probe_dl);
  if (P.sharded) {
    // exchange 2: the 4 step statistics the next decision needs
    hipLaunchKernelGGL(k_ba_xstat, dim3(1), dim3(256), 0, c->stream, P, it);
    const int32_t r = vo_comm_allreduce_f64(c, b->d_xstat, (size_t)BA_EVAL_VALS * B);
    if (r != VO_OK) return r;
  }
  return VO_OK;
}

// enqueue `n_it` LM iterations starting at iteration index `it0`
static int32_t ba_enqueue_iters(vo_ctx* c, const ba_params_dev& prm, int it0, int n_it) {
  vo_prof_scope prof(c, VO_PROF_BA);
  const ba_ptrs P = ba_make_ptrs(c);
  for (int it = it0; it < it0 + n_it; it++) {
    const int32_t r = ba_launch_iter(c, P, prm, it, -1.0, nullptr, nullptr, nullptr);
    if (r != VO_OK) return r;
    // pipelined frame step: the groups in which (nearly) every problem of the batch still runs are over -- the next frame's tracker may start
    if (c->ba_wide_event && it - it0 + 1 == c->ba_wide_groups) VO_HIP(c, hipEventRecord(c->ba_wide_event, c->stream));
  }
  if (c->ba_wide_event && n_it < c->ba_wide_groups) VO_HIP(c, hipEventRecord(c->ba_wide_event, c->stream));
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

static void ba_launch_finalize(vo_ctx* c, const ba_ptrs& P, const ba_params_dev& d, int n_it, ba_state* st_out, int st_stride) {
  vo_ba_ws* b = c->ba;
  hipLaunchKernelGGL(k_ba_finalize, dim3(c->batch, 8), dim3(256), 0, c->stream, P, d, n_it, b->d_pub, b->pub_bytes, st_out, st_stride);
}

extern "C" int32_t vo_ba_solve_resident(vo_ctx* c, const vo_ba_params* prm) {
  if (!c) return VO_E_INVALID;
  vo_ba_params def;
  if (!prm) { vo_ba_default_params(&def); prm = &def; }
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "vo_ba_upload first");
  VO_CHECK(c, prm->max_iters >= 0 && prm->max_iters <= 1000, VO_E_INVALID, "bad max_iters");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  const ba_params_dev d = ba_dev_params(prm);
  int32_t r = ba_enqueue_iters(c, d, 0, prm->max_iters);
  if (r != VO_OK) return r;
  vo_ba_ws* b = c->ba;
  const ba_ptrs P = ba_make_ptrs(c);
  // pipelined frame step: the copy of the PREVIOUS solution runs on another stream; it must have left d_pub before this one lands there
  if (c->ba_wait_before_publish) VO_HIP(c, hipStreamWaitEvent(c->stream, c->ba_wait_before_publish, 0));
  ba_launch_finalize(c, P, d, prm->max_iters, b->d_state + (prm->max_iters & 1), 2);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

// ---- closed-loop pipeline hooks: the problem (x0, obs) is written by a device kernel every frame ----
int32_t vo_ba_reserve(vo_ctx* c, const double* K_host, int W, int N) {
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
  if (c->ba && c->ba->bank_n > 0) { VO_HIP(c, hipStreamSynchronize(c->stream)); vo_ba_destroy(c); }
  int32_t r = ba_alloc(c, W, N);
  if (r != VO_OK) return r;
  vo_ba_ws* b = c->ba;
  const size_t B = c->batch;
  VO_HIP(c, hipMemcpyAsync(b->d_K, K_host, 9 * sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemsetAsync(b->d_x0, 0, sizeof(double) * ((size_t)6 * W + 3 * (size_t)N) * B, c->stream));
  VO_HIP(c, hipMemsetAsync(b->d_obs, 0xFF, sizeof(double) * 2 * (size_t)W * N * B, c->stream));       // all NaN: nothing observed
  VO_HIP(c, hipStreamSynchronize(c->stream));
  b->uploaded = true;
  return VO_OK;
}
int32_t vo_ba_get_view(vo_ctx* c, vo_ba_view* v) {
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "vo_ba_reserve first");
  vo_ba_ws* b = c->ba;
  v->x0 = b->d_x0; v->obs = b->d_obs; v->pub = b->d_pub; v->pub_bytes = b->pub_bytes;
  v->x_stride = (size_t)6 * b->W + 3 * (size_t)b->N; v->obs_stride = (size_t)2 * b->W * b->N; v->W = b->W; v->N = b->N;
  return VO_OK;
}
int32_t vo_ba_enqueue_budget(vo_ctx* c, const vo_ba_params* prm, int it0, int n_it) {
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "vo_ba_reserve first");
  const ba_params_dev d = ba_dev_params(prm);
  if (n_it > 0) { const int32_t r = ba_enqueue_iters(c, d, it0, n_it); if (r != VO_OK) return r; }
  const ba_ptrs P = ba_make_ptrs(c);
  ba_launch_finalize(c, P, d, it0 + n_it, c->ba->d_state + ((it0 + n_it) & 1), 2);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

static void ba_fill_stats(const ba_state& s, int n_obs, vo_ba_stats* st) {
  st->cost0 = s.cost0; st->cost = s.cost; st->lambda = s.lambda; st->iters = s.iter; st->accepted = s.accepted;
  st->status = s.status; st->n_obs = n_obs;
}

// internal: enqueue the D2H copy of the published results into the pinned mirror (used by the frame step)
int32_t vo_ba_enqueue_pub_copy(vo_ctx* c, int half) {
  vo_ba_ws* b = c->ba;
  VO_HIP(c, hipMemcpyAsync(b->h_pub + (size_t)half * b->pub_bytes * c->batch, b->d_pub, b->pub_bytes * c->batch, hipMemcpyDeviceToHost, c->stream));
  return VO_OK;
}

// internal: unpack the pinned mirror (after a stream sync): poses_out [batch][W][6], points_out [batch][N][3], stats [batch]
void vo_ba_unpack_pub(vo_ctx* c, int half, double* poses_out, double* points_out, vo_ba_stats* stats) {
  vo_ba_ws* b = c->ba;
  for (int q = 0; q < c->batch; q++) {
    const uint8_t* pub = b->h_pub + (size_t)half * b->pub_bytes * c->batch + (size_t)q * b->pub_bytes;
    const double* x = reinterpret_cast<const double*>(pub + VO_BA_PUB_HEADER);
    if (poses_out) memcpy(poses_out + (size_t)q * 6 * b->W, x, sizeof(double) * 6 * b->W);
    if (points_out) memcpy(points_out + (size_t)q * 3 * b->N, x + 6 * b->W, sizeof(double) * 3 * b->N);
    if (stats) ba_fill_stats(*reinterpret_cast<const ba_state*>(pub), -1, stats + q);
  }
}

extern "C" int32_t vo_ba_fetch(vo_ctx* c, double* poses_out, double* points_out, vo_ba_stats* stats) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "nothing to fetch");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  // its own pinned mirror: a frame step still in flight keeps the two halves vo_frame_fetch reads
  int32_t r = vo_ba_enqueue_pub_copy(c, 2);
  if (r != VO_OK) return r;
  VO_HIP(c, hipStreamSynchronize(c->stream));
  vo_ba_unpack_pub(c, 2, poses_out, points_out, stats);
  return VO_OK;
}

extern "C" int32_t vo_ba_set_sharded(vo_ctx* c, int32_t on) {
  if (!c) return VO_E_INVALID;
  c->ba_sharded = on ? 1 : 0;
  return VO_OK;
}

// points of every shard of every rank after a solve: points_all [n_ranks][batch][N][3]
extern "C" int32_t vo_ba_gather_points(vo_ctx* c, double* points_all) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, points_all, VO_E_INVALID, "null output");
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "nothing to gather");
  VO_HIP(c, hipSetDevice(c->device));
  vo_ba_ws* b = c->ba;
  const size_t B = c->batch, R = c->comm_ranks, cnt = B * 3 * (size_t)b->N;
  if (b->gather_cap < cnt * (1 + R)) {
    if (b->d_gather) (void)hipFree(b->d_gather);
    if (b->h_gather) (void)hipHostFree(b->h_gather);
    b->d_gather = nullptr; b->h_gather = nullptr; b->gather_cap = 0;
    VO_HIP(c, hipMalloc((void**)&b->d_gather, sizeof(double) * cnt * (1 + R)));
    VO_HIP(c, hipHostMalloc((void**)&b->h_gather, sizeof(double) * cnt * R, hipHostMallocDefault));
    b->gather_cap = cnt * (1 + R);
  }
  double* send = b->d_gather; double* recv = b->d_gather + cnt;
  // points part of the published x of every problem -> packed send buffer
  VO_HIP(c, hipMemcpy2DAsync(send, sizeof(double) * 3 * b->N, b->d_pub + VO_BA_PUB_HEADER + sizeof(double) * 6 * b->W, b->pub_bytes,
                             sizeof(double) * 3 * b->N, B, hipMemcpyDeviceToDevice, c->stream));
  const int32_t r = vo_comm_allgather_f64(c, send, recv, cnt);
  if (r != VO_OK) return r;
  VO_HIP(c, hipMemcpyAsync(b->h_gather, recv, sizeof(double) * cnt * R, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(points_all, b->h_gather, sizeof(double) * cnt * R);
  return VO_OK;
}

// arrays with a leading batch dimension; stats [batch]
extern "C" int32_t vo_ba_adjust(vo_ctx* c, const double* K, const double* poses, const double* points, const double* obs,
                                int32_t n_slots, int32_t n_pts, const vo_ba_params* prm, double* poses_out,
                                double* points_out, vo_ba_stats* stats) {
  if (!c) return VO_E_INVALID;
  vo_ba_params def;
  if (!prm) { vo_ba_default_params(&def); prm = &def; }
  VO_CHECK(c, poses_out && points_out, VO_E_INVALID, "null output");
  VO_CHECK(c, prm->max_iters >= 0 && prm->max_iters <= 1000, VO_E_INVALID, "bad max_iters");
  int32_t r = vo_ba_upload(c, K, poses, points, obs, n_slots, n_pts);
  if (r != VO_OK) return r;
  vo_ba_ws* b = c->ba;
  const ba_params_dev d = ba_dev_params(prm);
  const ba_ptrs P = ba_make_ptrs(c);
  const int B = c->batch;
  // iterations are enqueued in chunks; between chunks the host peeks at the states to stop early
  const int CH = 4;
  int it = 0;
  bool all_done = false;
  do {
    const int n = (prm->max_iters - it < CH) ? prm->max_iters - it : CH;
    if (n > 0) {
      r = ba_enqueue_iters(c, d, it, n);
      if (r != VO_OK) return r;
      it += n;
    }
    ba_launch_finalize(c, P, d, it, b->d_state + (it & 1), 2);
    VO_HIP(c, hipMemcpy2DAsync(b->h_state, sizeof(ba_state), b->d_state + (it & 1), 2 * sizeof(ba_state), sizeof(ba_state), B,
                               hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    all_done = true;
    for (int q = 0; q < B; q++) all_done = all_done && b->h_state[q].done;
  } while (!all_done && it < prm->max_iters);
  r = vo_ba_enqueue_pub_copy(c, 2);
  if (r != VO_OK) return r;
  VO_HIP(c, hipStreamSynchronize(c->stream));
  vo_ba_unpack_pub(c, 2, poses_out, points_out, stats);
  bool bad = false;
  for (int q = 0; q < B; q++) {
    if (stats) {
      int n_obs = 0;
      const double* o = obs + (size_t)q * 2 * b->W * b->N;
      for (size_t k = 0; k < (size_t)b->W * b->N; k++) n_obs += (o[2 * k] == o[2 * k]);
      stats[q].n_obs = n_obs;
    }
    bad = bad || (b->h_state[q].cost != b->h_state[q].cost);
  }
  if (bad) return vo_fail(c, VO_E_NUMERIC, "bundle adjustment produced a non-finite cost");
  return VO_OK;
}

// parity probe of problem 0 of the batch
extern "C" int32_t vo_ba_probe(vo_ctx* c, double lambda, double huber_delta, double* residual, int32_t* n_obs, double* cost,
                               double* Hpp, double* gp, double* Hll, double* gl, double* S, double* rhs, double* dposes,
                               double* dpoints) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "vo_ba_upload first");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  vo_ba_ws* b = c->ba;
  const int W = b->W, N = b->N, n = 6 * W;
  vo_ba_params prm;
  vo_ba_default_params(&prm);
  prm.huber_delta = huber_delta; prm.lambda0 = lambda; prm.max_iters = 1;
  const ba_params_dev d = ba_dev_params(&prm);
  const ba_ptrs P = ba_make_ptrs_dbg(c);
  hipLaunchKernelGGL(k_ba_residual, dim3(vo_div_up(N, 128), c->batch), dim3(128), 0, c->stream, P, b->d_x0, huber_delta, b->d_res);
  { const int32_t r = ba_launch_iter(c, P, d, 0, lambda, b->d_S, b->d_Hpp, b->d_dl); if (r != VO_OK) return r; }
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipStreamSynchronize(c->stream));
  // ---- copy out ----
  const size_t total = (size_t)W * N + (size_t)N * BA_AUX + (size_t)W * BA_POSE_VALS + (size_t)n * n + n + n + 3 * (size_t)N;
  double* h = (double*)malloc(sizeof(double) * total);
  if (!h) return vo_fail(c, VO_E_NOMEM, "probe host buffer");
  double* h_res = h; double* h_aux = h_res + (size_t)W * N; double* h_hpp = h_aux + (size_t)N * BA_AUX;
  double* h_S = h_hpp + (size_t)W * BA_POSE_VALS; double* h_dp = h_S + (size_t)n * n + n; double* h_dl = h_dp + n;
  hipError_t e = hipMemcpy(h_res, b->d_res, sizeof(double) * (size_t)W * N, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_aux, b->d_aux, sizeof(double) * (size_t)N * BA_AUX, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_hpp, b->d_Hpp, sizeof(double) * (size_t)W * BA_POSE_VALS, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_S, b->d_S, sizeof(double) * ((size_t)n * n + n), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_dp, b->d_dp, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_dl, b->d_dl, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(h); return vo_fail(c, VO_E_HIP, std::string("probe copy: ") + hipGetErrorString(e)); }
  int m = 0;
  double cs = 0;
  for (int i = 0; i < W; i++) {
    for (int j = 0; j < N; j++) {
      const double rv = h_res[(size_t)i * N + j];
      if (rv == rv) { if (residual) residual[m] = rv; m++; }
    }
    cs += h_hpp[i * BA_POSE_VALS + 27];
  }
  if (n_obs) *n_obs = m;
  if (cost) *cost = cs;
  for (int i = 0; i < W; i++) {
    int q = 0;
    for (int a = 0; a < 6; a++)
      for (int cidx = a; cidx < 6; cidx++) {
        const double v = h_hpp[i * BA_POSE_VALS + q++];
        if (Hpp) { Hpp[i * 36 + a * 6 + cidx] = v; Hpp[i * 36 + cidx * 6 + a] = v; }
      }
    if (gp) for (int a = 0; a < 6; a++) gp[i * 6 + a] = h_hpp[i * BA_POSE_VALS + 21 + a];
  }
  for (int j = 0; j < N; j++) {
    const double* ax = h_aux + (size_t)j * BA_AUX;
    if (Hll) {
      double* H = Hll + (size_t)j * 9;
      H[0] = ax[0]; H[1] = ax[1]; H[2] = ax[3]; H[3] = ax[1]; H[4] = ax[2]; H[5] = ax[4]; H[6] = ax[3]; H[7] = ax[4]; H[8] = ax[5];
    }
    if (gl) { gl[3 * j] = ax[6]; gl[3 * j + 1] = ax[7]; gl[3 * j + 2] = ax[8]; }
  }
  if (S) for (size_t k = 0; k < (size_t)n * n; k++) S[k] = h_S[k];
  if (rhs) for (int a = 0; a < n; a++) rhs[a] = h_S[(size_t)n * n + a];
  if (dposes) for (int a = 0; a < n; a++) dposes[a] = h_dp[a];
  if (dpoints) for (size_t k = 0; k < 3 * (size_t)N; k++) dpoints[k] = h_dl[k];
  free(h);
  return VO_OK;
}
