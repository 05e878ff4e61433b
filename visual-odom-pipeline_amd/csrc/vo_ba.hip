// Sliding-window bundle adjustment on gfx950: analytic Jacobians, IRLS-weighted J^T J, landmark Schur
// complement as an f64-MFMA SYRK, in-LDS Cholesky, Levenberg-Marquardt with the accept/reject logic
// on the device (no host round trip inside the solve).
//
// Replaces the scipy.optimize.least_squares(...) call of BundleAdjuster.adjust,
// /root/reference/src/bundle_adjuster/bundle_adjuster.py:189-194, and its objective :18-65 / :68-83.
// Same cost as the reference (Huber, f_scale 1, on the per-observation pixel-error norm; no gauge fixing);
// the solver is the one `north_star` asks for and oracle/ba_oracle.py defines (and the tests compare
// against step by step): residual e (2), blocks J_p (2x6), J_l (2x3), w = rho'(|e|^2),
//   H = sum w J^T J, g = sum w J^T e, (H + lambda diag H) d = -g via Schur on the 3x3 landmark blocks.
//
// Data in HBM (float64):
//   obs   [W][N][2]   slot-major (slot 0 = newest frame), NaN = not observed  -> coalesced over landmarks
//   x[2]  {poses [W][6], points [N][3]}  current / trial, selected by state.cur
//   Yt    [3N (+pad)][RP]   RP = roundup(6W+1, 16): row 3j+c holds column c of Y_j = H_pl,j L_j for every
//         pose parameter, and y_j[c] = (L_j^T g_l,j)[c] in column 6W, so that ONE SYRK  Yt^T Yt  yields both
//         E = sum_j H_pl M_j H_pl^T and r = sum_j H_pl M_j g_l  (M_j = (H_ll,j + lambda D_j)^-1 = L_j L_j^T).
//
// One LM iteration = 4 launches on the ctx stream (state is double buffered by iteration parity):
//   k_ba_linearize : [decide previous step] ; landmark role: thread per landmark (H_ll, g_l, L_j, Yt rows);
//                    camera role: block per (slot, landmark chunk) -> partial H_pp / g_p / cost
//   k_ba_syrk      : v_mfma_f64_16x16x4_f64 over K-slices of Yt, upper 16x16 tiles -> partial tiles
//   k_ba_solve     : one workgroup: reduce partials, assemble the damped reduced camera system, left-looking
//                    Cholesky of [S rhs] in LDS, back substitution -> d_poses
//   k_ba_update    : thread per landmark: back-substitute d_point, trial x, trial cost, step statistics
// k_ba_finalize applies the last decision and publishes x / stats.
#include "vo_internal.h"

#include <math.h>

#define BA_LIN_THREADS 128
#define BA_AUX 18            // per landmark: Hll(6) gl(3) Cinv(6) z(3)
#define BA_POSE_VALS 28      // Hpp upper (21) + gp (6) + cost (1)
#define BA_MAX_SLOTS 20
#define BA_SOLVE_THREADS 512
#define BA_EVAL_VALS 4

typedef double d4 __attribute__((ext_vector_type(4)));

struct ba_state {
  double lambda, nu, cost, cost0;
  int cur, iter, accepted, status, done, n_obs;
};

struct ba_info {            // written by k_ba_solve for the iteration
  double cost_cur, pred_pose, step2_pose, x2_pose, ginf;
  int chol_fail, pad;
};

struct ba_params_dev {
  double ftol, xtol, gtol, lambda0, delta;
  int max_iters;
};

struct vo_ba_ws {
  int W = 0, N = 0, RP = 0, RT = 0, n_tiles = 0, KS = 0, KL = 0, K4 = 0, n_chunk = 0, n_pblk = 0, n_eblk = 0;
  int cap_W = 0, cap_N = 0;
  double* d_K = nullptr;        // 9
  double* d_obs = nullptr;      // W*N*2
  double* d_x0 = nullptr;       // W*6 + N*3
  double* d_x[2] = {nullptr, nullptr};
  double* d_Yt = nullptr;       // K4 * RP
  double* d_aux = nullptr;      // N * BA_AUX
  double* d_posepart = nullptr; // W * n_chunk * 28
  double* d_gmax = nullptr;     // n_pblk
  double* d_tiles = nullptr;    // KS * n_tiles * 256
  double* d_dp = nullptr;       // 6W
  double* d_evalpart = nullptr; // n_eblk * 4
  double* d_S = nullptr;        // probe: (6W)^2 + 6W
  double* d_Hpp = nullptr;      // W*36 + W*6 (probe / reduced values)
  double* d_res = nullptr;      // probe residual W*N
  double* d_xout = nullptr;     // published solution 6W + 3N
  ba_state* d_state = nullptr;  // [2]
  ba_info* d_info = nullptr;
  ba_state* h_state = nullptr;  // pinned
  bool uploaded = false;
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_rodrigues(const double* r, double* R) {
  const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < 2.220446049250313e-16) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    return;
  }
  const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th;
  const double c = cos(th), s = sin(th), c1 = 1.0 - c;
  R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
  R[3] = c1 * ky * kx + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
  R[6] = c1 * kz * kx - s * ky; R[7] = c1 * kz * ky + s * kx; R[8] = c + c1 * kz * kz;
}

// right Jacobian of SO(3): R(r + dr) ~ R(r) Exp(Jr dr)
__device__ __forceinline__ void d_right_jacobian(const double* r, double* J) {
  const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double a, b;
  if (th2 < 1e-8) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
  else { const double th = sqrt(th2); a = (1.0 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
  const double x = r[0], y = r[1], z = r[2];
  // S = [r]x ; S^2 = r r^T - th2 I
  J[0] = 1.0 + b * (x * x - th2); J[1] = a * z + b * x * y;       J[2] = -a * y + b * x * z;
  J[3] = -a * z + b * y * x;      J[4] = 1.0 + b * (y * y - th2); J[5] = a * x + b * y * z;
  J[6] = a * y + b * z * x;       J[7] = -a * x + b * z * y;      J[8] = 1.0 + b * (z * z - th2);
}

// per-slot camera data staged in LDS: R(9) t(3) Jr(9)
#define BA_CAM 21
__device__ __forceinline__ void stage_cameras(const double* poses, int W, double* cam, int tid, int nthreads) {
  for (int i = tid; i < W; i += nthreads) {
    const double* p = poses + 6 * i;
    double* c = cam + BA_CAM * i;
    d_rodrigues(p, c);
    c[9] = p[3]; c[10] = p[4]; c[11] = p[5];
    d_right_jacobian(p, c + 12);
  }
}

struct ba_obs_lin {
  double e0, e1, w, rho;
  double Jl[2][3];
  double Jp[2][6];
};

// residual + Jacobian blocks of one observation.  returns false if unobserved.
template <bool WANT_JP>
__device__ __forceinline__ bool ba_linearize_obs(const double* __restrict__ K, const double* __restrict__ cam,
                                                 const double X[3], double uo, double vo, double delta, ba_obs_lin& o) {
  if (uo != uo) return false;
  const double* R = cam; const double* t = cam + 9; const double* Jr = cam + 12;
  const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  const double p0 = K[0] * xc + K[1] * yc + K[2] * zc;
  const double p1 = K[3] * xc + K[4] * yc + K[5] * zc;
  const double p2 = K[6] * xc + K[7] * yc + K[8] * zc;
  const double u = p0 / p2, v = p1 / p2;
  o.e0 = u - uo; o.e1 = v - vo;
  const double s = o.e0 * o.e0 + o.e1 * o.e1;
  const double d2 = delta * delta;
  if (s <= d2) { o.w = 1.0; o.rho = s; }
  else { const double rs = sqrt(s); o.w = delta / rs; o.rho = 2.0 * delta * rs - d2; }
  double A[2][3];
#pragma unroll
  for (int c = 0; c < 3; c++) { A[0][c] = (K[c] - u * K[6 + c]) / p2; A[1][c] = (K[3 + c] - v * K[6 + c]) / p2; }
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

// ------------------------------------------------------------------------------------------------
// LM decision (runs redundantly in every thread that needs the new state; inputs are identical)
// ------------------------------------------------------------------------------------------------
__device__ inline void ba_decide(const ba_state& in, const ba_info& info, const double* __restrict__ evalpart, int n_eblk,
                                 const ba_params_dev& prm, ba_state& out) {
  out = in;
  if (in.done) return;
  const double F = info.cost_cur;
  if (in.iter == 0) out.cost0 = F;
  out.cost = F;
  if (info.ginf < prm.gtol) { out.done = 1; out.status = 1; return; }
  double Ft = 0, predp = 0, step2 = 0, x2 = 0;
  for (int b = 0; b < n_eblk; b++) {
    Ft += evalpart[b * BA_EVAL_VALS + 0]; predp += evalpart[b * BA_EVAL_VALS + 1];
    step2 += evalpart[b * BA_EVAL_VALS + 2]; x2 += evalpart[b * BA_EVAL_VALS + 3];
  }
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
    if (lam < 1e-12) lam = 1e-12;
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

// ------------------------------------------------------------------------------------------------
// k_ba_linearize
// ------------------------------------------------------------------------------------------------
struct ba_ptrs {
  const double* K; const double* obs;
  double* x[2];
  double* Yt; double* aux; double* posepart; double* gmax; double* tiles; double* dp; double* evalpart;
  ba_state* state; ba_info* info;
  int W, N, RP, n_chunk, n_pblk, n_eblk, KS, KL, K4, n_tiles, RT;
};

__global__ void __launch_bounds__(BA_LIN_THREADS) k_ba_linearize(ba_ptrs P, ba_params_dev prm, int it, double probe_lambda) {
  __shared__ double s_cam[BA_CAM * BA_MAX_SLOTS];
  __shared__ double s_K[9];
  __shared__ double s_red[BA_POSE_VALS * BA_LIN_THREADS];
  __shared__ ba_state s_st;
  const int tid = threadIdx.x;
  // ---- state for this iteration ----
  if (tid == 0) {
    ba_state st;
    if (it == 0) st = P.state[0];                       // initialised by the host-enqueued memcpy
    else ba_decide(P.state[(it - 1) & 1], *P.info, P.evalpart, P.n_eblk, prm, st);
    if (probe_lambda >= 0) st.lambda = probe_lambda;
    s_st = st;
    if (blockIdx.x == 0 && it > 0) P.state[it & 1] = st;
  }
  __syncthreads();
  const ba_state st = s_st;
  if (st.done) return;
  const int W = P.W, N = P.N;
  const double* poses = P.x[st.cur];
  const double* pts = P.x[st.cur] + 6 * W;
  stage_cameras(poses, W, s_cam, tid, BA_LIN_THREADS);
  if (tid < 9) s_K[tid] = P.K[tid];
  __syncthreads();

  if ((int)blockIdx.x < P.n_pblk) {
    // ================= landmark role: one thread per landmark =================
    const int j = blockIdx.x * BA_LIN_THREADS + tid;
    double gm = 0;
    if (j < N) {
      const double X[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
      double h00 = 0, h10 = 0, h11 = 0, h20 = 0, h21 = 0, h22 = 0, g0 = 0, g1 = 0, g2 = 0;
      for (int i = 0; i < W; i++) {
        const double* ob = P.obs + ((size_t)i * N + j) * 2;
        ba_obs_lin o;
        if (!ba_linearize_obs<false>(s_K, s_cam + BA_CAM * i, X, ob[0], ob[1], prm.delta, o)) continue;
        h00 += o.w * (o.Jl[0][0] * o.Jl[0][0] + o.Jl[1][0] * o.Jl[1][0]);
        h10 += o.w * (o.Jl[0][1] * o.Jl[0][0] + o.Jl[1][1] * o.Jl[1][0]);
        h11 += o.w * (o.Jl[0][1] * o.Jl[0][1] + o.Jl[1][1] * o.Jl[1][1]);
        h20 += o.w * (o.Jl[0][2] * o.Jl[0][0] + o.Jl[1][2] * o.Jl[1][0]);
        h21 += o.w * (o.Jl[0][2] * o.Jl[0][1] + o.Jl[1][2] * o.Jl[1][1]);
        h22 += o.w * (o.Jl[0][2] * o.Jl[0][2] + o.Jl[1][2] * o.Jl[1][2]);
        g0 += o.w * (o.Jl[0][0] * o.e0 + o.Jl[1][0] * o.e1);
        g1 += o.w * (o.Jl[0][1] * o.e0 + o.Jl[1][1] * o.e1);
        g2 += o.w * (o.Jl[0][2] * o.e0 + o.Jl[1][2] * o.e1);
      }
      gm = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
      const double lam = st.lambda;
      const double a00 = h00 + lam * fmax(h00, 1e-12), a11 = h11 + lam * fmax(h11, 1e-12), a22 = h22 + lam * fmax(h22, 1e-12);
      // Cholesky of the damped 3x3 block and its inverse (lower triangular)
      const double c00 = sqrt(a00), c10 = h10 / c00, c20 = h20 / c00;
      const double c11 = sqrt(a11 - c10 * c10), c21 = (h21 - c20 * c10) / c11;
      const double c22 = sqrt(a22 - c20 * c20 - c21 * c21);
      const double i00 = 1.0 / c00, i11 = 1.0 / c11, i22 = 1.0 / c22;
      const double i10 = -c10 * i00 * i11;
      const double i21 = -c21 * i11 * i22;
      const double i20 = -(c20 * i00 + c21 * i10) * i22;
      const double y0 = i00 * g0, y1 = i10 * g0 + i11 * g1, y2 = i20 * g0 + i21 * g1 + i22 * g2;
      const double z0 = i00 * y0 + i10 * y1 + i20 * y2, z1 = i11 * y1 + i21 * y2, z2 = i22 * y2;
      double* ax = P.aux + (size_t)j * BA_AUX;
      ax[0] = h00; ax[1] = h10; ax[2] = h11; ax[3] = h20; ax[4] = h21; ax[5] = h22;
      ax[6] = g0; ax[7] = g1; ax[8] = g2;
      ax[9] = i00; ax[10] = i10; ax[11] = i11; ax[12] = i20; ax[13] = i21; ax[14] = i22;
      ax[15] = z0; ax[16] = z1; ax[17] = z2;
      double* y_rows = P.Yt + (size_t)(3 * j) * P.RP;
      y_rows[6 * W] = y0; y_rows[P.RP + 6 * W] = y1; y_rows[2 * P.RP + 6 * W] = y2;
      for (int i = 0; i < W; i++) {
        const double* ob = P.obs + ((size_t)i * N + j) * 2;
        ba_obs_lin o;
        double* d0 = y_rows + 6 * i;
        if (!ba_linearize_obs<true>(s_K, s_cam + BA_CAM * i, X, ob[0], ob[1], prm.delta, o)) {
#pragma unroll
          for (int a = 0; a < 6; a++) { d0[a] = 0; d0[P.RP + a] = 0; d0[2 * P.RP + a] = 0; }
          continue;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const double b0 = o.w * (o.Jp[0][a] * o.Jl[0][0] + o.Jp[1][a] * o.Jl[1][0]);
          const double b1 = o.w * (o.Jp[0][a] * o.Jl[0][1] + o.Jp[1][a] * o.Jl[1][1]);
          const double b2 = o.w * (o.Jp[0][a] * o.Jl[0][2] + o.Jp[1][a] * o.Jl[1][2]);
          d0[a] = b0 * i00;                                   // Y[a][0] = B[a][0] Cinv[0][0]
          d0[P.RP + a] = b0 * i10 + b1 * i11;                 // Y[a][1]
          d0[2 * P.RP + a] = b0 * i20 + b1 * i21 + b2 * i22;  // Y[a][2]
        }
      }
    }
    // block max of |g_l| (deterministic tree)
    s_red[tid] = gm;
    __syncthreads();
    for (int o = BA_LIN_THREADS / 2; o > 0; o >>= 1) {
      if (tid < o) s_red[tid] = fmax(s_red[tid], s_red[tid + o]);
      __syncthreads();
    }
    if (tid == 0) P.gmax[blockIdx.x] = s_red[0];
  } else {
    // ================= camera role: block = (slot, landmark chunk) =================
    const int b = blockIdx.x - P.n_pblk;
    const int slot = b / P.n_chunk, chunk = b - slot * P.n_chunk;
    const int j = chunk * BA_LIN_THREADS + tid;
    double v[BA_POSE_VALS];
#pragma unroll
    for (int q = 0; q < BA_POSE_VALS; q++) v[q] = 0;
    if (j < N) {
      const double X[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
      const double* ob = P.obs + ((size_t)slot * N + j) * 2;
      ba_obs_lin o;
      if (ba_linearize_obs<true>(s_K, s_cam + BA_CAM * slot, X, ob[0], ob[1], prm.delta, o)) {
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = a; c < 6; c++) v[q++] = o.w * (o.Jp[0][a] * o.Jp[0][c] + o.Jp[1][a] * o.Jp[1][c]);
#pragma unroll
        for (int a = 0; a < 6; a++) v[21 + a] = o.w * (o.Jp[0][a] * o.e0 + o.Jp[1][a] * o.e1);
        v[27] = 0.5 * o.rho;
      }
    }
#pragma unroll
    for (int q = 0; q < BA_POSE_VALS; q++) s_red[q * BA_LIN_THREADS + tid] = v[q];
    __syncthreads();
    // fixed-order reduction: 4 lanes per value sum 32 entries each, then combine
    if (tid < BA_POSE_VALS * 4) {
      const int q = tid >> 2, part = tid & 3;
      double s = 0;
      const double* src = s_red + q * BA_LIN_THREADS + part * (BA_LIN_THREADS / 4);
      for (int k = 0; k < BA_LIN_THREADS / 4; k++) s += src[k];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0) P.posepart[((size_t)slot * P.n_chunk + chunk) * BA_POSE_VALS + q] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_ba_syrk : partial tiles of Yt^T Yt with v_mfma_f64_16x16x4_f64
//   grid (KS, ceil(n_tiles/4)), block 256 = 4 waves, one upper tile per wave
//   A[i][k] = Yt[k0+k][16 ta + i]  (lane: i = l&15, k = l>>4),  B[k][j] = Yt[k0+k][16 tb + j]
//   D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ba_syrk(ba_ptrs P, int it) {
  const ba_state st = P.state[it & 1];
  if (st.done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= P.n_tiles) return;
  // tile index -> (ta <= tb)
  int ta = 0, rem = tile;
  while (rem >= P.RT - ta) { rem -= P.RT - ta; ta++; }
  const int tb = ta + rem;
  const int k_begin = blockIdx.x * P.KL;
  int k_end = k_begin + P.KL;
  if (k_end > P.K4) k_end = P.K4;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  const double* base = P.Yt + (size_t)(lane >> 4) * P.RP + (lane & 15);
  for (int k0 = k_begin; k0 < k_end; k0 += 4) {
    const double a = base[(size_t)k0 * P.RP + 16 * ta];
    const double b = base[(size_t)k0 * P.RP + 16 * tb];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  double* out = P.tiles + ((size_t)blockIdx.x * P.n_tiles + tile) * 256 + lane * 4;
  out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; out[3] = acc[3];
}

// ------------------------------------------------------------------------------------------------
// k_ba_solve : one workgroup
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ba_tile_elem(const double* __restrict__ tiles, int KS, int n_tiles, int RT, int a, int b) {
  // element (a, b) of Yt^T Yt, any order of a, b (symmetric); sums the K-slice partials in fixed order
  if (a > b) { const int t = a; a = b; b = t; }
  const int ta = a >> 4, tb = b >> 4, ii = a & 15, jj = b & 15;
  const int tile = ta * RT - (ta * (ta - 1)) / 2 + (tb - ta);
  const int lane = (ii & 3) * 16 + jj, reg = ii >> 2;
  const double* p = tiles + (size_t)tile * 256 + lane * 4 + reg;
  double s = 0;
  for (int ks = 0; ks < KS; ks++) s += p[(size_t)ks * n_tiles * 256];
  return s;
}

__global__ void __launch_bounds__(BA_SOLVE_THREADS) k_ba_solve(ba_ptrs P, ba_params_dev prm, int it, double* __restrict__ probe_S,
                                                               double* __restrict__ hpp_out) {
  extern __shared__ double sm[];
  const ba_state st = P.state[it & 1];
  if (st.done) return;
  const int tid = threadIdx.x;
  const int W = P.W, n = 6 * W, n1 = n + 1, pitch = n1 + 1;
  double* A = sm;                       // n1 x pitch (lower triangle used), row n = rhs^T
  double* s_tmp = A + (size_t)n1 * pitch;      // n1
  double* s_hpp = s_tmp + n1;           // W * 28 reduced pose values
  double* s_dp = s_hpp + W * BA_POSE_VALS;     // n
  double* s_misc = s_dp + n;            // 8
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  // ---- reduce pose partials ----
  for (int q = tid; q < W * BA_POSE_VALS; q += BA_SOLVE_THREADS) {
    const int slot = q / BA_POSE_VALS, k = q - slot * BA_POSE_VALS;
    double s = 0;
    for (int c = 0; c < P.n_chunk; c++) s += P.posepart[((size_t)slot * P.n_chunk + c) * BA_POSE_VALS + k];
    s_hpp[q] = s;
  }
  // ---- -E (lower triangle) and r (row n) ----
  for (int e = tid; e < n1 * n1; e += BA_SOLVE_THREADS) {
    const int a = e / n1, b = e - a * n1;
    if (b > a) continue;
    if (a == n && b == n) { A[(size_t)a * pitch + b] = 1.0; continue; }
    const double v = ba_tile_elem(P.tiles, P.KS, P.n_tiles, P.RT, a, b);
    A[(size_t)a * pitch + b] = (a == n) ? v : -v;   // row n: +r ; block: -E
  }
  __syncthreads();
  // ---- + damped Hpp blocks, rhs = -gp + r ----
  const double lam = st.lambda;
  for (int q = tid; q < W * 36; q += BA_SOLVE_THREADS) {
    const int slot = q / 36, rr = (q % 36) / 6, cc = q % 6;
    if (cc > rr) continue;
    // upper-packed index of (cc, rr), cc <= rr
    const int idx = cc * 6 - (cc * (cc - 1)) / 2 + (rr - cc);
    double v = s_hpp[slot * BA_POSE_VALS + idx];
    if (rr == cc) v += lam * fmax(v, 1e-12);
    A[(size_t)(6 * slot + rr) * pitch + 6 * slot + cc] += v;
  }
  for (int a = tid; a < n; a += BA_SOLVE_THREADS) A[(size_t)n * pitch + a] -= s_hpp[(a / 6) * BA_POSE_VALS + 21 + a % 6];
  __syncthreads();
  if (probe_S) {   // reduced camera system before factorisation (parity probe)
    for (int e = tid; e < n * n; e += BA_SOLVE_THREADS) {
      const int a = e / n, b = e - a * n;
      probe_S[e] = (b <= a) ? A[(size_t)a * pitch + b] : A[(size_t)b * pitch + a];
    }
    for (int a = tid; a < n; a += BA_SOLVE_THREADS) probe_S[(size_t)n * n + a] = A[(size_t)n * pitch + a];
  }
  if (hpp_out) for (int q = tid; q < W * BA_POSE_VALS; q += BA_SOLVE_THREADS) hpp_out[q] = s_hpp[q];
  // ---- left-looking Cholesky of the augmented matrix; 4 threads per row ----
  const int row = tid >> 2, part = tid & 3;
  for (int k = 0; k < n; k++) {
    double c = 0;
    if (row >= k && row < n1) {
      const double* ri = A + (size_t)row * pitch;
      const double* rk = A + (size_t)k * pitch;
      for (int j = part; j < k; j += 4) c += ri[j] * rk[j];
    }
    c += __shfl_xor(c, 1);
    c += __shfl_xor(c, 2);
    if (part == 0 && row >= k && row < n1) s_tmp[row] = A[(size_t)row * pitch + k] - c;
    __syncthreads();
    const double ck = s_tmp[k];
    if (!(ck > 0)) { if (tid == 0) s_fail = 1; }
    const double dinv = 1.0 / sqrt(ck > 0 ? ck : 1.0);
    if (part == 0 && row >= k && row < n1) A[(size_t)row * pitch + k] = (row == k) ? sqrt(ck > 0 ? ck : 1.0) : s_tmp[row] * dinv;
    __syncthreads();
  }
  // ---- back substitution  L^T dp = y  (y = row n of A) ----
  for (int a = tid; a < n; a += BA_SOLVE_THREADS) s_dp[a] = A[(size_t)n * pitch + a];
  __syncthreads();
  for (int k = n - 1; k >= 0; k--) {
    const double dk = s_dp[k] / A[(size_t)k * pitch + k];
    __syncthreads();
    if (tid == 0) s_dp[k] = dk;
    for (int i = tid; i < k; i += BA_SOLVE_THREADS) s_dp[i] -= A[(size_t)k * pitch + i] * dk;
    __syncthreads();
  }
  // ---- publish ----
  for (int a = tid; a < n; a += BA_SOLVE_THREADS) P.dp[a] = s_fail ? 0.0 : s_dp[a];
  if (tid == 0) {
    const double* poses = P.x[st.cur];
    double cost = 0, pred = 0, step2 = 0, x2 = 0, ginf = 0;
    for (int i = 0; i < W; i++) {
      cost += s_hpp[i * BA_POSE_VALS + 27];
      for (int a = 0; a < 6; a++) {
        const double g = s_hpp[i * BA_POSE_VALS + 21 + a];
        const int idx = a * 6 - (a * (a - 1)) / 2;
        const double D = fmax(s_hpp[i * BA_POSE_VALS + idx], 1e-12);
        const double d = s_fail ? 0.0 : s_dp[6 * i + a];
        pred += lam * D * d * d - g * d;
        step2 += d * d;
        x2 += poses[6 * i + a] * poses[6 * i + a];
        ginf = fmax(ginf, fabs(g));
      }
    }
    for (int b = 0; b < P.n_pblk; b++) ginf = fmax(ginf, P.gmax[b]);
    ba_info inf;
    inf.cost_cur = cost; inf.pred_pose = pred; inf.step2_pose = step2; inf.x2_pose = x2; inf.ginf = ginf;
    inf.chol_fail = s_fail; inf.pad = 0;
    *P.info = inf;
  }
  (void)s_misc;
}

// ------------------------------------------------------------------------------------------------
// k_ba_update : back-substitute landmarks, form the trial x, evaluate the trial cost
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BA_LIN_THREADS) k_ba_update(ba_ptrs P, ba_params_dev prm, int it, double* __restrict__ probe_dl) {
  __shared__ double s_cam[BA_CAM * BA_MAX_SLOTS];
  __shared__ double s_K[9];
  __shared__ double s_dp[6 * BA_MAX_SLOTS];
  __shared__ double s_pose[6 * BA_MAX_SLOTS];
  __shared__ double s_red[BA_EVAL_VALS * BA_LIN_THREADS];
  const ba_state st = P.state[it & 1];
  if (st.done) return;
  const int tid = threadIdx.x, W = P.W, N = P.N;
  const double* poses = P.x[st.cur];
  const double* pts = P.x[st.cur] + 6 * W;
  double* tposes = P.x[st.cur ^ 1];
  double* tpts = P.x[st.cur ^ 1] + 6 * W;
  for (int a = tid; a < 6 * W; a += BA_LIN_THREADS) {
    const double d = P.dp[a];
    s_dp[a] = d;
    s_pose[a] = poses[a] + d;
    if (blockIdx.x == 0) tposes[a] = poses[a] + d;
  }
  if (tid < 9) s_K[tid] = P.K[tid];
  __syncthreads();
  stage_cameras(s_pose, W, s_cam, tid, BA_LIN_THREADS);
  __syncthreads();
  const int j = blockIdx.x * BA_LIN_THREADS + tid;
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
  if (j < N) {
    const double* ax = P.aux + (size_t)j * BA_AUX;
    const double* yr = P.Yt + (size_t)(3 * j) * P.RP;
    double t0 = 0, t1 = 0, t2 = 0;
    for (int a = 0; a < 6 * W; a++) {
      const double d = s_dp[a];
      t0 += yr[a] * d; t1 += yr[P.RP + a] * d; t2 += yr[2 * P.RP + a] * d;
    }
    const double i00 = ax[9], i10 = ax[10], i11 = ax[11], i20 = ax[12], i21 = ax[13], i22 = ax[14];
    const double dl0 = -(ax[15] + i00 * t0 + i10 * t1 + i20 * t2);
    const double dl1 = -(ax[16] + i11 * t1 + i21 * t2);
    const double dl2 = -(ax[17] + i22 * t2);
    const double X0[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    const double X[3] = {X0[0] + dl0, X0[1] + dl1, X0[2] + dl2};
    tpts[3 * j] = X[0]; tpts[3 * j + 1] = X[1]; tpts[3 * j + 2] = X[2];
    if (probe_dl) { probe_dl[3 * j] = dl0; probe_dl[3 * j + 1] = dl1; probe_dl[3 * j + 2] = dl2; }
    double cost = 0;
    for (int i = 0; i < W; i++) {
      const double* ob = P.obs + ((size_t)i * N + j) * 2;
      ba_obs_lin o;
      if (ba_linearize_obs<false>(s_K, s_cam + BA_CAM * i, X, ob[0], ob[1], prm.delta, o)) cost += 0.5 * o.rho;
    }
    const double lam = st.lambda;
    v0 = cost;
    v1 = lam * (fmax(ax[0], 1e-12) * dl0 * dl0 + fmax(ax[2], 1e-12) * dl1 * dl1 + fmax(ax[5], 1e-12) * dl2 * dl2)
         - (ax[6] * dl0 + ax[7] * dl1 + ax[8] * dl2);
    v2 = dl0 * dl0 + dl1 * dl1 + dl2 * dl2;
    v3 = X0[0] * X0[0] + X0[1] * X0[1] + X0[2] * X0[2];
  }
  s_red[tid] = v0; s_red[BA_LIN_THREADS + tid] = v1; s_red[2 * BA_LIN_THREADS + tid] = v2; s_red[3 * BA_LIN_THREADS + tid] = v3;
  __syncthreads();
  for (int o = BA_LIN_THREADS / 2; o > 0; o >>= 1) {
    if (tid < o)
#pragma unroll
      for (int q = 0; q < BA_EVAL_VALS; q++) s_red[q * BA_LIN_THREADS + tid] += s_red[q * BA_LIN_THREADS + tid + o];
    __syncthreads();
  }
  if (tid < BA_EVAL_VALS) P.evalpart[blockIdx.x * BA_EVAL_VALS + tid] = s_red[tid * BA_LIN_THREADS];
}

__global__ void k_ba_finalize(ba_ptrs P, ba_params_dev prm, int n_it, double* __restrict__ x_out, ba_state* __restrict__ st_out) {
  __shared__ ba_state s_st;
  if (threadIdx.x == 0) {
    ba_state st;
    if (n_it == 0) st = P.state[0];
    else ba_decide(P.state[(n_it - 1) & 1], *P.info, P.evalpart, P.n_eblk, prm, st);
    s_st = st;
    *st_out = st;
    P.state[n_it & 1] = st;
  }
  __syncthreads();
  const double* x = P.x[s_st.cur];
  const int total = 6 * P.W + 3 * P.N;
  for (int i = threadIdx.x; i < total; i += blockDim.x) x_out[i] = x[i];
}

// per-observation residual norms at x (dense [W][N], NaN where unobserved) -- parity probe
__global__ void __launch_bounds__(BA_LIN_THREADS) k_ba_residual(ba_ptrs P, const double* __restrict__ x, double delta,
                                                                double* __restrict__ res) {
  __shared__ double s_cam[BA_CAM * BA_MAX_SLOTS];
  __shared__ double s_K[9];
  const int tid = threadIdx.x;
  stage_cameras(x, P.W, s_cam, tid, BA_LIN_THREADS);
  if (tid < 9) s_K[tid] = P.K[tid];
  __syncthreads();
  const int j = blockIdx.x * BA_LIN_THREADS + tid;
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
  void* bufs[] = {b->d_K, b->d_obs, b->d_x0, b->d_x[0], b->d_x[1], b->d_Yt, b->d_aux, b->d_posepart, b->d_gmax,
                  b->d_tiles, b->d_dp, b->d_evalpart, b->d_S, b->d_Hpp, b->d_res, b->d_xout, b->d_state, b->d_info};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (b->h_state) (void)hipHostFree(b->h_state);
  delete b;
  c->ba = nullptr;
}

extern "C" int32_t vo_ba_default_params(vo_ba_params* p) {
  if (!p) return VO_E_INVALID;
  p->max_iters = 50; p->_pad = 0; p->ftol = 1e-3; p->xtol = 1e-3; p->gtol = 1e-8; p->lambda0 = 1e-4; p->huber_delta = 1.0;
  return VO_OK;
}

static size_t ba_solve_lds(int W) {
  const int n = 6 * W, n1 = n + 1, pitch = n1 + 1;
  return sizeof(double) * ((size_t)n1 * pitch + n1 + (size_t)W * BA_POSE_VALS + n + 8);
}

static int32_t ba_alloc(vo_ctx* c, int W, int N) {
  VO_CHECK(c, W >= 1 && W <= BA_MAX_SLOTS, VO_E_CAPACITY, "window size must be 1..20");
  VO_CHECK(c, N >= 1, VO_E_INVALID, "no landmarks");
  if (c->ba && (c->ba->cap_W != W || c->ba->cap_N < N)) vo_ba_destroy(c);
  if (!c->ba) {
    vo_ba_ws* b = new vo_ba_ws();
    c->ba = b;
    b->cap_W = W; b->cap_N = N;
    const int RP = ((6 * W + 1 + 15) / 16) * 16, RT = RP / 16;
    const int K4 = ((3 * N + 3) / 4) * 4;
    const size_t nx = (size_t)6 * W + 3 * N;
    VO_HIP(c, hipMalloc((void**)&b->d_K, 9 * sizeof(double)));
    VO_HIP(c, hipMalloc((void**)&b->d_obs, sizeof(double) * 2 * W * N));
    VO_HIP(c, hipMalloc((void**)&b->d_x0, sizeof(double) * nx));
    VO_HIP(c, hipMalloc((void**)&b->d_x[0], sizeof(double) * nx));
    VO_HIP(c, hipMalloc((void**)&b->d_x[1], sizeof(double) * nx));
    VO_HIP(c, hipMalloc((void**)&b->d_Yt, sizeof(double) * (size_t)K4 * RP));
    VO_HIP(c, hipMalloc((void**)&b->d_aux, sizeof(double) * (size_t)N * BA_AUX));
    const int n_chunk = vo_div_up(N, BA_LIN_THREADS);
    VO_HIP(c, hipMalloc((void**)&b->d_posepart, sizeof(double) * (size_t)W * n_chunk * BA_POSE_VALS));
    VO_HIP(c, hipMalloc((void**)&b->d_gmax, sizeof(double) * n_chunk));
    const int n_tiles = RT * (RT + 1) / 2;
    VO_HIP(c, hipMalloc((void**)&b->d_tiles, sizeof(double) * (size_t)64 * n_tiles * 256));   // KS <= 64
    VO_HIP(c, hipMalloc((void**)&b->d_dp, sizeof(double) * 6 * W));
    VO_HIP(c, hipMalloc((void**)&b->d_evalpart, sizeof(double) * n_chunk * BA_EVAL_VALS));
    VO_HIP(c, hipMalloc((void**)&b->d_S, sizeof(double) * ((size_t)36 * W * W + 6 * W)));
    VO_HIP(c, hipMalloc((void**)&b->d_Hpp, sizeof(double) * (size_t)W * BA_POSE_VALS));
    VO_HIP(c, hipMalloc((void**)&b->d_res, sizeof(double) * (size_t)W * N));
    VO_HIP(c, hipMalloc((void**)&b->d_xout, sizeof(double) * nx));
    VO_HIP(c, hipMalloc((void**)&b->d_state, sizeof(ba_state) * 2));
    VO_HIP(c, hipMalloc((void**)&b->d_info, sizeof(ba_info)));
    VO_HIP(c, hipHostMalloc((void**)&b->h_state, sizeof(ba_state) * 2, hipHostMallocDefault));
    VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ba_solve_lds(BA_MAX_SLOTS)));
  }
  vo_ba_ws* b = c->ba;
  b->W = W; b->N = N;
  b->RP = ((6 * W + 1 + 15) / 16) * 16; b->RT = b->RP / 16; b->n_tiles = b->RT * (b->RT + 1) / 2;
  b->K4 = ((3 * N + 3) / 4) * 4;
  b->n_chunk = vo_div_up(N, BA_LIN_THREADS); b->n_pblk = b->n_chunk; b->n_eblk = b->n_chunk;
  // K-slices: ~256 rows (64 MFMA steps) per slice, at most 64 slices
  int KS = vo_div_up(b->K4, 256);
  if (KS > 64) KS = 64;
  if (KS < 1) KS = 1;
  b->KS = KS;
  b->KL = ((vo_div_up(b->K4, KS) + 3) / 4) * 4;
  return VO_OK;
}

static ba_ptrs ba_make_ptrs(vo_ba_ws* b) {
  ba_ptrs P;
  P.K = b->d_K; P.obs = b->d_obs; P.x[0] = b->d_x[0]; P.x[1] = b->d_x[1]; P.Yt = b->d_Yt; P.aux = b->d_aux;
  P.posepart = b->d_posepart; P.gmax = b->d_gmax; P.tiles = b->d_tiles; P.dp = b->d_dp; P.evalpart = b->d_evalpart;
  P.state = b->d_state; P.info = b->d_info;
  P.W = b->W; P.N = b->N; P.RP = b->RP; P.n_chunk = b->n_chunk; P.n_pblk = b->n_pblk; P.n_eblk = b->n_eblk;
  P.KS = b->KS; P.KL = b->KL; P.K4 = b->K4; P.n_tiles = b->n_tiles; P.RT = b->RT;
  return P;
}

static ba_params_dev ba_dev_params(const vo_ba_params* p) {
  ba_params_dev d;
  d.ftol = p->ftol; d.xtol = p->xtol; d.gtol = p->gtol; d.lambda0 = p->lambda0; d.delta = p->huber_delta;
  d.max_iters = p->max_iters;
  return d;
}

extern "C" int32_t vo_ba_upload(vo_ctx* c, const double* K, const double* poses, const double* points, const double* obs,
                                int32_t n_slots, int32_t n_pts) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, K && poses && points && obs, VO_E_INVALID, "null buffer");
  VO_HIP(c, hipSetDevice(c->device));
  int32_t r = ba_alloc(c, n_slots, n_pts);
  if (r != VO_OK) return r;
  vo_ba_ws* b = c->ba;
  const int W = b->W, N = b->N;
  VO_HIP(c, hipMemcpyAsync(b->d_K, K, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpyAsync(b->d_obs, obs, sizeof(double) * 2 * W * N, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpyAsync(b->d_x0, poses, sizeof(double) * 6 * W, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpyAsync(b->d_x0 + 6 * W, points, sizeof(double) * 3 * N, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemsetAsync(b->d_Yt, 0, sizeof(double) * (size_t)b->K4 * b->RP, c->stream));   // zero padding rows / columns
  VO_HIP(c, hipStreamSynchronize(c->stream));
  b->uploaded = true;
  return VO_OK;
}

// enqueue `n_it` LM iterations starting at iteration index `it0` (state must be in place)
static int32_t ba_enqueue_iters(vo_ctx* c, const ba_params_dev& prm, int it0, int n_it) {
  vo_ba_ws* b = c->ba;
  vo_prof_scope prof(c, VO_PROF_BA);
  const ba_ptrs P = ba_make_ptrs(b);
  const size_t lds = ba_solve_lds(b->W);
  for (int it = it0; it < it0 + n_it; it++) {
    hipLaunchKernelGGL(k_ba_linearize, dim3(b->n_pblk + b->W * b->n_chunk), dim3(BA_LIN_THREADS), 0, c->stream, P, prm, it, -1.0);
    hipLaunchKernelGGL(k_ba_syrk, dim3(b->KS, vo_div_up(b->n_tiles, 4)), dim3(256), 0, c->stream, P, it);
    hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(BA_SOLVE_THREADS), lds, c->stream, P, prm, it, (double*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(k_ba_update, dim3(b->n_eblk), dim3(BA_LIN_THREADS), 0, c->stream, P, prm, it, (double*)nullptr);
  }
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

static int32_t ba_begin(vo_ctx* c, const vo_ba_params* prm) {
  vo_ba_ws* b = c->ba;
  const size_t nx = (size_t)6 * b->W + 3 * b->N;
  VO_HIP(c, hipMemcpyAsync(b->d_x[0], b->d_x0, sizeof(double) * nx, hipMemcpyDeviceToDevice, c->stream));
  ba_state* s = b->h_state;
  s->lambda = prm->lambda0; s->nu = 2.0; s->cost = 0; s->cost0 = 0;
  s->cur = 0; s->iter = 0; s->accepted = 0; s->status = 0; s->done = 0; s->n_obs = 0;
  VO_HIP(c, hipMemcpyAsync(b->d_state, s, sizeof(ba_state), hipMemcpyHostToDevice, c->stream));
  return VO_OK;
}

extern "C" int32_t vo_ba_solve_resident(vo_ctx* c, const vo_ba_params* prm) {
  if (!c) return VO_E_INVALID;
  vo_ba_params def;
  if (!prm) { vo_ba_default_params(&def); prm = &def; }
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "vo_ba_upload first");
  VO_CHECK(c, prm->max_iters >= 0 && prm->max_iters <= 1000, VO_E_INVALID, "bad max_iters");
  VO_HIP(c, hipSetDevice(c->device));
  int32_t r = ba_begin(c, prm);
  if (r != VO_OK) return r;
  const ba_params_dev d = ba_dev_params(prm);
  r = ba_enqueue_iters(c, d, 0, prm->max_iters);
  if (r != VO_OK) return r;
  // publish: final decision + x_cur -> d_xout; results are read by vo_ba_fetch
  vo_ba_ws* b = c->ba;
  const ba_ptrs P = ba_make_ptrs(b);
  hipLaunchKernelGGL(k_ba_finalize, dim3(1), dim3(256), 0, c->stream, P, d, prm->max_iters, b->d_xout, b->d_state + 0);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

static void ba_fill_stats(const ba_state& s, int n_obs, vo_ba_stats* st) {
  st->cost0 = s.cost0; st->cost = s.cost; st->lambda = s.lambda; st->iters = s.iter; st->accepted = s.accepted;
  st->status = s.status; st->n_obs = n_obs;
}

extern "C" int32_t vo_ba_fetch(vo_ctx* c, double* poses_out, double* points_out, vo_ba_stats* stats) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "nothing to fetch");
  VO_HIP(c, hipSetDevice(c->device));
  vo_ba_ws* b = c->ba;
  VO_HIP(c, hipMemcpyAsync(b->h_state, b->d_state, sizeof(ba_state), hipMemcpyDeviceToHost, c->stream));
  if (poses_out) VO_HIP(c, hipMemcpyAsync(poses_out, b->d_xout, sizeof(double) * 6 * b->W, hipMemcpyDeviceToHost, c->stream));
  if (points_out) VO_HIP(c, hipMemcpyAsync(points_out, b->d_xout + 6 * b->W, sizeof(double) * 3 * b->N, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (stats) ba_fill_stats(b->h_state[0], -1, stats);
  return VO_OK;
}

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
  r = ba_begin(c, prm);
  if (r != VO_OK) return r;
  const ba_params_dev d = ba_dev_params(prm);
  const ba_ptrs P = ba_make_ptrs(b);
  // iterations are enqueued in chunks; between chunks the host peeks at the state to stop early
  const int CH = 4;
  int it = 0;
  while (it < prm->max_iters) {
    const int n = (prm->max_iters - it < CH) ? prm->max_iters - it : CH;
    r = ba_enqueue_iters(c, d, it, n);
    if (r != VO_OK) return r;
    it += n;
    hipLaunchKernelGGL(k_ba_finalize, dim3(1), dim3(256), 0, c->stream, P, d, it, b->d_xout, b->d_state + (it & 1));
    VO_HIP(c, hipMemcpyAsync(b->h_state, b->d_state + (it & 1), sizeof(ba_state), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    if (b->h_state[0].done) break;
  }
  if (prm->max_iters == 0) {
    hipLaunchKernelGGL(k_ba_finalize, dim3(1), dim3(256), 0, c->stream, P, d, 0, b->d_xout, b->d_state + 0);
    VO_HIP(c, hipMemcpyAsync(b->h_state, b->d_state, sizeof(ba_state), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
  }
  VO_HIP(c, hipMemcpy(poses_out, b->d_xout, sizeof(double) * 6 * b->W, hipMemcpyDeviceToHost));
  VO_HIP(c, hipMemcpy(points_out, b->d_xout + 6 * b->W, sizeof(double) * 3 * b->N, hipMemcpyDeviceToHost));
  if (stats) {
    int n_obs = 0;
    for (size_t k = 0; k < (size_t)b->W * b->N; k++) n_obs += (obs[2 * k] == obs[2 * k]);
    ba_fill_stats(b->h_state[0], n_obs, stats);
  }
  if (b->h_state[0].cost != b->h_state[0].cost) return vo_fail(c, VO_E_NUMERIC, "bundle adjustment produced a non-finite cost");
  return VO_OK;
}

extern "C" int32_t vo_ba_probe(vo_ctx* c, double lambda, double huber_delta, double* residual, int32_t* n_obs, double* cost,
                               double* Hpp, double* gp, double* Hll, double* gl, double* S, double* rhs, double* dposes,
                               double* dpoints) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->ba && c->ba->uploaded, VO_E_STATE, "vo_ba_upload first");
  VO_HIP(c, hipSetDevice(c->device));
  vo_ba_ws* b = c->ba;
  const int W = b->W, N = b->N, n = 6 * W;
  vo_ba_params prm;
  vo_ba_default_params(&prm);
  prm.huber_delta = huber_delta; prm.lambda0 = lambda; prm.max_iters = 1;
  int32_t r = ba_begin(c, &prm);
  if (r != VO_OK) return r;
  const ba_params_dev d = ba_dev_params(&prm);
  const ba_ptrs P = ba_make_ptrs(b);
  const size_t lds = ba_solve_lds(W);
  hipLaunchKernelGGL(k_ba_residual, dim3(b->n_chunk), dim3(BA_LIN_THREADS), 0, c->stream, P, b->d_x0, huber_delta, b->d_res);
  hipLaunchKernelGGL(k_ba_linearize, dim3(b->n_pblk + W * b->n_chunk), dim3(BA_LIN_THREADS), 0, c->stream, P, d, 0, lambda);
  hipLaunchKernelGGL(k_ba_syrk, dim3(b->KS, vo_div_up(b->n_tiles, 4)), dim3(256), 0, c->stream, P, 0);
  hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(BA_SOLVE_THREADS), lds, c->stream, P, d, 0, b->d_S, b->d_Hpp);
  // d_points land in the front of the eval scratch-free buffer: reuse d_x[1] after the update via probe_dl
  double* d_dl = nullptr;
  VO_HIP(c, hipMalloc((void**)&d_dl, sizeof(double) * 3 * N));
  hipLaunchKernelGGL(k_ba_update, dim3(b->n_eblk), dim3(BA_LIN_THREADS), 0, c->stream, P, d, 0, d_dl);
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipStreamSynchronize(c->stream));
  // ---- copy out ----
  double* h = (double*)malloc(sizeof(double) * ((size_t)W * N + (size_t)N * BA_AUX + (size_t)W * BA_POSE_VALS + (size_t)n * n + n + n + 3 * (size_t)N));
  if (!h) { (void)hipFree(d_dl); return vo_fail(c, VO_E_NOMEM, "probe host buffer"); }
  double* h_res = h; double* h_aux = h_res + (size_t)W * N; double* h_hpp = h_aux + (size_t)N * BA_AUX;
  double* h_S = h_hpp + (size_t)W * BA_POSE_VALS; double* h_dp = h_S + (size_t)n * n + n; double* h_dl = h_dp + n;
  hipError_t e = hipMemcpy(h_res, b->d_res, sizeof(double) * (size_t)W * N, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_aux, b->d_aux, sizeof(double) * (size_t)N * BA_AUX, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_hpp, b->d_Hpp, sizeof(double) * (size_t)W * BA_POSE_VALS, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_S, b->d_S, sizeof(double) * ((size_t)n * n + n), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_dp, b->d_dp, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_dl, d_dl, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToHost);
  (void)hipFree(d_dl);
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
