// 3D-2D pose from tracked landmarks: RANSAC over P3P hypotheses + Gauss-Newton refinement on the consensus set
// (SURVEY.md 8f "next" row 1).
//
// Replaces cv2.solvePnPRansac(pts3d, pts2d, K, None, reprojectionError, iterationsCount=1e6, confidence=0.9999) in
// Extractor.camera_pose(corr='3D-2D'), /root/reference/src/extractor/extractor.py:174-191 (Pipeline.step then prunes
// the non-inlier landmarks, pipeline.py:124-137).  OpenCV's RANSAC (own MWC generator, EPnP on 5-point samples, LM
// refinement) cannot be reproduced draw by draw; what the call's contract determines is: consensus set = points with
// squared reprojection error <= reprojectionError^2, stop when an all-inlier sample has been drawn with probability
// `confidence` (RANSACUpdateNumIters), returned pose = minimiser of the reprojection error over the consensus set.
// The algorithm here is defined by oracle/pnp_oracle.py (hypothesis h = 4 indices from splitmix64(seed, h, draw);
// Grunert P3P on three, the fourth disambiguates; batches of 256 hypotheses; ties to the smallest h) and is compared
// with it hypothesis by hypothesis.
//
// GPU mapping: a LANE per hypothesis for the minimal solve, then FOUR WAVES per hypothesis for the consensus count (lanes
// stride over the points).  32 (first batch of a search) or 256 hypotheses x batch sequences per launch; a one-workgroup-per-sequence kernel keeps the running best and the
// iteration bound on the device, another one refines.
#include "vo_internal.h"

#include <math.h>
#include <string.h>

#define PNP_BATCH 256
#ifndef PNP_FIRST_BATCH
#define PNP_FIRST_BATCH 32   // hypotheses of the FIRST batch of a search (oracle/pnp_oracle.py: first_batch).  OpenCV re-evaluates its iteration
                             // bound after every hypothesis -- at the closed loop's ~98 % inliers and 0.9999 confidence it stops after 4 --, so a
                             // small first batch is closer to the reference AND cheaper: scoring 256 hypotheses x 2 000 points was 35 of the 59 us
                             // a frame's first batch took in a batch of 32 sequences, all but the first handful for nothing
#endif

struct pnp_hyp { double R[9]; double t[3]; int count; int h; };
struct pnp_ctrl { int niters; int h_done; int done; int pad; pnp_hyp best; };

struct vo_pnp_ws {
  int cap = 0;
  double* d_K = nullptr;        // [B][9]
  float* d_X = nullptr;         // [B][cap][3]
  float* d_uv = nullptr;        // [B][cap][2]
  pnp_hyp* d_hyp = nullptr;     // [B][PNP_BATCH]
  pnp_ctrl* d_ctrl = nullptr;   // [B]
  uint8_t* d_mask = nullptr;    // [B][cap]
  double* d_out = nullptr;      // [B][8]: rvec, t, cost, n_inliers
  pnp_ctrl* h_ctrl = nullptr;   // pinned
  double* h_out = nullptr;      // pinned
  int n = 0;                    // resident correspondences per sequence
};

// ------------------------------------------------------------------------------------------------
// device helpers (mirror oracle/pnp_oracle.py line by line)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pnp_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ inline void pnp_sample4(unsigned seed, unsigned h, int n, int idx[4]) {
  int cnt = 0;
  unsigned k = 0;
  while (cnt < 4) {
    const unsigned long long r = (k < 256) ? pnp_splitmix64(((unsigned long long)(seed & 0xFFFFFFu) << 40) ^ ((unsigned long long)h << 8) ^ (unsigned long long)(k & 0xFFu))
                                           : pnp_splitmix64((unsigned long long)k);
    const int i = (int)((r >> 11) % (unsigned long long)n);
    k++;
    bool dup = false;
    for (int j = 0; j < cnt; j++) dup = dup || (idx[j] == i);
    if (!dup) idx[cnt++] = i;
  }
}

__device__ inline double pnp_cubic_root(double A, double B, double C) {   // largest real root of z^3 + A z^2 + B z + C
  const double P = B - A * A / 3.0;
  const double Q = 2.0 * A * A * A / 27.0 - A * B / 3.0 + C;
  const double disc = Q * Q / 4.0 + P * P * P / 27.0;
  double t;
  if (disc > 0) {
    const double sq = sqrt(disc);
    t = cbrt(-Q / 2.0 + sq) + cbrt(-Q / 2.0 - sq);
  } else if (P == 0.0) {
    t = 0.0;
  } else {
    const double m = 2.0 * sqrt(-P / 3.0);
    double arg = 3.0 * Q / (P * m);
    arg = fmin(1.0, fmax(-1.0, arg));
    t = m * cos(acos(arg) / 3.0);
  }
  return t - A / 3.0;
}

__device__ inline int pnp_quartic(double c4, double c3, double c2, double c1, double c0, double x[4]) {
  if (fabs(c4) < 1e-300) return 0;
  const double a = c3 / c4, b = c2 / c4, c = c1 / c4, d = c0 / c4;
  const double p = b - 3.0 * a * a / 8.0;
  const double q = c - a * b / 2.0 + a * a * a / 8.0;
  const double r = d - a * c / 4.0 + a * a * b / 16.0 - 3.0 * a * a * a * a / 256.0;
  double ys[4];
  int ny = 0;
  if (fabs(q) < 1e-14 * (1.0 + pow(fabs(p), 1.5))) {
    const double disc = p * p - 4.0 * r;
    if (disc >= 0) {
      const double sq = sqrt(disc);
      const double y2a = (-p + sq) / 2.0, y2b = (-p - sq) / 2.0;
      if (y2a >= 0) { ys[ny++] = sqrt(y2a); ys[ny++] = -sqrt(y2a); }
      if (y2b >= 0) { ys[ny++] = sqrt(y2b); ys[ny++] = -sqrt(y2b); }
    }
  } else {
    const double z0 = pnp_cubic_root(2.0 * p, p * p - 4.0 * r, -q * q);
    if (z0 > 0) {
      const double s = sqrt(z0);
      for (int k = 0; k < 2; k++) {
        const double sg = k ? -1.0 : 1.0;
        const double bb = sg * s, cc = (p + z0) / 2.0 - sg * q / (2.0 * s);
        const double disc = bb * bb - 4.0 * cc;
        if (disc >= 0) {
          const double sq = sqrt(disc);
          ys[ny++] = (-bb + sq) / 2.0; ys[ny++] = (-bb - sq) / 2.0;
        }
      }
    }
  }
  for (int i = 0; i < ny; i++) {
    double xv = ys[i] - a / 4.0;
    for (int it = 0; it < 2; it++) {
      const double f = (((c4 * xv + c3) * xv + c2) * xv + c1) * xv + c0;
      const double fp = ((4.0 * c4 * xv + 3.0 * c3) * xv + 2.0 * c2) * xv + c1;
      if (fp != 0.0) xv -= f / fp;
    }
    x[i] = xv;
  }
  return ny;
}

__device__ __forceinline__ void v3_sub(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
__device__ __forceinline__ double v3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void v3_cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// 1 / x from v_rcp_f64 + two Newton steps (the IEEE divide expands to ~30 dependent instructions and the projection is
// evaluated 16 M times per batch); relative error ~1e-16: a point within that of the threshold may be classified
// differently from the oracle's true division -- the tests allow for such borderline points
__device__ __forceinline__ double pnp_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  return fma(fma(-x, y, 1.0), y, y);
}

__device__ __forceinline__ double pnp_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}

// wave-wide sum in every lane: DPP row rotations + permlane swaps (no LDS traffic, unlike __shfl_xor on f64)
template <int CTRL>
__device__ __forceinline__ double pnp_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pnp_wave_sum(double v) {
  v += pnp_dpp<0x128>(v); v += pnp_dpp<0x124>(v); v += pnp_dpp<0x122>(v); v += pnp_dpp<0x121>(v);   // row_ror 8, 4, 2, 1
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
  }
  return v;
}

// squared reprojection error of one point
__device__ __forceinline__ double pnp_err2(const double* K, const double* R, const double* t, const double* X, double u, double v) {
  const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  const double p0 = K[0] * xc + K[1] * yc + K[2] * zc, p1 = K[3] * xc + K[4] * yc + K[5] * zc, p2 = K[6] * xc + K[7] * yc + K[8] * zc;
  const double ip2 = pnp_rcp(p2);
  const double du = p0 * ip2 - u, dv = p1 * ip2 - v;
  return du * du + dv * dv;
}

// Grunert P3P on three correspondences, the fourth picks among the solutions.  -> true if a pose was found
__device__ inline bool pnp_hypothesis(const double* K, const double* Kinv, const double P[4][3], const double uv[4][2], double* Rb, double* tb) {
  double f[3][3];
  for (int i = 0; i < 3; i++) {
    const double bx = Kinv[0] * uv[i][0] + Kinv[1] * uv[i][1] + Kinv[2];
    const double by = Kinv[3] * uv[i][0] + Kinv[4] * uv[i][1] + Kinv[5];
    const double bz = Kinv[6] * uv[i][0] + Kinv[7] * uv[i][1] + Kinv[8];
    const double nn = sqrt(bx * bx + by * by + bz * bz);
    f[i][0] = bx / nn; f[i][1] = by / nn; f[i][2] = bz / nn;
  }
  double d12[3], d02[3], d01[3];
  v3_sub(P[1], P[2], d12); v3_sub(P[0], P[2], d02); v3_sub(P[0], P[1], d01);
  const double a2 = v3_dot(d12, d12), b2 = v3_dot(d02, d02), c2 = v3_dot(d01, d01);
  if (!(b2 > 0) || !(a2 > 0) || !(c2 > 0)) return false;
  const double ca = v3_dot(f[1], f[2]), cb = v3_dot(f[0], f[2]), cg = v3_dot(f[0], f[1]);
  const double A = a2 / b2, C = c2 / b2;
  const double A4 = A * A - 2 * A * C - 2 * A + C * C - 4 * C * ca * ca + 2 * C + 1;
  const double A3 = -4 * (A * A * cb - 2 * A * C * cb - A * ca * cg - A * cb + C * C * cb - 2 * C * ca * ca * cb - C * ca * cg + C * cb + ca * cg);
  const double A2 = 2 * (2 * A * A * cb * cb + A * A - 4 * A * C * cb * cb - 2 * A * C - 4 * A * ca * cb * cg - 2 * A * cg * cg + 2 * C * C * cb * cb + C * C
                         - 2 * C * ca * ca - 4 * C * ca * cb * cg + 2 * ca * ca + 2 * cg * cg - 1);
  const double A1 = -4 * (A * A * cb - 2 * A * C * cb - A * ca * cg - 2 * A * cb * cg * cg + A * cb + C * C * cb - C * ca * cg - C * cb + ca * cg);
  const double A0 = A * A - 2 * A * C - 4 * A * cg * cg + 2 * A + C * C - 2 * C + 1;
  const double qq = A - C;
  double e1[3], e2[3], e3[3], tmp[3];
  v3_sub(P[1], P[0], e1);
  { const double nn = sqrt(v3_dot(e1, e1)); e1[0] /= nn; e1[1] /= nn; e1[2] /= nn; }
  v3_sub(P[2], P[0], tmp); v3_cross(e1, tmp, e3);
  const double n3 = sqrt(v3_dot(e3, e3));
  if (!(n3 > 0)) return false;
  e3[0] /= n3; e3[1] /= n3; e3[2] /= n3;
  v3_cross(e3, e1, e2);
  double roots[4];
  const int nr = pnp_quartic(A4, A3, A2, A1, A0, roots);
  bool found = false;
  double best_e = 0;
  for (int k = 0; k < nr; k++) {
    const double v = roots[k];
    const double den = 2.0 * (cg - v * ca);
    if (!(v > 0) || fabs(den) < 1e-12) continue;
    const double u = ((qq - 1.0) * v * v - 2.0 * qq * cb * v + 1.0 + qq) / den;
    const double w = 1.0 + v * v - 2.0 * v * cb;
    if (!(u > 0) || !(w > 0)) continue;
    const double s1 = sqrt(b2 / w);
    double Q[3][3];
    for (int c = 0; c < 3; c++) { Q[0][c] = s1 * f[0][c]; Q[1][c] = u * s1 * f[1][c]; Q[2][c] = v * s1 * f[2][c]; }
    double g1[3], g2[3], g3[3];
    v3_sub(Q[1], Q[0], g1);
    { const double nn = sqrt(v3_dot(g1, g1)); g1[0] /= nn; g1[1] /= nn; g1[2] /= nn; }
    v3_sub(Q[2], Q[0], tmp); v3_cross(g1, tmp, g3);
    const double m3 = sqrt(v3_dot(g3, g3));
    if (!(m3 > 0)) continue;
    g3[0] /= m3; g3[1] /= m3; g3[2] /= m3;
    v3_cross(g3, g1, g2);
    double R[9], t[3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[3 * i + j] = g1[i] * e1[j] + g2[i] * e2[j] + g3[i] * e3[j];
    for (int i = 0; i < 3; i++) t[i] = Q[0][i] - (R[3 * i] * P[0][0] + R[3 * i + 1] * P[0][1] + R[3 * i + 2] * P[0][2]);
    const double e = pnp_err2(K, R, t, P[3], uv[3][0], uv[3][1]);
    if (e < (found ? best_e : __builtin_inf())) {          // NaN never wins
      found = true; best_e = e;
      for (int i = 0; i < 9; i++) Rb[i] = R[i];
      for (int i = 0; i < 3; i++) tb[i] = t[i];
    }
  }
  return found;
}

__device__ inline void pnp_inv3(const double* K, double* Ki) {
  const double c0 = K[4] * K[8] - K[5] * K[7], c1 = K[5] * K[6] - K[3] * K[8], c2 = K[3] * K[7] - K[4] * K[6];
  const double det = K[0] * c0 + K[1] * c1 + K[2] * c2;
  const double id = 1.0 / det;
  Ki[0] = c0 * id; Ki[1] = (K[2] * K[7] - K[1] * K[8]) * id; Ki[2] = (K[1] * K[5] - K[2] * K[4]) * id;
  Ki[3] = c1 * id; Ki[4] = (K[0] * K[8] - K[2] * K[6]) * id; Ki[5] = (K[2] * K[3] - K[0] * K[5]) * id;
  Ki[6] = c2 * id; Ki[7] = (K[1] * K[6] - K[0] * K[7]) * id; Ki[8] = (K[0] * K[4] - K[1] * K[3]) * id;
}

// ------------------------------------------------------------------------------------------------
// k_pnp_solve : grid (PNP_BATCH / 64, batch): LANE per hypothesis -- the minimal solve (sample, Grunert quartic, triad,
//               disambiguation; ~3 k f64 operations with cbrt / acos / cos) runs once per hypothesis instead of once per
//               lane of a wave
// k_pnp_score : grid (hypotheses of the batch, batch): FOUR WAVES per hypothesis -- the lanes stride over the points, shuffle + LDS sum
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_pnp_solve(const double* __restrict__ Kall, const float* __restrict__ Xall, const float* __restrict__ uvall,
                                                  int cap, int n, unsigned seed, pnp_hyp* __restrict__ hyps, const pnp_ctrl* __restrict__ ctrl,
                                                  const int32_t* __restrict__ counts, int first, int nb) {
  const int b = blockIdx.y, slot = blockIdx.x * 64 + threadIdx.x;
  if (slot >= nb) return;                                   // nb: hypotheses of this batch
  if (counts) n = counts[b];
  // first = 1: the first batch of a search -- the control block is taken as freshly initialised (k_pnp_select of this batch writes it),
  // which saves the closed loop the k_pnp_init launch in front of every frame's search
  const int h_done = first ? 0 : ctrl[b].h_done, done = first ? 0 : ctrl[b].done;
  pnp_hyp* out = hyps + (size_t)b * PNP_BATCH + slot;
  const int h = h_done + slot;
  out->h = h;
  if (done || n < 4) { out->count = -1; return; }         // fewer than 4 correspondences: no sample exists
  const double* Kp = Kall + 9 * b;
  const float* X = Xall + (size_t)b * cap * 3;
  const float* uv = uvall + (size_t)b * cap * 2;
  double K[9], Kinv[9];
  for (int i = 0; i < 9; i++) K[i] = Kp[i];
  pnp_inv3(K, Kinv);
  int idx[4];
  pnp_sample4(seed, (unsigned)h, n, idx);
  double P[4][3], q[4][2];
  for (int i = 0; i < 4; i++) {
    for (int c = 0; c < 3; c++) P[i][c] = (double)X[3 * idx[i] + c];
    q[i][0] = (double)uv[2 * idx[i]]; q[i][1] = (double)uv[2 * idx[i] + 1];
  }
  double R[9], t[3];
  const bool ok = pnp_hypothesis(K, Kinv, P, q, R, t);
  out->count = ok ? 0 : -1;               // -1: no pose, k_pnp_score skips it
  for (int i = 0; i < 9; i++) out->R[i] = ok ? R[i] : 0.0;
  for (int i = 0; i < 3; i++) out->t[i] = ok ? t[i] : 0.0;
}

// (four waves per hypothesis: with one, a wave walked 2 000 points in 32 dependent trips -- 16 us for the 32 hypotheses of a first batch,
//  where the launch is 1 024 waves on 1 024 SIMDs and nothing hides the latency)
#define PNP_SCORE_THREADS 256
__global__ void __launch_bounds__(PNP_SCORE_THREADS) k_pnp_score(const double* __restrict__ Kall, const float* __restrict__ Xall, const float* __restrict__ uvall,
                                                  int cap, int n, double thr2, pnp_hyp* __restrict__ hyps, const int32_t* __restrict__ counts) {
  __shared__ int s_part[PNP_SCORE_THREADS / 64];
  const int b = blockIdx.y, lane = threadIdx.x;
  if (counts) n = counts[b];
  pnp_hyp* hp = hyps + (size_t)b * PNP_BATCH + blockIdx.x;
  const int c0 = hp->count;
  __syncthreads();                                                      // every wave has read the flag before thread 0 overwrites it
  if (c0 < 0) { if (lane == 0) hp->count = 0; return; }                 // no pose for this sample (uniform over the workgroup)
  const double* Kp = Kall + 9 * b;
  const float* X = Xall + (size_t)b * cap * 3;
  const float* uv = uvall + (size_t)b * cap * 2;
  double K[9], R[9], t[3];
  for (int i = 0; i < 9; i++) { K[i] = Kp[i]; R[i] = hp->R[i]; }
  for (int i = 0; i < 3; i++) t[i] = hp->t[i];
  int cnt = 0;
  for (int i = lane; i < n; i += PNP_SCORE_THREADS) {
    const double Xi[3] = {(double)X[3 * i], (double)X[3 * i + 1], (double)X[3 * i + 2]};
    cnt += (pnp_err2(K, R, t, Xi, (double)uv[2 * i], (double)uv[2 * i + 1]) <= thr2) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((lane & 63) == 0) s_part[lane >> 6] = cnt;
  __syncthreads();
  if (lane == 0) { int tot = 0; for (int w = 0; w < PNP_SCORE_THREADS / 64; w++) tot += s_part[w]; hp->count = tot; }
}

// OpenCV RANSACUpdateNumIters (calib3d/ptsetreg.cpp)
__device__ inline int pnp_update_iters(double p, double ep, int model_points, int max_iters) {
  p = fmin(fmax(p, 0.0), 1.0); ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1.0 - p, 2.2250738585072014e-308);
  double denom = 1.0 - pow(1.0 - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num); denom = log(denom);
  return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)rint(num / denom);
}

// ------------------------------------------------------------------------------------------------
// k_pnp_select : grid (batch): running best (most inliers, ties to the smallest h) and the iteration bound
// ------------------------------------------------------------------------------------------------
__device__ inline void pnp_ctrl_reset(pnp_ctrl* c, int max_iters) {
  c->niters = max_iters; c->h_done = 0; c->done = 0; c->pad = 0;
  c->best.count = 0; c->best.h = -1;
  for (int i = 0; i < 9; i++) c->best.R[i] = 0;
  for (int i = 0; i < 3; i++) c->best.t[i] = 0;
}

__global__ void __launch_bounds__(PNP_BATCH) k_pnp_select(const pnp_hyp* __restrict__ hyps, pnp_ctrl* __restrict__ ctrl, int n, double conf,
                                                          int max_iters, const int32_t* __restrict__ counts, int first, int nb) {
  __shared__ int s_cnt[PNP_BATCH];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (counts) n = counts[b];
  pnp_ctrl* c = ctrl + b;
  if (!first && c->done) return;
  if (n < 4) { if (tid == 0) { if (first) pnp_ctrl_reset(c, max_iters); c->done = 1; } return; }
  const pnp_hyp* H = hyps + (size_t)b * PNP_BATCH;
  s_cnt[tid] = tid < nb ? H[tid].count : -1;
  __syncthreads();
  if (tid == 0) {
    if (first) pnp_ctrl_reset(c, max_iters);               // (see k_pnp_solve)
    int bi = -1, bc = c->best.count;
    for (int i = 0; i < nb; i++) if (s_cnt[i] > bc) { bc = s_cnt[i]; bi = i; }    // batch order = ascending h
    if (bi >= 0) c->best = H[bi];
    c->h_done += nb;
    if (c->best.count > 0) {
      const int ni = pnp_update_iters(conf, (double)(n - c->best.count) / (double)n, 4, max_iters);
      if (ni < c->niters) c->niters = ni;
    }
    c->done = (c->h_done >= c->niters) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// k_pnp_refine : grid (batch), 256 threads: consensus set of the best hypothesis, Gauss-Newton with step halving
// ------------------------------------------------------------------------------------------------
__device__ inline void pnp_rodrigues(const double* r, double* R) {
  const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  double a, bq;
  if (th < 1e-12) { a = 1.0; bq = 0.0; }
  else { double sn, cs; sincos(th, &sn, &cs); a = sn / th; bq = (1.0 - cs) / (th * th); }      // one range reduction for both
  const double K0[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double kk = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) kk += K0[3 * i + k] * K0[3 * k + j];
      R[3 * i + j] = (i == j ? 1.0 : 0.0) + a * K0[3 * i + j] + bq * kk;
    }
}

// trial pose of the Gauss-Newton step d scaled by `step`: Rn = exp(step d[0:3]) R, tn = t + step d[3:6]
__device__ inline void pnp_trial_pose(const double* R, const double* t, const double* d, double step, double* Rn, double* tn) {
  const double w[3] = {step * d[0], step * d[1], step * d[2]};
  double E[9];
  pnp_rodrigues(w, E);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Rn[3 * i + j] = E[3 * i] * R[j] + E[3 * i + 1] * R[3 + j] + E[3 * i + 2] * R[6 + j];
#pragma unroll
  for (int i = 0; i < 3; i++) tn[i] = t[i] + step * d[3 + i];
}

__device__ inline void pnp_log_so3(const double* R, double* r) {
  double c = (R[0] + R[4] + R[8] - 1.0) / 2.0;
  c = fmin(1.0, fmax(-1.0, c));
  const double th = acos(c);
  const double w[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  if (th < 1e-10) { r[0] = 0.5 * w[0]; r[1] = 0.5 * w[1]; r[2] = 0.5 * w[2]; return; }
  if (3.141592653589793 - th < 1e-6) {
    const double A[9] = {(R[0] + 1) / 2, R[1] / 2 + R[3] / 2, R[2] / 2 + R[6] / 2, R[3] / 2 + R[1] / 2, (R[4] + 1) / 2, R[5] / 2 + R[7] / 2,
                         R[6] / 2 + R[2] / 2, R[7] / 2 + R[5] / 2, (R[8] + 1) / 2};
    const double ax[3] = {sqrt(fmax(A[0], 0.0)), sqrt(fmax(A[4], 0.0)), sqrt(fmax(A[8], 0.0))};
    int i = 0;
    if (ax[1] > ax[i]) i = 1;
    if (ax[2] > ax[i]) i = 2;
    double v[3] = {A[i] / ax[i], A[3 + i] / ax[i], A[6 + i] / ax[i]};
    if (w[0] * v[0] + w[1] * v[1] + w[2] * v[2] < 0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
    r[0] = th * v[0]; r[1] = th * v[1]; r[2] = th * v[2];
    return;
  }
  const double k = th / (2.0 * sin(th));
  r[0] = k * w[0]; r[1] = k * w[1]; r[2] = k * w[2];
}

#define PNP_NRED 28    // 21 (J^T J upper) + 6 (J^T e) + 1 (cost)

#define PNP_REFINE_THREADS 512      // 8 waves per sequence (256 VGPRs each: the 28 partial sums + the Jacobian stay in registers; 1024 threads spilled 41)
#define PNP_REFINE_WAVES (PNP_REFINE_THREADS / 64)

// Reduce-scatter steps of the 28 wave sums: x and y are summed over lane pairs at once, one half of the lanes keeps x's sum, the other y's
// (three instructions for two values; the all-reduce of pnp_wave_sum spends eighteen per value -- 504 per block sum, now ~130)
__device__ __forceinline__ double pnp_rs32(double x, double y) {       // lanes 0..31: x[l] + x[l + 32]; lanes 32..63: y[l - 32] + y[l]
  const auto l2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
__device__ __forceinline__ double pnp_rs16(double x, double y) {       // rows 0, 2: x[row] + x[row + 1]; rows 1, 3: y[row - 1] + y[row]
  const auto l2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}
__device__ __forceinline__ double pnp_rs8(double x, double y, int lane) {   // bit 3 clear: x[l] + x[l ^ 8]; set: y[l ^ 8] + y[l]
  const bool hi = (lane & 8) != 0;
  const double keep = hi ? y : x, send = hi ? x : y;
  return keep + pnp_dpp<0x128>(send);                                       // row_ror:8 = lane ^ 8 inside a row of 16
}

// fixed-order block sum of v[0 .. nv): wave sums (nv = PNP_NRED: as a reduce-scatter -- after the three scatter stages the lane whose bits
// 5, 4, 3 spell (b5, b4, b3) holds value 8 i + 4 b3 + 2 b4 + b5 in slot i, summed over eight lanes; one all-reduce over the remaining eight
// lanes finishes four values at a time), then the wave partials as a fixed pairwise tree
__device__ inline void pnp_block_sum(double* v, int nv, double* s_red /* PNP_REFINE_WAVES * PNP_NRED */, double* s_out /* PNP_NRED */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (nv == PNP_NRED) {
    double a[14], c[8], e[4];
#pragma unroll
    for (int i = 0; i < 14; i++) a[i] = pnp_rs32(v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 7; i++) c[i] = pnp_rs16(a[2 * i], a[2 * i + 1]);
    c[7] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double x = pnp_rs8(c[2 * i], c[2 * i + 1], lane);
      x += pnp_dpp<0x141>(x);        // row_half_mirror: x[l] + x[7 - l] inside every group of 8
      x += pnp_dpp<0xB1>(x);         // quad_perm [1, 0, 3, 2]
      x += pnp_dpp<0x4E>(x);         // quad_perm [2, 3, 0, 1]
      e[i] = x;
    }
    if ((lane & 7) == 0) {
      const int sub = ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
#pragma unroll
      for (int i = 0; i < 4; i++) if (8 * i + sub < PNP_NRED) s_red[wave * PNP_NRED + 8 * i + sub] = e[i];
    }
  } else
  for (int k = 0; k < nv; k++) {
    const double x = pnp_wave_sum(v[k]);
    if (lane == 0) s_red[wave * PNP_NRED + k] = x;
  }
  __syncthreads();
  if (tid < nv) {
    double p[PNP_REFINE_WAVES];
#pragma unroll
    for (int w = 0; w < PNP_REFINE_WAVES; w++) p[w] = s_red[w * PNP_NRED + tid];
#pragma unroll
    for (int st = 1; st < PNP_REFINE_WAVES; st <<= 1)
#pragma unroll
      for (int w = 0; w + st < PNP_REFINE_WAVES; w += 2 * st) p[w] += p[w + st];
    s_out[tid] = p[0];
  }
  __syncthreads();
}

// PPT > 0: a thread's points (i = tid + k * PNP_REFINE_THREADS, k < PPT) and their inlier bits stay in registers for all passes -- every
// pass is then arithmetic only, not a round trip to the L2 for X, uv and the mask; PPT = 0: any n, the passes re-read memory
template <int PPT>
__global__ void __launch_bounds__(PNP_REFINE_THREADS) k_pnp_refine(const double* __restrict__ Kall, const float* __restrict__ Xall, const float* __restrict__ uvall,
                                                    int cap, int n, double thr2, const pnp_ctrl* __restrict__ ctrl, uint8_t* __restrict__ mask_all,
                                                    double* __restrict__ out_all, const int32_t* __restrict__ counts) {
  __shared__ double s_red[PNP_REFINE_WAVES * PNP_NRED], s_sum[PNP_NRED];
  __shared__ double s_R[9], s_t[3], s_Rn[9], s_tn[3], s_d[6];
  __shared__ int s_flag;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (counts) n = counts[b];
  const double* K = Kall + 9 * b;
  const float* X = Xall + (size_t)b * cap * 3;
  const float* uv = uvall + (size_t)b * cap * 2;
  uint8_t* mask = mask_all + (size_t)b * cap;
  double* out = out_all + 8 * b;
  const pnp_hyp* best = &ctrl[b].best;     // (a by-value copy indexed by tid below would live in scratch memory)
  if (best->count < 4) {
    for (int i = tid; i < n; i += PNP_REFINE_THREADS) mask[i] = 0;
    if (tid < 8) out[tid] = (tid == 7) ? 0.0 : __builtin_nan("");
    return;
  }
  if (tid < 9) s_R[tid] = best->R[tid];
  if (tid < 3) s_t[tid] = best->t[tid];
  __syncthreads();
  constexpr int NP = PPT > 0 ? PPT : 1;
  float px[NP][3], pu[NP][2];
  unsigned mbits = 0;
  if (PPT > 0) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int i = tid + k * PNP_REFINE_THREADS;
      const bool in = i < n;                            // (n <= PPT * PNP_REFINE_THREADS: the launcher picks PPT)
      const int ic = in ? i : 0;
      px[k][0] = X[3 * ic]; px[k][1] = X[3 * ic + 1]; px[k][2] = X[3 * ic + 2]; pu[k][0] = uv[2 * ic]; pu[k][1] = uv[2 * ic + 1];
      mbits |= in ? (1u << k) : 0u;
    }
  }
  // f(X, u, v) for every point of this thread whose bit is set in `bits` (register copy) / whose mask byte is set (memory), in index order
  auto for_points = [&](const bool use_mask, auto&& f) {
    if (PPT > 0) {
#pragma unroll
      for (int k = 0; k < NP; k++) {
        if (!((mbits >> k) & 1u)) continue;
        const double Xw[3] = {(double)px[k][0], (double)px[k][1], (double)px[k][2]};
        f(tid + k * PNP_REFINE_THREADS, k, Xw, (double)pu[k][0], (double)pu[k][1]);
      }
    } else {
      for (int i = tid; i < n; i += PNP_REFINE_THREADS) {
        if (use_mask && !mask[i]) continue;
        const double Xw[3] = {(double)X[3 * i], (double)X[3 * i + 1], (double)X[3 * i + 2]};
        f(i, 0, Xw, (double)uv[2 * i], (double)uv[2 * i + 1]);
      }
    }
  };
  int n_in = 0;
  {
    unsigned inl = 0;
    for_points(false, [&](int i, int k, const double* Xw, double u, double v) {
      const uint8_t m = (pnp_err2(K, s_R, s_t, Xw, u, v) <= thr2) ? 1 : 0;
      mask[i] = m; n_in += m; inl |= (unsigned)m << k;
    });
    if (PPT > 0) mbits = inl;                           // from here on: the consensus set
  }
  __syncthreads();     // (PPT = 0: the mask is read back below by the same threads that wrote it, same indices: no hazard, keeps phases tidy)
  // normal equations J^T J (21), J^T e (6) and the cost (1) of the masked points at pose (R, t): per-thread partial sums
  auto linearise = [&](const double* R, const double* t, double* acc) {
#pragma unroll
    for (int k = 0; k < PNP_NRED; k++) acc[k] = 0;
    for_points(true, [&](int, int, const double* Xw, double uo, double vo) {
      const double rx = R[0] * Xw[0] + R[1] * Xw[1] + R[2] * Xw[2];
      const double ry = R[3] * Xw[0] + R[4] * Xw[1] + R[5] * Xw[2];
      const double rz = R[6] * Xw[0] + R[7] * Xw[1] + R[8] * Xw[2];
      const double xc = rx + t[0], yc = ry + t[1], zc = rz + t[2];
      const double p0 = K[0] * xc + K[1] * yc + K[2] * zc, p1 = K[3] * xc + K[4] * yc + K[5] * zc, p2 = K[6] * xc + K[7] * yc + K[8] * zc;
      const double ip2 = pnp_rcp(p2);
      const double u = p0 * ip2, v = p1 * ip2;
      const double e0 = u - uo, e1 = v - vo;
      double A[2][3];
#pragma unroll
      for (int c = 0; c < 3; c++) { A[0][c] = (K[c] - u * K[6 + c]) * ip2; A[1][c] = (K[3 + c] - v * K[6 + c]) * ip2; }
      // left perturbation exp(w) R X: d(RX)/dw_k = e_k x (RX)
      const double G[3][3] = {{0.0, rz, -ry}, {-rz, 0.0, rx}, {ry, -rx, 0.0}};   // -[RX]x: column k = e_k x RX
      double J[2][6];
#pragma unroll
      for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int k = 0; k < 3; k++) J[r][k] = A[r][0] * G[0][k] + A[r][1] * G[1][k] + A[r][2] * G[2][k];
#pragma unroll
        for (int k = 0; k < 3; k++) J[r][3 + k] = A[r][k];
      }
      int q = 0;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = a; c < 6; c++) acc[q++] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
#pragma unroll
      for (int a = 0; a < 6; a++) acc[21 + a] += J[0][a] * e0 + J[1][a] * e1;
      acc[27] += e0 * e0 + e1 * e1;                    // (the same operations as pnp_err2: a trial pose's cost is this entry)
    });
  };
  // Gauss-Newton with step halving (oracle/pnp_oracle.py refine).  The FULL step is tried by linearising at the trial pose straight away:
  // its cost entry decides the step, and when the step is taken (the normal case) the sums ARE the next iteration's normal equations --
  // one pass, one block sum and one one-lane solve per iteration instead of two passes and two sums.  Shorter steps (rare) are tried with
  // cost-only passes as before, followed by one linearisation at the accepted pose.  Same sums in the same order either way.
  double cost = -1.0;
  {
    double acc[PNP_NRED];
    linearise(s_R, s_t, acc);
    pnp_block_sum(acc, PNP_NRED, s_red, s_sum);
    cost = s_sum[27];
  }
  for (int iter = 0; iter < 20; iter++) {
    if (tid == 0) {
      // Cholesky of the 6x6 normal matrix, solve H d = -g.  Every loop is unrolled over compile-time bounds (no early exit: a bad pivot
      // only clears the flag), so the factor lives in registers -- indexed by run-time loop counters it sat in scratch memory
      // (480 bytes per thread) and every access of this one-lane chain was a memory round trip
      double Hm[6][6], g[6], d[6];
      {
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = a; c < 6; c++) { Hm[a][c] = s_sum[q]; Hm[c][a] = s_sum[q]; q++; }
      }
#pragma unroll
      for (int a = 0; a < 6; a++) g[a] = -s_sum[21 + a];
      bool okc = true;
#pragma unroll
      for (int j = 0; j < 6; j++) {
        double sdiag = Hm[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) sdiag -= Hm[j][k] * Hm[j][k];
        if (!(sdiag > 0)) { okc = false; sdiag = 1.0; }
        const double rs = pnp_rsqrt(sdiag);           // 1 / L_jj
        Hm[j][j] = rs;                                // the diagonal holds the INVERSE pivot
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
          double sv = Hm[i][j];
#pragma unroll
          for (int k = 0; k < j; k++) sv -= Hm[i][k] * Hm[j][k];
          Hm[i][j] = sv * rs;
        }
      }
#pragma unroll
      for (int i = 0; i < 6; i++) {
        double sv = g[i];
#pragma unroll
        for (int k = 0; k < i; k++) sv -= Hm[i][k] * d[k];
        d[i] = sv * Hm[i][i];
      }
#pragma unroll
      for (int i = 5; i >= 0; i--) {
        double sv = d[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) sv -= Hm[k][i] * d[k];
        d[i] = sv * Hm[i][i];
      }
      if (okc) {
#pragma unroll
        for (int i = 0; i < 6; i++) s_d[i] = d[i];
        pnp_trial_pose(s_R, s_t, d, 1.0, s_Rn, s_tn);
      }
      s_flag = okc ? 1 : 0;
    }
    __syncthreads();
    if (!s_flag) break;
    // the full step
    double acc[PNP_NRED];
    linearise(s_Rn, s_tn, acc);
    pnp_block_sum(acc, PNP_NRED, s_red, s_sum);       // (its barriers also order the reads of s_Rn / s_tn before the next trial pose)
    double cn = s_sum[27];
    bool accepted = cn < cost, halved = false;
    double step = 0.5;
    for (int hh = 1; hh < 6 && !accepted; hh++) {      // uniform: cn and cost come out of LDS
      halved = true;
      __syncthreads();
      if (tid == 0) { double d[6]; for (int i = 0; i < 6; i++) d[i] = s_d[i]; pnp_trial_pose(s_R, s_t, d, step, s_Rn, s_tn); }
      __syncthreads();
      double cpart[1] = {0.0};
      for_points(true, [&](int, int, const double* Xw, double uo, double vo) { cpart[0] += pnp_err2(K, s_Rn, s_tn, Xw, uo, vo); });
      pnp_block_sum(cpart, 1, s_red, s_sum);
      cn = s_sum[0];
      accepted = cn < cost;
      step *= 0.5;
    }
    if (!accepted) break;
    const bool small = (cost - cn) <= 1e-12 * fmax(cost, 1e-300);
    __syncthreads();
    if (tid < 9) s_R[tid] = s_Rn[tid];
    if (tid < 3) s_t[tid] = s_tn[tid];
    cost = cn;
    __syncthreads();
    if (small) break;
    if (halved) {                                      // the sums in LDS belong to a rejected pose: linearise at the accepted one
      linearise(s_R, s_t, acc);
      pnp_block_sum(acc, PNP_NRED, s_red, s_sum);
    }
  }
  if (tid == 0) {
    double rv[3];
    pnp_log_so3(s_R, rv);
    out[0] = rv[0]; out[1] = rv[1]; out[2] = rv[2]; out[3] = s_t[0]; out[4] = s_t[1]; out[5] = s_t[2];
    out[6] = cost;
  }
  // number of inliers: block sum of the per-thread counts
  double cntv[1] = {(double)n_in};
  pnp_block_sum(cntv, 1, s_red, s_sum);
  if (tid == 0) out[7] = s_sum[0];
}

__global__ void k_pnp_init(pnp_ctrl* ctrl, int max_iters) { pnp_ctrl_reset(ctrl + blockIdx.x, max_iters); }

// ================================================================================================
// host
// ================================================================================================
void vo_pnp_destroy(vo_ctx* c) {
  if (!c->pnp) return;
  vo_pnp_ws* w = c->pnp;
  void* bufs[] = {w->d_K, w->d_X, w->d_uv, w->d_hyp, w->d_ctrl, w->d_mask, w->d_out};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (w->h_ctrl) (void)hipHostFree(w->h_ctrl);
  if (w->h_out) (void)hipHostFree(w->h_out);
  delete w;
  c->pnp = nullptr;
}

extern "C" int32_t vo_pnp_default_params(vo_pnp_params* p) {
  if (!p) return VO_E_INVALID;
  p->reproj_err = 2.0; p->confidence = 0.9999; p->max_iters = 1000000; p->seed = 0;
  return VO_OK;
}

static int32_t pnp_alloc(vo_ctx* c, int n) {
  const size_t B = c->batch;
  if (c->pnp && c->pnp->cap < n) vo_pnp_destroy(c);
  if (!c->pnp) {
    vo_pnp_ws* w = new vo_pnp_ws();
    c->pnp = w;
    w->cap = n > c->max_pts ? n : c->max_pts;
    VO_HIP(c, hipMalloc((void**)&w->d_K, sizeof(double) * 9 * B));
    VO_HIP(c, hipMalloc((void**)&w->d_X, sizeof(float) * 3 * B * w->cap));
    VO_HIP(c, hipMalloc((void**)&w->d_uv, sizeof(float) * 2 * B * w->cap));
    VO_HIP(c, hipMalloc((void**)&w->d_hyp, sizeof(pnp_hyp) * B * PNP_BATCH));
    VO_HIP(c, hipMalloc((void**)&w->d_ctrl, sizeof(pnp_ctrl) * B));
    VO_HIP(c, hipMalloc((void**)&w->d_mask, B * w->cap));
    VO_HIP(c, hipMalloc((void**)&w->d_out, sizeof(double) * 8 * B));
    VO_HIP(c, hipMemsetAsync(w->d_out, 0, sizeof(double) * 8 * B, c->stream));
    VO_HIP(c, hipMemsetAsync(w->d_ctrl, 0, sizeof(pnp_ctrl) * B, c->stream));
    VO_HIP(c, hipHostMalloc((void**)&w->h_ctrl, sizeof(pnp_ctrl) * B, hipHostMallocDefault));
    VO_HIP(c, hipHostMalloc((void**)&w->h_out, sizeof(double) * 8 * B, hipHostMallocDefault));
  }
  return VO_OK;
}

// k: index of the batch within its search (0: the small first batch); first: the control block is reset by this batch's own kernels
static void pnp_enqueue_batch(vo_ctx* c, const vo_pnp_params* prm, int k, const int32_t* counts = nullptr, int first = 0) {
  vo_pnp_ws* w = c->pnp;
  const unsigned B = (unsigned)c->batch;
  const double thr2 = prm->reproj_err * prm->reproj_err;
  const int nb = k == 0 ? PNP_FIRST_BATCH : PNP_BATCH;
  hipLaunchKernelGGL(k_pnp_solve, dim3((nb + 63) / 64, B), dim3(64), 0, c->stream, w->d_K, w->d_X, w->d_uv, w->cap, w->n, (unsigned)prm->seed,
                     w->d_hyp, w->d_ctrl, counts, first, nb);
  hipLaunchKernelGGL(k_pnp_score, dim3(nb, B), dim3(PNP_SCORE_THREADS), 0, c->stream, w->d_K, w->d_X, w->d_uv, w->cap, w->n, thr2, w->d_hyp, counts);
  hipLaunchKernelGGL(k_pnp_select, dim3(B), dim3(PNP_BATCH), 0, c->stream, w->d_hyp, w->d_ctrl, w->n, prm->confidence, prm->max_iters, counts, first, nb);
}

static int32_t pnp_enqueue_refine(vo_ctx* c, const vo_pnp_params* prm, const int32_t* counts = nullptr) {
  vo_pnp_ws* w = c->pnp;
  const size_t B = c->batch;
  const int n_most = counts ? w->cap : w->n;          // (device-side counts never exceed the capacity)
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3((unsigned)B), dim3(PNP_REFINE_THREADS), 0, c->stream, w->d_K, w->d_X, w->d_uv, w->cap, w->n,
                       prm->reproj_err * prm->reproj_err, w->d_ctrl, w->d_mask, w->d_out, counts);
  };
  if (n_most <= 4 * PNP_REFINE_THREADS) launch(k_pnp_refine<4>);
  else if (n_most <= 8 * PNP_REFINE_THREADS) launch(k_pnp_refine<8>);
  else launch(k_pnp_refine<0>);
  VO_HIP(c, hipGetLastError());
  if (counts) return VO_OK;       // closed-loop pipeline: the results are consumed on the device
  VO_HIP(c, hipMemcpyAsync(w->h_out, w->d_out, sizeof(double) * 8 * B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipMemcpyAsync(w->h_ctrl, w->d_ctrl, sizeof(pnp_ctrl) * B, hipMemcpyDeviceToHost, c->stream));
  return VO_OK;
}

// ---- closed-loop pipeline hooks: the correspondences are written by a device kernel, counts[b] per sequence ----
int32_t vo_pnp_reserve(vo_ctx* c, const double* K_host) {
  int32_t r = pnp_alloc(c, c->max_pts);
  if (r != VO_OK) return r;
  VO_HIP(c, hipMemcpyAsync(c->pnp->d_K, K_host, sizeof(double) * 9 * c->batch, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  c->pnp->n = c->pnp->cap;
  return VO_OK;
}
int32_t vo_pnp_get_view(vo_ctx* c, vo_pnp_view* v) {
  VO_CHECK(c, c->pnp, VO_E_STATE, "vo_pnp_reserve first");
  vo_pnp_ws* w = c->pnp;
  v->X = w->d_X; v->uv = w->d_uv; v->mask = w->d_mask; v->out = w->d_out; v->ctrl = reinterpret_cast<const int32_t*>(w->d_ctrl);
  v->ctrl_stride = sizeof(pnp_ctrl) / sizeof(int32_t); v->cap = w->cap;
  return VO_OK;
}
int32_t vo_pnp_enqueue_counts(vo_ctx* c, const vo_pnp_params* prm, int blind_batches, const int32_t* d_counts) {
  VO_CHECK(c, c->pnp, VO_E_STATE, "vo_pnp_reserve first");
  for (int k = 0; k < blind_batches; k++) pnp_enqueue_batch(c, prm, k, d_counts, k == 0 ? 1 : 0);     // (blind_batches >= 1: vo_pipe_create)
  return pnp_enqueue_refine(c, prm, d_counts);
}

// resident form: the correspondences stay in HBM (vo_pnp_upload), a solve is enqueued per frame without any host
// synchronisation (vo_pnp_solve_resident), the results come back with vo_pnp_fetch
extern "C" int32_t vo_pnp_upload(vo_ctx* c, const double* K, const float* pts3d, const float* pts2d, int32_t n) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, K && pts3d && pts2d, VO_E_INVALID, "null buffer");
  VO_CHECK(c, n >= 4, VO_E_INVALID, "at least 4 correspondences");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = pnp_alloc(c, n);
  if (r != VO_OK) return r;
  vo_pnp_ws* w = c->pnp;
  const size_t B = c->batch;
  const int cap = w->cap;
  VO_HIP(c, hipMemcpyAsync(w->d_K, K, sizeof(double) * 9 * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(w->d_X, sizeof(float) * 3 * cap, pts3d, sizeof(float) * 3 * n, sizeof(float) * 3 * n, B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(w->d_uv, sizeof(float) * 2 * cap, pts2d, sizeof(float) * 2 * n, sizeof(float) * 2 * n, B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  w->n = n;
  return VO_OK;
}

// Enqueues `blind_batches` batches of hypotheses -- 32, then 256 each -- (a batch exits at once for sequences that have reached their
// iteration bound) and the refinement.  Two batches cover the bound down to ~42 % inliers (0.9999 confidence); a
// sequence that would need more keeps the best pose found so far -- vo_pnp_fetch reports hypotheses < the bound through
// status VO_E_CAPACITY so that the caller can fall back to the synchronous vo_pnp_ransac.
extern "C" int32_t vo_pnp_solve_resident(vo_ctx* c, const vo_pnp_params* prm, int32_t blind_batches) {
  if (!c) return VO_E_INVALID;
  vo_pnp_params def;
  if (!prm) { vo_pnp_default_params(&def); prm = &def; }
  VO_CHECK(c, c->pnp && c->pnp->n >= 4, VO_E_STATE, "vo_pnp_upload first");
  VO_CHECK(c, prm->max_iters >= 1 && prm->reproj_err > 0 && blind_batches >= 1 && blind_batches <= 64, VO_E_INVALID, "bad parameters");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  vo_pnp_ws* w = c->pnp;
  hipLaunchKernelGGL(k_pnp_init, dim3((unsigned)c->batch), dim3(1), 0, c->stream, w->d_ctrl, prm->max_iters);
  for (int k = 0; k < blind_batches; k++) pnp_enqueue_batch(c, prm, k);
  return pnp_enqueue_refine(c, prm);
}

extern "C" int32_t vo_pnp_fetch(vo_ctx* c, double* rvec, double* tvec, uint8_t* inlier_mask, vo_pnp_stats* stats) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pnp && c->pnp->n >= 4, VO_E_STATE, "nothing to fetch");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  vo_pnp_ws* w = c->pnp;
  const size_t B = c->batch;
  if (inlier_mask) VO_HIP(c, hipMemcpy2DAsync(inlier_mask, w->n, w->d_mask, w->cap, w->n, B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t b = 0; b < B; b++) {
    const double* out = w->h_out + 8 * b;
    if (rvec) for (int k = 0; k < 3; k++) rvec[3 * b + k] = out[k];
    if (tvec) for (int k = 0; k < 3; k++) tvec[3 * b + k] = out[3 + k];
    if (stats) {
      stats[b].n_inliers = (int32_t)out[7]; stats[b].hypotheses = w->h_ctrl[b].h_done; stats[b].best = w->h_ctrl[b].best.h;
      stats[b].cost = out[6];
      stats[b].status = !(out[7] >= 4) ? VO_E_NUMERIC : (w->h_ctrl[b].done ? 0 : VO_E_CAPACITY);
    }
  }
  return VO_OK;
}

// K [batch][9], pts3d [batch][n][3] f32, pts2d [batch][n][2] f32 -> rvec, tvec [batch][3] f64 (world -> camera),
// inlier_mask [batch][n] u8, stats [batch].  Points with NaN coordinates are never inliers.
extern "C" int32_t vo_pnp_ransac(vo_ctx* c, const double* K, const float* pts3d, const float* pts2d, int32_t n, const vo_pnp_params* prm,
                                 double* rvec, double* tvec, uint8_t* inlier_mask, vo_pnp_stats* stats) {
  if (!c) return VO_E_INVALID;
  vo_pnp_params def;
  if (!prm) { vo_pnp_default_params(&def); prm = &def; }
  VO_CHECK(c, rvec && tvec, VO_E_INVALID, "null buffer");
  VO_CHECK(c, prm->max_iters >= 1 && prm->reproj_err > 0, VO_E_INVALID, "bad parameters");
  int32_t r = vo_pnp_upload(c, K, pts3d, pts2d, n);
  if (r != VO_OK) return r;
  vo_pnp_ws* w = c->pnp;
  const size_t B = c->batch;
  hipLaunchKernelGGL(k_pnp_init, dim3((unsigned)B), dim3(1), 0, c->stream, w->d_ctrl, prm->max_iters);
  // batches of 32, then 256 hypotheses per sequence until every sequence has reached its iteration bound (typically one or
  // two batches: the bound is 33 iterations at 70 % inliers, 145 at 50 %)
  for (int guard = 0; guard < (prm->max_iters + PNP_BATCH - 1) / PNP_BATCH + 1; guard++) {
    pnp_enqueue_batch(c, prm, guard);
    VO_HIP(c, hipMemcpyAsync(w->h_ctrl, w->d_ctrl, sizeof(pnp_ctrl) * B, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    bool all = true;
    for (size_t b = 0; b < B; b++) all = all && w->h_ctrl[b].done;
    if (all) break;
  }
  r = pnp_enqueue_refine(c, prm);
  if (r != VO_OK) return r;
  r = vo_pnp_fetch(c, rvec, tvec, inlier_mask, stats);
  if (r != VO_OK) return r;
  if (stats) for (size_t b = 0; b < B; b++) if (stats[b].status == VO_E_CAPACITY) stats[b].status = 0;   // max_iters reached = a valid RANSAC outcome
  return VO_OK;
}
