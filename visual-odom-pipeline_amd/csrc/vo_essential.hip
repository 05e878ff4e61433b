// 2D-2D bootstrap pose: RANSAC over five-point essential matrices + recoverPose (SURVEY.md 8f "next" row 4, the
// pose part of the bootstrap).
//
// Replaces  E, mask = cv2.findEssentialMat(p1, p2, K, prob=0.9999, method=cv2.RANSAC, threshold=1.0)
//           retval, R, t, mask = cv2.recoverPose(E, p1[inliers], p2[inliers], K)
// in Extractor.camera_pose(corr='2D-2D'), /root/reference/src/extractor/extractor.py:162-172 (called once per sequence
// by Pipeline._get_init_state, pipeline.py:63).  OpenCV's sample draws cannot be reproduced; the algorithm here is
// defined by oracle/essential_oracle.py (Nister's five-point solver, squared Sampson distance, RANSACUpdateNumIters with
// 5 model points, OpenCV's four-candidate cheirality vote) and is compared with it hypothesis by hypothesis.
//
// GPU mapping (as for the 3D-2D pose, vo_pnp.hip): a LANE per hypothesis for the minimal solve -- null space, 10 x 20
// constraint matrix, Gauss-Jordan, 10th-degree polynomial, real roots by bisection between critical points; all of it in
// per-lane scratch arrays, it runs 256 x once per call -- then a WAVE per (hypothesis, root) for the consensus count,
// one workgroup per sequence for the running best / iteration bound, and one for the cheirality vote.
// The arithmetic follows the oracle operation by operation (this library is built with -ffp-contract=off).
#include "vo_internal.h"

#include <math.h>
#include <string.h>

#define E5_BATCH 256
#define E5_MAXSOL 10

struct e5_hyp { double E[E5_MAXSOL][9]; int count[E5_MAXSOL]; int nsol; int h; };
struct e5_ctrl { int niters, h_done, done, best_h, best_k, best_count; double E[9]; };

struct vo_ess_ws {
  int cap = 0, n = 0;
  double* d_K = nullptr;       // [B][9]
  float* d_p = nullptr;        // [B][2][cap][2] pixel coordinates of the two views
  double* d_q = nullptr;       // [B][2][cap][2] normalised
  e5_hyp* d_hyp = nullptr;     // [B][E5_BATCH]
  e5_ctrl* d_ctrl = nullptr;   // [B]
  uint8_t* d_mask = nullptr;   // [B][cap]
  double* d_out = nullptr;     // [B][32]: E(9) R(9) t(3) n_inliers n_good good[4]
  e5_ctrl* h_ctrl = nullptr;   // pinned
  double* h_out = nullptr;     // pinned
};

__constant__ int c_mul11[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
__constant__ int c_mul21[10][4] = {{0, 2, 4, 5}, {3, 1, 6, 7}, {10, 13, 16, 17}, {2, 3, 8, 9}, {4, 8, 10, 11}, {8, 6, 13, 14}, {5, 9, 11, 12},
                                   {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};

__device__ __forceinline__ unsigned long long e5_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ inline void e5_sample5(unsigned seed, unsigned h, int n, int idx[5]) {
  int cnt = 0;
  unsigned k = 0;
  while (cnt < 5) {
    const unsigned long long r = (k < 256) ? e5_splitmix64(((unsigned long long)(seed & 0xFFFFFFu) << 40) ^ ((unsigned long long)h << 8) ^ (unsigned long long)(k & 0xFFu))
                                           : e5_splitmix64((unsigned long long)k);
    const int i = (int)((r >> 11) % (unsigned long long)n);
    k++;
    bool dup = false;
    for (int j = 0; j < cnt; j++) dup = dup || (idx[j] == i);
    if (!dup) idx[cnt++] = i;
  }
}

// polynomial helpers on the bases of the oracle: p1 = [x y z 1], p2 (10), p3 (20 = the matrix columns)
__device__ inline void e5_mul11(const double* a, const double* b, double* o) {
  for (int k = 0; k < 10; k++) o[k] = 0.0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) o[c_mul11[i][j]] += a[i] * b[j];
}
__device__ inline void e5_mul21(const double* a, const double* b, double* o) {
  for (int k = 0; k < 20; k++) o[k] = 0.0;
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 4; j++) o[c_mul21[i][j]] += a[i] * b[j];
}

// orthonormal null-space basis of the five epipolar constraints (oracle: null_space_5x9)
__device__ inline bool e5_null_space(const double q1[5][2], const double q2[5][2], double basis[4][9]) {
  double A[5][9];
  for (int i = 0; i < 5; i++) {
    A[i][0] = q2[i][0] * q1[i][0]; A[i][1] = q2[i][0] * q1[i][1]; A[i][2] = q2[i][0];
    A[i][3] = q2[i][1] * q1[i][0]; A[i][4] = q2[i][1] * q1[i][1]; A[i][5] = q2[i][1];
    A[i][6] = q1[i][0]; A[i][7] = q1[i][1]; A[i][8] = 1.0;
  }
  int piv[5];
  bool used[9];
  for (int c = 0; c < 9; c++) used[c] = false;
  for (int r = 0; r < 5; r++) {
    double best = 0.0; int br = -1, bc = -1;
    for (int rr = r; rr < 5; rr++)
      for (int c = 0; c < 9; c++)
        if (!used[c] && fabs(A[rr][c]) > best) { best = fabs(A[rr][c]); br = rr; bc = c; }
    if (!(best > 1e-12)) return false;
    for (int c = 0; c < 9; c++) { const double tmp = A[r][c]; A[r][c] = A[br][c]; A[br][c] = tmp; }
    piv[r] = bc; used[bc] = true;
    const double inv = 1.0 / A[r][bc];
    for (int c = 0; c < 9; c++) A[r][c] = A[r][c] * inv;
    for (int rr = 0; rr < 5; rr++) {
      if (rr == r) continue;
      const double f = A[rr][bc];
      for (int c = 0; c < 9; c++) A[rr][c] = A[rr][c] - f * A[r][c];
    }
  }
  int nb = 0;
  for (int f = 0; f < 9; f++) {
    if (used[f]) continue;
    for (int c = 0; c < 9; c++) basis[nb][c] = 0.0;
    basis[nb][f] = 1.0;
    for (int r = 0; r < 5; r++) basis[nb][piv[r]] = -A[r][f];
    nb++;
  }
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < i; j++) {
      double d = 0.0;
      for (int c = 0; c < 9; c++) d += basis[i][c] * basis[j][c];
      for (int c = 0; c < 9; c++) basis[i][c] = basis[i][c] - d * basis[j][c];
    }
    double nn = 0.0;
    for (int c = 0; c < 9; c++) nn += basis[i][c] * basis[i][c];
    nn = sqrt(nn);
    for (int c = 0; c < 9; c++) basis[i][c] = basis[i][c] / nn;
  }
  return true;
}

// 10 x 20 constraint matrix (oracle: constraint_matrix)
__device__ inline void e5_constraints(const double basis[4][9], double A[10][20]) {
  double e[3][3][4];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 4; k++) e[i][j][k] = basis[k][3 * i + j];
  double m0[10], m1[10], d[3][20];
  // det E: cofactor expansion along the first row
  const int cof[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  for (int c = 0; c < 3; c++) {
    e5_mul11(e[1][cof[c][0]], e[2][cof[c][1]], m0);
    e5_mul11(e[1][cof[c][1]], e[2][cof[c][0]], m1);
    for (int k = 0; k < 10; k++) m0[k] = m0[k] - m1[k];
    e5_mul21(m0, e[0][c], d[c]);
  }
  for (int k = 0; k < 20; k++) A[0][k] = (d[0][k] - d[1][k]) + d[2][k];
  double eet[3][3][10], tr[10];
  for (int i = 0; i < 3; i++)
    for (int j = i; j < 3; j++) {
      e5_mul11(e[i][0], e[j][0], eet[i][j]);
      e5_mul11(e[i][1], e[j][1], m0);
      for (int k = 0; k < 10; k++) eet[i][j][k] = eet[i][j][k] + m0[k];
      e5_mul11(e[i][2], e[j][2], m0);
      for (int k = 0; k < 10; k++) eet[i][j][k] = eet[i][j][k] + m0[k];
      if (j != i) for (int k = 0; k < 10; k++) eet[j][i][k] = eet[i][j][k];
    }
  for (int k = 0; k < 10; k++) tr[k] = (eet[0][0][k] + eet[1][1][k]) + eet[2][2][k];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 10; k++) eet[i][i][k] = eet[i][i][k] - 0.5 * tr[k];          // Lambda = E E^T - tr / 2 I
  double s[20];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double* row = A[1 + 3 * i + j];
      e5_mul21(eet[i][0], e[0][j], row);
      e5_mul21(eet[i][1], e[1][j], s);
      for (int k = 0; k < 20; k++) row[k] = row[k] + s[k];
      e5_mul21(eet[i][2], e[2][j], s);
      for (int k = 0; k < 20; k++) row[k] = row[k] + s[k];
    }
}

__device__ inline bool e5_gauss_jordan(double A[10][20]) {
  for (int c = 0; c < 10; c++) {
    double best = 0.0; int br = -1;
    for (int r = c; r < 10; r++)
      if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); br = r; }
    if (!(best > 1e-14)) return false;
    for (int k = 0; k < 20; k++) { const double tmp = A[c][k]; A[c][k] = A[br][k]; A[br][k] = tmp; }
    const double inv = 1.0 / A[c][c];
    for (int k = 0; k < 20; k++) A[c][k] = A[c][k] * inv;
    for (int r = 0; r < 10; r++) {
      if (r == c) continue;
      const double f = A[r][c];
      if (f != 0.0)
        for (int k = 0; k < 20; k++) A[r][k] = A[r][k] - f * A[c][k];
    }
  }
  return true;
}

__device__ inline void e5_poly_mul(const double* a, int na, const double* b, int nb, double* o) {   // ascending powers
  for (int k = 0; k < na + nb - 1; k++) o[k] = 0.0;
  for (int i = 0; i < na; i++)
    for (int j = 0; j < nb; j++) o[i + j] += a[i] * b[j];
}

__device__ inline double e5_poly_eval(const double* p, int deg, double x) {
  double s = p[deg];
  for (int k = deg - 1; k >= 0; k--) s = s * x + p[k];
  return s;
}

// real roots in ascending order (oracle: real_roots): derivative chain, sign changes between critical points, bisection
__device__ inline int e5_real_roots(const double* p, int len, double* roots) {
  double scale = 0.0;
  for (int k = 0; k < len; k++) scale = fmax(scale, fabs(p[k]));
  if (!(scale > 0)) return 0;
  int deg = len - 1;
  while (deg > 0 && fabs(p[deg]) <= 1e-14 * scale) deg--;
  if (deg == 0) return 0;
  double chain[66];                      // chain[off[k] ...]: k-th derivative, deg + 1 - k coefficients
  int off[11];
  off[0] = 0;
  for (int k = 0; k <= deg; k++) chain[k] = p[k];
  for (int k = 1; k < deg; k++) {
    off[k] = off[k - 1] + (deg + 2 - k);
    const double* q = chain + off[k - 1];
    for (int i = 1; i < deg + 2 - k; i++) chain[off[k] + i - 1] = q[i] * (double)i;
  }
  const double* lin = chain + off[deg - 1];
  int nr = 1;
  roots[0] = -lin[0] / lin[1];
  double nw[10];
  for (int k = deg - 2; k >= 0; k--) {
    const double* q = chain + off[k];
    const int d = deg - k;
    double bound = 0.0;
    for (int i = 0; i < d; i++) bound = fmax(bound, fabs(q[i] / q[d]));
    bound = 1.0 + bound;
    int nn = 0;
    for (int s = 0; s <= nr; s++) {
      const double a = (s == 0) ? -bound : roots[s - 1], b = (s == nr) ? bound : roots[s];
      const double fa = e5_poly_eval(q, d, a), fb = e5_poly_eval(q, d, b);
      if ((fa < 0) == (fb < 0)) continue;
      double lo = a, hi = b;
      for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        const double fm = e5_poly_eval(q, d, mid);
        if ((fm < 0) == (fa < 0)) lo = mid; else hi = mid;
      }
      nw[nn++] = 0.5 * (lo + hi);
    }
    nr = nn;
    for (int i = 0; i < nn; i++) roots[i] = nw[i];
  }
  return nr;
}

// Nister five-point: -> number of essential matrices (unit Frobenius norm, ascending z) written to Eout
__device__ inline int e5_five_point(const double q1[5][2], const double q2[5][2], double Eout[E5_MAXSOL][9]) {
  double basis[4][9];
  if (!e5_null_space(q1, q2, basis)) return 0;
  double A[10][20];
  e5_constraints(basis, A);
  if (!e5_gauss_jordan(A)) return 0;
  double bx[3][4], by[3][4], b1[3][5];
  for (int r = 0; r < 3; r++) {
    const double* e = &A[4 + 2 * r][10];
    const double* f = &A[5 + 2 * r][10];
    bx[r][0] = e[2]; bx[r][1] = e[1] - f[2]; bx[r][2] = e[0] - f[1]; bx[r][3] = -f[0];
    by[r][0] = e[5]; by[r][1] = e[4] - f[5]; by[r][2] = e[3] - f[4]; by[r][3] = -f[3];
    b1[r][0] = e[9]; b1[r][1] = e[8] - f[9]; b1[r][2] = e[7] - f[8]; b1[r][3] = e[6] - f[7]; b1[r][4] = -f[6];
  }
  double m[3][7], tmp[7], det[11], t11[11];
  const int pr[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  for (int c = 0; c < 3; c++) {
    e5_poly_mul(bx[pr[c][0]], 4, by[pr[c][1]], 4, m[c]);
    e5_poly_mul(by[pr[c][0]], 4, bx[pr[c][1]], 4, tmp);
    for (int k = 0; k < 7; k++) m[c][k] = m[c][k] - tmp[k];
  }
  e5_poly_mul(b1[0], 5, m[0], 7, det);
  e5_poly_mul(b1[1], 5, m[1], 7, t11);
  for (int k = 0; k < 11; k++) det[k] = det[k] - t11[k];
  e5_poly_mul(b1[2], 5, m[2], 7, t11);
  for (int k = 0; k < 11; k++) det[k] = det[k] + t11[k];
  double roots[10];
  const int nr = e5_real_roots(det, 11, roots);
  int ns = 0;
  for (int ri = 0; ri < nr; ri++) {
    const double z = roots[ri];
    double rows[3][3];
    for (int r = 0; r < 3; r++) { rows[r][0] = e5_poly_eval(bx[r], 3, z); rows[r][1] = e5_poly_eval(by[r], 3, z); rows[r][2] = e5_poly_eval(b1[r], 4, z); }
    double best = -1.0, nv[3] = {0, 0, 0};
    const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int pi = 0; pi < 3; pi++) {
      const double* ra = rows[pairs[pi][0]];
      const double* rb = rows[pairs[pi][1]];
      const double c0 = ra[1] * rb[2] - ra[2] * rb[1], c1 = ra[2] * rb[0] - ra[0] * rb[2], c2 = ra[0] * rb[1] - ra[1] * rb[0];
      const double nn = c0 * c0 + c1 * c1 + c2 * c2;
      if (nn > best) { best = nn; nv[0] = c0; nv[1] = c1; nv[2] = c2; }
    }
    if (!(best > 0) || fabs(nv[2]) <= 1e-10 * sqrt(best)) continue;
    const double x = nv[0] / nv[2], y = nv[1] / nv[2];
    double E[9], nn = 0.0;
    for (int k = 0; k < 9; k++) { E[k] = x * basis[0][k] + y * basis[1][k] + z * basis[2][k] + basis[3][k]; nn += E[k] * E[k]; }
    nn = sqrt(nn);
    if (!(nn > 0) || !(nn < __builtin_inf())) continue;
    for (int k = 0; k < 9; k++) Eout[ns][k] = E[k] / nn;
    ns++;
  }
  return ns;
}

__device__ __forceinline__ double e5_sampson(const double* E, double x1, double y1, double x2, double y2) {
  const double a0 = E[0] * x1 + E[1] * y1 + E[2], a1 = E[3] * x1 + E[4] * y1 + E[5], a2 = E[6] * x1 + E[7] * y1 + E[8];   // E x1
  const double b0 = E[0] * x2 + E[3] * y2 + E[6], b1 = E[1] * x2 + E[4] * y2 + E[7];                                       // E^T x2
  const double num = x2 * a0 + y2 * a1 + a2;
  return num * num / (a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1);
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_e5_prep(const double* __restrict__ Kall, const float* __restrict__ p, double* __restrict__ q, int cap, int n) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* K = Kall + 9 * b;
  for (int v = 0; v < 2; v++) {
    const size_t o = (((size_t)b * 2 + v) * cap + i) * 2;
    q[o] = ((double)p[o] - K[2]) / K[0];
    q[o + 1] = ((double)p[o + 1] - K[5]) / K[4];
  }
}

__global__ void k_e5_init(e5_ctrl* ctrl, int max_iters) {
  e5_ctrl* c = ctrl + blockIdx.x;
  c->niters = max_iters; c->h_done = 0; c->done = 0; c->best_h = -1; c->best_k = -1; c->best_count = 4;
  for (int i = 0; i < 9; i++) c->E[i] = 0;
}

__global__ void __launch_bounds__(64) k_e5_solve(const double* __restrict__ q, int cap, int n, unsigned seed, e5_hyp* __restrict__ hyps,
                                                 const e5_ctrl* __restrict__ ctrl) {
  const int b = blockIdx.y, slot = blockIdx.x * 64 + threadIdx.x;
  const e5_ctrl cs = ctrl[b];
  e5_hyp* out = hyps + (size_t)b * E5_BATCH + slot;
  const int h = cs.h_done + slot;
  out->h = h;
  if (cs.done) { out->nsol = 0; return; }
  const double* qa = q + ((size_t)b * 2) * cap * 2;
  const double* qb = qa + (size_t)cap * 2;
  int idx[5];
  e5_sample5(seed, (unsigned)h, n, idx);
  double q1[5][2], q2[5][2];
  for (int i = 0; i < 5; i++) {
    q1[i][0] = qa[2 * idx[i]]; q1[i][1] = qa[2 * idx[i] + 1];
    q2[i][0] = qb[2 * idx[i]]; q2[i][1] = qb[2 * idx[i] + 1];
  }
  double E[E5_MAXSOL][9];
  const int ns = e5_five_point(q1, q2, E);
  out->nsol = ns;
  for (int k = 0; k < ns; k++)
    for (int i = 0; i < 9; i++) out->E[k][i] = E[k][i];
}

// grid (E5_BATCH, batch), E5_MAXSOL waves: wave k counts the consensus of root k of the hypothesis
__global__ void __launch_bounds__(64 * E5_MAXSOL) k_e5_score(const double* __restrict__ q, int cap, int n, const double* __restrict__ thr2_all,
                                                              e5_hyp* __restrict__ hyps) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  e5_hyp* hp = hyps + (size_t)b * E5_BATCH + blockIdx.x;
  if (k >= hp->nsol) { if (lane == 0) hp->count[k] = 0; return; }
  const double* qa = q + ((size_t)b * 2) * cap * 2;
  const double* qb = qa + (size_t)cap * 2;
  const double thr2 = thr2_all[b];
  double E[9];
  for (int i = 0; i < 9; i++) E[i] = hp->E[k][i];
  int cnt = 0;
  for (int i = lane; i < n; i += 64)
    cnt += (e5_sampson(E, qa[2 * i], qa[2 * i + 1], qb[2 * i], qb[2 * i + 1]) <= thr2) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) hp->count[k] = cnt;
}

__device__ inline int e5_update_iters(double p, double ep, int model_points, int max_iters) {   // OpenCV RANSACUpdateNumIters
  p = fmin(fmax(p, 0.0), 1.0); ep = fmin(fmax(ep, 0.0), 1.0);
  double num = fmax(1.0 - p, 2.2250738585072014e-308);
  double denom = 1.0 - pow(1.0 - ep, (double)model_points);
  if (denom < 2.2250738585072014e-308) return 0;
  num = log(num); denom = log(denom);
  return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)rint(num / denom);
}

// grid (batch): running best (most inliers; ties to the smallest h, then the smallest root) and the iteration bound
__global__ void __launch_bounds__(E5_BATCH) k_e5_select(const e5_hyp* __restrict__ hyps, e5_ctrl* __restrict__ ctrl, int n, double prob, int max_iters) {
  __shared__ int s_cnt[E5_BATCH];
  __shared__ int s_k[E5_BATCH];
  const int b = blockIdx.x, tid = threadIdx.x;
  e5_ctrl* c = ctrl + b;
  if (c->done) return;
  const e5_hyp* H = hyps + (size_t)b * E5_BATCH;
  {
    int bc = -1, bk = 0;
    for (int k = 0; k < H[tid].nsol; k++) if (H[tid].count[k] > bc) { bc = H[tid].count[k]; bk = k; }
    s_cnt[tid] = bc; s_k[tid] = bk;
  }
  __syncthreads();
  if (tid == 0) {
    int bi = -1, bc = c->best_count;
    for (int i = 0; i < E5_BATCH; i++) if (s_cnt[i] > bc) { bc = s_cnt[i]; bi = i; }     // batch order = ascending h
    if (bi >= 0) {
      c->best_count = bc; c->best_h = H[bi].h; c->best_k = s_k[bi];
      for (int i = 0; i < 9; i++) c->E[i] = H[bi].E[s_k[bi]][i];
    }
    c->h_done += E5_BATCH;
    if (c->best_h >= 0) {
      const int ni = e5_update_iters(prob, (double)(n - c->best_count) / (double)n, 5, max_iters);
      if (ni < c->niters) c->niters = ni;
    }
    c->done = (c->h_done >= c->niters) ? 1 : 0;
  }
}

// DLT triangulation of one normalised correspondence against [I | 0] and [R | t] (OpenCV cvTriangulatePoints: null
// vector of the 4 x 4 system by one-sided Jacobi, as in k_dlt)
__device__ inline void e5_triangulate(const double* R, const double* t, double x1, double y1, double x2, double y2, double* Q) {
  double U[4][4], V[4][4];
  U[0][0] = -1.0; U[0][1] = 0.0; U[0][2] = x1; U[0][3] = 0.0;
  U[1][0] = 0.0; U[1][1] = -1.0; U[1][2] = y1; U[1][3] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) { U[2][k] = x2 * R[6 + k] - R[k]; U[3][k] = y2 * R[6 + k] - R[3 + k]; }
  U[2][3] = x2 * t[2] - t[0]; U[3][3] = y2 * t[2] - t[1];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) V[r][k] = (r == k) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool changed = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { al += U[k][p] * U[k][p]; be += U[k][q] * U[k][q]; ga += U[k][p] * U[k][q]; }
        if (ga * ga > 4.930380657631324e-32 * (al * be)) {
          changed = true;
          const double zeta = (be - al) / (2.0 * ga);
          const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const double up = U[k][p], uq = U[k][q];
            U[k][p] = c * up - s * uq; U[k][q] = s * up + c * uq;
            const double vp = V[k][p], vq = V[k][q];
            V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
          }
        }
      }
    if (!changed) break;
  }
  double bn = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double nn = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) nn += U[k][j] * U[k][j];
    if (j == 0 || nn < bn) {
      bn = nn;
#pragma unroll
      for (int k = 0; k < 4; k++) Q[k] = V[k][j];
    }
  }
}

// E = U diag(s, s, 0) V^T by one-sided Jacobi on the columns; u2 = u0 x u1, v2 = v0 x v1 (det U = det V = +1)
// -> R1 = U W V^T, R2 = U W^T V^T, t = u2 (OpenCV decomposeEssentialMat up to the sign conventions of its SVD)
__device__ inline void e5_decompose(const double* E, double* R1, double* R2, double* t) {
  double U[3][3], V[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U[i][j] = E[3 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool changed = false;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; k++) { al += U[k][p] * U[k][p]; be += U[k][q] * U[k][q]; ga += U[k][p] * U[k][q]; }
        if (ga * ga > 4.930380657631324e-32 * (al * be)) {
          changed = true;
          const double zeta = (be - al) / (2.0 * ga);
          const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
          for (int k = 0; k < 3; k++) {
            const double up = U[k][p], uq = U[k][q];
            U[k][p] = c * up - s * uq; U[k][q] = s * up + c * uq;
            const double vp = V[k][p], vq = V[k][q];
            V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
          }
        }
      }
    if (!changed) break;
  }
  double nrm[3];
  for (int j = 0; j < 3; j++) nrm[j] = U[0][j] * U[0][j] + U[1][j] * U[1][j] + U[2][j] * U[2][j];
  int jmin = 0;
  if (nrm[1] < nrm[jmin]) jmin = 1;
  if (nrm[2] < nrm[jmin]) jmin = 2;
  const int j0 = (jmin == 0) ? 1 : 0, j1 = (jmin == 2) ? 1 : 2;
  double u0[3], u1[3], u2[3], v0[3], v1[3], v2[3];
  const double n0 = sqrt(nrm[j0]), n1 = sqrt(nrm[j1]);
  for (int k = 0; k < 3; k++) { u0[k] = U[k][j0] / n0; u1[k] = U[k][j1] / n1; v0[k] = V[k][j0]; v1[k] = V[k][j1]; }
  u2[0] = u0[1] * u1[2] - u0[2] * u1[1]; u2[1] = u0[2] * u1[0] - u0[0] * u1[2]; u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
  v2[0] = v0[1] * v1[2] - v0[2] * v1[1]; v2[1] = v0[2] * v1[0] - v0[0] * v1[2]; v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
  { const double nn = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]); for (int k = 0; k < 3; k++) u2[k] /= nn; }
  { const double nn = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]); for (int k = 0; k < 3; k++) v2[k] /= nn; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double a = u0[i] * v1[j] - u1[i] * v0[j], c = u2[i] * v2[j];
      R1[3 * i + j] = a + c; R2[3 * i + j] = -a + c;
    }
  for (int k = 0; k < 3; k++) t[k] = u2[k];
}

// grid (batch), 256 threads: consensus mask of the best model, the four pose candidates, cheirality vote
__global__ void __launch_bounds__(256) k_e5_finish(const double* __restrict__ q, int cap, int n, const double* __restrict__ thr2_all, double dist,
                                                   const e5_ctrl* __restrict__ ctrl, uint8_t* __restrict__ mask_all, double* __restrict__ out_all) {
  __shared__ double s_R[2][9], s_t[3];
  __shared__ int s_cnt[4][5];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* qa = q + ((size_t)b * 2) * cap * 2;
  const double* qb = qa + (size_t)cap * 2;
  uint8_t* mask = mask_all + (size_t)b * cap;
  double* out = out_all + 32 * b;
  const e5_ctrl cs = ctrl[b];
  if (cs.best_h < 0) {
    for (int i = tid; i < n; i += 256) mask[i] = 0;
    if (tid < 32) out[tid] = (tid < 21) ? __builtin_nan("") : 0.0;
    return;
  }
  const double thr2 = thr2_all[b];
  if (tid == 0) e5_decompose(cs.E, s_R[0], s_R[1], s_t);
  __syncthreads();
  int cnt[5] = {0, 0, 0, 0, 0};
  for (int i = tid; i < n; i += 256) {
    const double x1 = qa[2 * i], y1 = qa[2 * i + 1], x2 = qb[2 * i], y2 = qb[2 * i + 1];
    const uint8_t m = (e5_sampson(cs.E, x1, y1, x2, y2) <= thr2) ? 1 : 0;
    mask[i] = m;
    if (!m) continue;
    cnt[4]++;
    for (int k = 0; k < 4; k++) {
      const double* R = s_R[k & 1];
      const double sg = (k & 2) ? -1.0 : 1.0;
      const double t[3] = {sg * s_t[0], sg * s_t[1], sg * s_t[2]};
      double Q[4];
      e5_triangulate(R, t, x1, y1, x2, y2, Q);
      if (!(Q[2] * Q[3] > 0)) continue;
      const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
      if (!(Z < dist)) continue;
      const double z2 = (R[6] * X + R[7] * Y + R[8] * Z) + t[2];
      cnt[k] += (z2 > 0 && z2 < dist) ? 1 : 0;
    }
  }
  for (int k = 0; k < 5; k++) {
    int v = cnt[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_cnt[wave][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    int g[5];
    for (int k = 0; k < 5; k++) g[k] = s_cnt[0][k] + s_cnt[1][k] + s_cnt[2][k] + s_cnt[3][k];
    int k;
    if (g[0] >= g[1] && g[0] >= g[2] && g[0] >= g[3]) k = 0;
    else if (g[1] >= g[0] && g[1] >= g[2] && g[1] >= g[3]) k = 1;
    else if (g[2] >= g[0] && g[2] >= g[1] && g[2] >= g[3]) k = 2;
    else k = 3;
    for (int i = 0; i < 9; i++) { out[i] = cs.E[i]; out[9 + i] = s_R[k & 1][i]; }
    for (int i = 0; i < 3; i++) out[18 + i] = ((k & 2) ? -1.0 : 1.0) * s_t[i];
    out[21] = (double)g[4]; out[22] = (double)g[k];
    for (int i = 0; i < 4; i++) out[23 + i] = (double)g[i];
  }
}

// ================================================================================================
// host
// ================================================================================================
void vo_ess_destroy(vo_ctx* c) {
  if (!c->ess) return;
  vo_ess_ws* w = c->ess;
  void* bufs[] = {w->d_K, w->d_p, w->d_q, w->d_hyp, w->d_ctrl, w->d_mask, w->d_out};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (w->h_ctrl) (void)hipHostFree(w->h_ctrl);
  if (w->h_out) (void)hipHostFree(w->h_out);
  delete w;
  c->ess = nullptr;
}

extern "C" int32_t vo_essential_default_params(vo_ess_params* p) {
  if (!p) return VO_E_INVALID;
  p->threshold = 1.0; p->prob = 0.9999; p->distance_thresh = 50.0; p->max_iters = 1000; p->seed = 0;
  return VO_OK;
}

static int32_t ess_alloc(vo_ctx* c, int n) {
  const size_t B = c->batch;
  if (c->ess && c->ess->cap < n) vo_ess_destroy(c);
  if (!c->ess) {
    vo_ess_ws* w = new vo_ess_ws();
    c->ess = w;
    w->cap = n > c->max_pts ? n : c->max_pts;
    VO_HIP(c, hipMalloc((void**)&w->d_K, sizeof(double) * (9 + 1) * B));           // [B][9] K, then [B] squared thresholds
    VO_HIP(c, hipMalloc((void**)&w->d_p, sizeof(float) * 4 * B * w->cap));
    VO_HIP(c, hipMalloc((void**)&w->d_q, sizeof(double) * 4 * B * w->cap));
    VO_HIP(c, hipMalloc((void**)&w->d_hyp, sizeof(e5_hyp) * B * E5_BATCH));
    VO_HIP(c, hipMalloc((void**)&w->d_ctrl, sizeof(e5_ctrl) * B));
    VO_HIP(c, hipMalloc((void**)&w->d_mask, B * w->cap));
    VO_HIP(c, hipMalloc((void**)&w->d_out, sizeof(double) * 32 * B));
    VO_HIP(c, hipHostMalloc((void**)&w->h_ctrl, sizeof(e5_ctrl) * B, hipHostMallocDefault));
    VO_HIP(c, hipHostMalloc((void**)&w->h_out, sizeof(double) * (32 + 10) * B, hipHostMallocDefault));
  }
  return VO_OK;
}

// K [batch][9], pts1 / pts2 [batch][n][2] f32 (pixels, view 1 / view 2) -> E [batch][9] (unit Frobenius norm),
// R [batch][9], t [batch][3] (|t| = 1; x2 ~ R x1 + t), inlier_mask [batch][n] u8, stats [batch].
// Points with NaN coordinates are never inliers.
extern "C" int32_t vo_essential_ransac(vo_ctx* c, const double* K, const float* pts1, const float* pts2, int32_t n, const vo_ess_params* prm,
                                       double* E, double* R, double* t, uint8_t* inlier_mask, vo_ess_stats* stats) {
  if (!c) return VO_E_INVALID;
  vo_ess_params def;
  if (!prm) { vo_essential_default_params(&def); prm = &def; }
  VO_CHECK(c, K && pts1 && pts2 && R && t, VO_E_INVALID, "null buffer");
  VO_CHECK(c, n >= 5, VO_E_INVALID, "at least 5 correspondences");
  VO_CHECK(c, prm->max_iters >= 1 && prm->threshold > 0 && prm->distance_thresh > 0, VO_E_INVALID, "bad parameters");
  VO_HIP(c, hipSetDevice(c->device));
  int32_t r = ess_alloc(c, n);
  if (r != VO_OK) return r;
  vo_ess_ws* w = c->ess;
  const size_t B = c->batch;
  const int cap = w->cap;
  w->n = n;
  double* h_K = w->h_out + 32 * B;                       // staging for K and the squared normalised thresholds
  double* d_thr2 = w->d_K + 9 * B;
  for (size_t b = 0; b < B; b++) {
    const double* Kb = K + 9 * b;
    VO_CHECK(c, Kb[0] != 0.0 && Kb[4] != 0.0, VO_E_INVALID, "K has a zero focal length");
    const double tn = prm->threshold / ((Kb[0] + Kb[4]) / 2.0);
    h_K[b] = tn * tn;
  }
  VO_HIP(c, hipMemcpyAsync(w->d_K, K, sizeof(double) * 9 * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpyAsync(d_thr2, h_K, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
  const size_t row = sizeof(float) * 2 * n, pitch = sizeof(float) * 2 * 2 * (size_t)cap;
  VO_HIP(c, hipMemcpy2DAsync(w->d_p, pitch, pts1, row, row, B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(w->d_p + (size_t)cap * 2, pitch, pts2, row, row, B, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_e5_prep, dim3(vo_div_up(n, 256), (unsigned)B), dim3(256), 0, c->stream, w->d_K, w->d_p, w->d_q, cap, n);
  hipLaunchKernelGGL(k_e5_init, dim3((unsigned)B), dim3(1), 0, c->stream, w->d_ctrl, prm->max_iters);
  for (int guard = 0; guard < (prm->max_iters + E5_BATCH - 1) / E5_BATCH; guard++) {
    hipLaunchKernelGGL(k_e5_solve, dim3(E5_BATCH / 64, (unsigned)B), dim3(64), 0, c->stream, w->d_q, cap, n, (unsigned)prm->seed, w->d_hyp, w->d_ctrl);
    hipLaunchKernelGGL(k_e5_score, dim3(E5_BATCH, (unsigned)B), dim3(64 * E5_MAXSOL), 0, c->stream, w->d_q, cap, n, d_thr2, w->d_hyp);
    hipLaunchKernelGGL(k_e5_select, dim3((unsigned)B), dim3(E5_BATCH), 0, c->stream, w->d_hyp, w->d_ctrl, n, prm->prob, prm->max_iters);
    VO_HIP(c, hipGetLastError());
    VO_HIP(c, hipMemcpyAsync(w->h_ctrl, w->d_ctrl, sizeof(e5_ctrl) * B, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    bool all = true;
    for (size_t b = 0; b < B; b++) all = all && w->h_ctrl[b].done;
    if (all) break;
  }
  hipLaunchKernelGGL(k_e5_finish, dim3((unsigned)B), dim3(256), 0, c->stream, w->d_q, cap, n, d_thr2, prm->distance_thresh, w->d_ctrl, w->d_mask, w->d_out);
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipMemcpyAsync(w->h_out, w->d_out, sizeof(double) * 32 * B, hipMemcpyDeviceToHost, c->stream));
  if (inlier_mask) VO_HIP(c, hipMemcpy2DAsync(inlier_mask, n, w->d_mask, cap, n, B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t b = 0; b < B; b++) {
    const double* o = w->h_out + 32 * b;
    if (E) for (int k = 0; k < 9; k++) E[9 * b + k] = o[k];
    for (int k = 0; k < 9; k++) R[9 * b + k] = o[9 + k];
    for (int k = 0; k < 3; k++) t[3 * b + k] = o[18 + k];
    if (stats) {
      stats[b].n_inliers = (int32_t)o[21]; stats[b].n_good = (int32_t)o[22];
      stats[b].hypotheses = w->h_ctrl[b].h_done; stats[b].best = w->h_ctrl[b].best_h;
      stats[b].status = (w->h_ctrl[b].best_h < 0) ? VO_E_NUMERIC : 0;
      stats[b].pad = 0;
    }
  }
  return VO_OK;
}
