// Two-view DLT triangulation, one thread per point, float64 one-sided Jacobi SVD in registers.
//
// Replaces cv2.triangulatePoints(P0, P1, uv0, uv1) at /root/reference/src/extractor/extractor.py:270
// (OpenCV 4.4 calib3d/triangulate.cpp: per point a 4x4 double matrix A with rows x*P[2]-P[0],
// y*P[2]-P[1] for both views, SVD, last right-singular vector) and fuses the
// statistics TriangulatorNL.refine filters on (/root/reference/src/extractor/triangulate.py:87-111):
// camera-1 depth and the mean reprojection error (|e0| + |e1|) / 2, evaluated in float64 on the
// float32-rounded, float32-dehomogenised point exactly like extractor.py:271 / triangulate.py:15-29.
#include "vo_internal.h"

// 1/sqrt(x) and 1/x from the hardware estimates + two Newton steps (relative error ~1e-16): the IEEE-exact sqrt / divide
// expansions are ~30 dependent f64 instructions each and a Jacobi rotation needs three of each; the kernel is one long
// dependent chain per point (500 waves on 1024 SIMDs), so their latency IS the run time.  Parity here is by tolerance
// (1e-4 relative on X), not bit-exact.
__device__ __forceinline__ double dlt_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}
__device__ __forceinline__ double dlt_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  return fma(fma(-x, y, 1.0), y, y);
}

// grid (ceil(n / 64), batch)
__global__ void __launch_bounds__(64) k_dlt(const vo_dlt_cam* __restrict__ cams, int n, int want_stats, size_t uv_seq,
                                            size_t slab_seq, const float* __restrict__ uv0, const float* __restrict__ uv1,
                                            float* __restrict__ X4, double* __restrict__ depth1, double* __restrict__ reproj,
                                            const int32_t* __restrict__ counts, const int32_t* __restrict__ cam_sel, int cams_per_seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int bseq = blockIdx.y;
  if (counts && i >= counts[bseq]) return;
  // closed-loop pipeline: a sequence triangulates groups of tracks born at different frames in one launch, each group against its
  // own first camera (extractor.py:210-220): cam_sel[i] picks the pair of point i
  const vo_dlt_cam& a = cam_sel ? cams[(size_t)bseq * cams_per_seq + cam_sel[(size_t)bseq * n + i]] : cams[bseq];
  uv0 += (size_t)bseq * uv_seq; uv1 += (size_t)bseq * uv_seq;
  X4 = vo_seq(X4, slab_seq, bseq); depth1 = vo_seq(depth1, slab_seq, bseq); reproj = vo_seq(reproj, slab_seq, bseq);
  const float u0 = uv0[2 * i], v0 = uv0[2 * i + 1], u1 = uv1[2 * i], v1 = uv1[2 * i + 1];
  double U[4][4], V[4][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    U[0][k] = (double)u0 * (double)a.P0[8 + k] - (double)a.P0[k];
    U[1][k] = (double)v0 * (double)a.P0[8 + k] - (double)a.P0[4 + k];
    U[2][k] = (double)u1 * (double)a.P1[8 + k] - (double)a.P1[k];
    U[3][k] = (double)v1 * (double)a.P1[8 + k] - (double)a.P1[4 + k];
  }
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
        // |ga| > eps sqrt(al be)  <=>  ga^2 > eps^2 al be  (no square root in the test)
        if (ga * ga > 4.930380657631324e-32 * (al * be)) {
          changed = true;
          const double zeta = (be - al) * (0.5 * dlt_rcp(ga));
          const double z2 = 1.0 + zeta * zeta;
          const double t = (zeta >= 0 ? 1.0 : -1.0) * dlt_rcp(fabs(zeta) + z2 * dlt_rsqrt(z2));
          const double c = dlt_rsqrt(1.0 + t * t), s = c * t;
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
  // smallest singular value <-> column of U with the smallest norm
  double bn = 0; double x[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double nn = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) nn += U[k][j] * U[k][j];
    if (j == 0 || nn < bn) {
      bn = nn;
#pragma unroll
      for (int k = 0; k < 4; k++) x[k] = V[k][j];
    }
  }
  float xf[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { xf[k] = (float)x[k]; X4[(size_t)k * n + i] = xf[k]; }

  if (want_stats) {
    // float32 dehomogenisation (numpy float32 divide), then float64 statistics
    const double X = (double)(xf[0] / xf[3]), Y = (double)(xf[1] / xf[3]), Z = (double)(xf[2] / xf[3]);
    depth1[i] = a.H1z[0] * X + a.H1z[1] * Y + a.H1z[2] * Z + a.H1z[3];
    double e[2];
#pragma unroll
    for (int view = 0; view < 2; view++) {
      const double* M = view ? a.M1 : a.M0;
      const double px = M[0] * X + M[1] * Y + M[2] * Z + M[3];
      const double py = M[4] * X + M[5] * Y + M[6] * Z + M[7];
      const double pz = M[8] * X + M[9] * Y + M[10] * Z + M[11];
      const double du = (double)(view ? u1 : u0) - px / pz, dv = (double)(view ? v1 : v0) - py / pz;
      e[view] = sqrt(du * du + dv * dv);
    }
    reproj[i] = (e[0] + e[1]) / 2;
  }
}

// host: per-sequence camera data from [batch] x (P0, P1[, K, H0, H1])
static int32_t dlt_set_cams(vo_ctx* c, const float* P0, const float* P1, const double* K, const double* H0, const double* H1) {
  const int B = c->batch;
  std::vector<vo_dlt_cam> cams((size_t)B);
  for (int b = 0; b < B; b++) {
    vo_dlt_cam& a = cams[b];
    for (int k = 0; k < 12; k++) { a.P0[k] = P0[12 * b + k]; a.P1[k] = P1[12 * b + k]; a.M0[k] = 0; a.M1[k] = 0; }
    for (int k = 0; k < 4; k++) a.H1z[k] = 0;
    if (K) {
      const double* Kb = K + 9 * b; const double* H0b = H0 + 16 * b; const double* H1b = H1 + 16 * b;
      for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++) {
          double s0 = 0, s1 = 0;
          for (int k = 0; k < 3; k++) { s0 += Kb[r * 3 + k] * H0b[k * 4 + col]; s1 += Kb[r * 3 + k] * H1b[k * 4 + col]; }
          a.M0[r * 4 + col] = s0; a.M1[r * 4 + col] = s1;
        }
      for (int k = 0; k < 4; k++) a.H1z[k] = H1b[8 + k];
    }
  }
  if (!c->d_dlt_cam) VO_HIP(c, hipMalloc((void**)&c->d_dlt_cam, sizeof(vo_dlt_cam) * B));
  VO_HIP(c, hipMemcpy(c->d_dlt_cam, cams.data(), sizeof(vo_dlt_cam) * B, hipMemcpyHostToDevice));   // synchronous: `cams` is a local
  c->dlt_stats = K ? 1 : 0;
  return VO_OK;
}

static int32_t dlt_launch(vo_ctx* c, int n) {
  vo_prof_scope prof(c, VO_PROF_DLT);
  hipLaunchKernelGGL(k_dlt, dim3(vo_div_up(n, 64), c->batch), dim3(64), 0, c->stream, c->d_dlt_cam, n, c->dlt_stats,
                     (size_t)c->max_pts * 2, c->slab_seq, c->d_uv0, c->d_uv1, vo_slab<float>(c, c->off_X4),
                     vo_slab<double>(c, c->off_depth), vo_slab<double>(c, c->off_reproj), nullptr, nullptr, 1);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

// closed-loop pipeline: cameras (cams_per_seq per sequence), their per-point selection and the pixel pairs were written by device
// kernels; counts[b] pairs are valid; X4 keeps the row stride n_hi
int32_t vo_dlt_enqueue_counts(vo_ctx* c, int n_hi, const int32_t* d_counts, const vo_dlt_cam* d_cams, const int32_t* d_cam_sel, int cams_per_seq) {
  vo_prof_scope prof(c, VO_PROF_DLT);
  hipLaunchKernelGGL(k_dlt, dim3(vo_div_up(n_hi, 64), c->batch), dim3(64), 0, c->stream, d_cams, n_hi, 1,
                     (size_t)c->max_pts * 2, c->slab_seq, c->d_uv0, c->d_uv1, vo_slab<float>(c, c->off_X4),
                     vo_slab<double>(c, c->off_depth), vo_slab<double>(c, c->off_reproj), d_counts, d_cam_sel, cams_per_seq);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

static int32_t dlt_upload_uv(vo_ctx* c, const float* uv0, const float* uv1, int n) {
  const size_t row = sizeof(float) * 2 * n, pitch = sizeof(float) * 2 * (size_t)c->max_pts;
  VO_HIP(c, hipMemcpy2DAsync(c->d_uv0, pitch, uv0, row, row, c->batch, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(c->d_uv1, pitch, uv1, row, row, c->batch, hipMemcpyHostToDevice, c->stream));
  return VO_OK;
}

static int32_t dlt_download(vo_ctx* c, int n, float* X4, double* depth1, double* reproj) {
  if (X4) VO_HIP(c, hipMemcpy2DAsync(X4, sizeof(float) * 4 * n, c->d_slab + c->off_X4, c->slab_seq, sizeof(float) * 4 * n, c->batch, hipMemcpyDeviceToHost, c->stream));
  if (depth1 && c->dlt_stats) VO_HIP(c, hipMemcpy2DAsync(depth1, sizeof(double) * n, c->d_slab + c->off_depth, c->slab_seq, sizeof(double) * n, c->batch, hipMemcpyDeviceToHost, c->stream));
  if (reproj && c->dlt_stats) VO_HIP(c, hipMemcpy2DAsync(reproj, sizeof(double) * n, c->d_slab + c->off_reproj, c->slab_seq, sizeof(double) * n, c->batch, hipMemcpyDeviceToHost, c->stream));
  return VO_OK;
}

// All arrays carry a leading batch dimension: P0/P1 [batch][12], uv [batch][n][2], X4 [batch][4][n],
// K [batch][9], H0/H1 [batch][16], depth1/reproj [batch][n].
extern "C" int32_t vo_triangulate_dlt(vo_ctx* c, const float* P0, const float* P1, const float* uv0,
                                      const float* uv1, int32_t n, float* X4, const double* K,
                                      const double* H0, const double* H1, double* depth1, double* reproj) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, n >= 0 && n <= c->max_pts, VO_E_CAPACITY, "n exceeds max_pts");
  if (n == 0) return VO_OK;
  VO_CHECK(c, P0 && P1 && uv0 && uv1 && X4, VO_E_INVALID, "null buffer");
  if (K) VO_CHECK(c, H0 && H1 && depth1 && reproj, VO_E_INVALID, "statistics need K, H0, H1, depth1, reproj");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = dlt_set_cams(c, P0, P1, K, H0, H1);
  if (r != VO_OK) return r;
  r = dlt_upload_uv(c, uv0, uv1, n);
  if (r != VO_OK) return r;
  r = dlt_launch(c, n);
  if (r != VO_OK) return r;
  r = dlt_download(c, n, X4, depth1, reproj);
  if (r != VO_OK) return r;
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

extern "C" int32_t vo_dlt_upload(vo_ctx* c, const float* P0, const float* P1, const float* uv0, const float* uv1,
                                 int32_t n, const double* K, const double* H0, const double* H1) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, n >= 1 && n <= c->max_pts, VO_E_CAPACITY, "n exceeds max_pts");
  VO_CHECK(c, P0 && P1 && uv0 && uv1, VO_E_INVALID, "null buffer");
  if (K) VO_CHECK(c, H0 && H1, VO_E_INVALID, "statistics need K, H0, H1");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = dlt_set_cams(c, P0, P1, K, H0, H1);
  if (r != VO_OK) return r;
  r = dlt_upload_uv(c, uv0, uv1, n);
  if (r != VO_OK) return r;
  VO_HIP(c, hipStreamSynchronize(c->stream));
  c->dlt_n = n;
  return VO_OK;
}

extern "C" int32_t vo_dlt_resident(vo_ctx* c) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->dlt_n > 0, VO_E_STATE, "vo_dlt_upload first");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  return dlt_launch(c, c->dlt_n);
}

extern "C" int32_t vo_dlt_fetch(vo_ctx* c, float* X4, double* depth1, double* reproj) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->dlt_n > 0, VO_E_STATE, "nothing to fetch");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = dlt_download(c, c->dlt_n, X4, depth1, reproj);
  if (r != VO_OK) return r;
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}
