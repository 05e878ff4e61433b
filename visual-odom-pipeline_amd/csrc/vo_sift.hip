// SIFT detector / descriptor for the bootstrap (SURVEY.md 8f "next" row 4, feature part).
//
// Replaces  self._features = cv2.SIFT_create(nfeatures=1000);  cv_kp = self._features.detect(img, mask=mask);
//           cv_kp, desc = self._features.compute(img, cv_kp)
// in Extractor.extract(detector='custom', describe=True), /root/reference/src/extractor/extractor.py:26-28, 114-122
// (called for the two bootstrap frames by Pipeline._get_init_state, pipeline.py:48-49).  OpenCV 4.4's SIFT
// (features2d/src/sift.dispatch.cpp, sift.simd.hpp) is restated in oracle/sift_oracle.py, which defines the float32
// operation order this file follows operation by operation (the library is built with -ffp-contract=off; exp / cos / sin /
// pow go through float64 and are rounded once, atan2 is OpenCV's fastAtan2 polynomial): the tests compare keypoints and
// descriptors with the oracle exactly.
//
// GPU mapping: the scale space is bandwidth-bound image work (a thread per pixel: 2x bilinear upsample, separable Gaussian
// rows / columns with REFLECT_101, DoG, nearest decimation, 26-neighbour extremum test appending candidates); the per-
// candidate work (quadratic fit, 36-bin orientation histogram) runs a LANE per candidate, the 4 x 4 x 8 descriptor a WAVE
// per keypoint (samples evaluated 64 at a time); in both the histograms are accumulated SEQUENTIALLY in sample order --
// that order is part of the float32 result.  Sorting / duplicate removal /
// retainBest of the few thousand raw keypoints is host code (std::sort), as in OpenCV.  Once per sequence: ~10 ms.
#include "vo_internal.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define SIFT_MAX_OCT 14
#define SIFT_LAYERS 3
#define SIFT_BORDER 5
#define SIFT_CAND_CAP (1 << 19)
#define SIFT_RAW_CAP (1 << 17)
#define SIFT_MAX_TAPS 32

struct sift_geom {
  int n_oct;
  int w[SIFT_MAX_OCT], h[SIFT_MAX_OCT];
  size_t goff[SIFT_MAX_OCT], doff[SIFT_MAX_OCT];     // floats, within one sequence
  size_t g_seq, d_seq;
};

struct vo_sift_ws {
  sift_geom G;
  uint8_t* d_img = nullptr;      // [B][H][W]
  float* d_a = nullptr;          // [B][2H][2W] scratch (upsampled image / row pass)
  float* d_b = nullptr;
  float* d_gauss = nullptr; float* d_dog = nullptr;
  float* d_taps = nullptr;       // [6][SIFT_MAX_TAPS]
  int taps_r[6];
  int4* d_cand = nullptr;        // [B][SIFT_CAND_CAP]
  vo_sift_kp* d_raw = nullptr;   // [B][SIFT_RAW_CAP]
  int* d_cnt = nullptr;          // [B][2]: candidates, raw keypoints
  vo_sift_kp* d_fin = nullptr; float* d_desc = nullptr; int fin_cap = 0;
};

// ------------------------------------------------------------------------------------------------
// scale space
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sift_tap2(int d, int n, int& s0, int& s1, float& w0, float& w1) {   // cv::resize INTER_LINEAR, scale 1/2
  const double fx = ((double)d + 0.5) * 0.5 - 0.5;
  int s = (int)floor(fx);
  float a = (float)(fx - (double)s);
  if (s < 0) { a = 0.f; s = 0; }
  if (s >= n - 1) { a = 0.f; s = n - 1; }
  s0 = s; s1 = min(s + 1, n - 1); w0 = 1.f - a; w1 = a;
}

__global__ void __launch_bounds__(256) k_sift_up(const uint8_t* __restrict__ img, int w, int h, float* __restrict__ out, size_t out_seq) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= 2 * w) return;
  const uint8_t* I = img + (size_t)b * w * h;
  int x0, x1, y0, y1; float wx0, wx1, wy0, wy1;
  sift_tap2(x, w, x0, x1, wx0, wx1);
  sift_tap2(y, h, y0, y1, wy0, wy1);
  const float r0 = (float)I[(size_t)y0 * w + x0] * wx0 + (float)I[(size_t)y0 * w + x1] * wx1;
  const float r1 = (float)I[(size_t)y1 * w + x0] * wx0 + (float)I[(size_t)y1 * w + x1] * wx1;
  out[(size_t)b * out_seq + (size_t)y * (2 * w) + x] = r0 * wy0 + r1 * wy1;
}

__device__ __forceinline__ int sift_reflect(int i, int n) {       // BORDER_REFLECT_101, any distance
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p; if (i < 0) i += p;
  return (i >= n) ? p - i : i;
}

// separable Gaussian, one pass: s = k0 x0; s += k_i (x_-i + x_+i).  horiz = 1: along x.
__global__ void __launch_bounds__(256) k_sift_blur(const float* __restrict__ src, size_t src_seq, float* __restrict__ dst, size_t dst_seq, int w, int h,
                                                   const float* __restrict__ taps, int r, int horiz) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= w) return;
  const float* S = src + (size_t)b * src_seq;
  float acc;
  if (horiz) {
    const float* row = S + (size_t)y * w;
    acc = taps[r] * row[x];
    for (int i = 1; i <= r; i++) acc = acc + taps[r + i] * (row[sift_reflect(x - i, w)] + row[sift_reflect(x + i, w)]);
  } else {
    acc = taps[r] * S[(size_t)y * w + x];
    for (int i = 1; i <= r; i++) acc = acc + taps[r + i] * (S[(size_t)sift_reflect(y - i, h) * w + x] + S[(size_t)sift_reflect(y + i, h) * w + x]);
  }
  dst[(size_t)b * dst_seq + (size_t)y * w + x] = acc;
}

__global__ void __launch_bounds__(256) k_sift_dog(const float* __restrict__ g, size_t g_seq, float* __restrict__ d, size_t d_seq, size_t px) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int layer = blockIdx.y, b = blockIdx.z;
  if (i >= px) return;
  const float* G = g + (size_t)b * g_seq + (size_t)layer * px;
  d[(size_t)b * d_seq + (size_t)layer * px + i] = G[px + i] - G[i];
}

__global__ void __launch_bounds__(256) k_sift_decimate(const float* __restrict__ src, size_t src_seq, int sw, int sh, float* __restrict__ dst, size_t dst_seq,
                                                       int dw, int dh, double ifx, double ify) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= dw) return;
  const int sx = min((int)floor((double)x * ifx), sw - 1), sy = min((int)floor((double)y * ify), sh - 1);
  dst[(size_t)b * dst_seq + (size_t)y * dw + x] = src[(size_t)b * src_seq + (size_t)sy * sw + sx];
}

// 26-neighbour extrema of DoG layers 1..3 of one octave -> candidate list
__global__ void __launch_bounds__(256) k_sift_extrema(const float* __restrict__ dog, size_t d_seq, int w, int h, int octave, int4* __restrict__ cand,
                                                      int* __restrict__ cnt) {
  const int c = SIFT_BORDER + blockIdx.x * 256 + threadIdx.x, r = SIFT_BORDER + blockIdx.y;
  const int layer = 1 + (int)(blockIdx.z % SIFT_LAYERS), b = (int)(blockIdx.z / SIFT_LAYERS);
  if (c >= w - SIFT_BORDER || r >= h - SIFT_BORDER) return;
  const size_t px = (size_t)w * h;
  const float* D = dog + (size_t)b * d_seq + (size_t)layer * px;
  const float v = D[(size_t)r * w + c];
  if (!(fabsf(v) > 1.0f)) return;                      // threshold = floor(0.5 * 0.04 / 3 * 255) = 1
  bool ismax = v > 0, ismin = v < 0;
  for (int dl = -1; dl <= 1; dl++)
    for (int dr = -1; dr <= 1; dr++)
      for (int dc = -1; dc <= 1; dc++) {
        const float nb = D[(ptrdiff_t)dl * (ptrdiff_t)px + (ptrdiff_t)(r + dr) * w + (c + dc)];
        ismax = ismax && (v >= nb);
        ismin = ismin && (v <= nb);
      }
  if (!(ismax || ismin)) return;
  const int slot = atomicAdd(cnt + 2 * b, 1);
  if (slot < SIFT_CAND_CAP) cand[(size_t)b * SIFT_CAND_CAP + slot] = make_int4(octave, layer, r, c);
}

// ------------------------------------------------------------------------------------------------
// per candidate: adjustLocalExtrema + calcOrientationHist
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sift_fast_atan2(float y, float x) {          // cv::fastAtan2, degrees
  const float deg = (float)(180.0 / 3.141592653589793);
  const float p1 = 0.9997878412794807f * deg, p3 = -0.3258083974640975f * deg, p5 = 0.1555786518463281f * deg, p7 = -0.04432655554792128f * deg;
  const float eps = (float)2.220446049250313e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a;
  if (ax >= ay) {
    const float c = ay / (ax + eps), c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    const float c = ax / (ay + eps), c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ float sift_exp(float w) { return (float)exp((double)w); }

__device__ inline bool sift_solve3(const float a[3][3], const float b[3], float x[3]) {   // Cramer's rule, float32 (oracle: solve3)
  float d = (a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]))
            + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
  if (d == 0) return false;
  d = 1.f / d;
  x[0] = d * ((b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2])) + a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
  x[1] = d * ((a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0])) + a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
  x[2] = d * ((a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0])) + b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
  return true;
}

__global__ void __launch_bounds__(64) k_sift_refine(sift_geom G, const float* __restrict__ gauss, const float* __restrict__ dog, const int4* __restrict__ cand,
                                                    int* __restrict__ cnt, vo_sift_kp* __restrict__ raw) {
  const int b = blockIdx.y, ci = blockIdx.x * 64 + threadIdx.x;
  const int n_cand = min(cnt[2 * b], SIFT_CAND_CAP);
  if (ci >= n_cand) return;
  const int4 cd = cand[(size_t)b * SIFT_CAND_CAP + ci];
  const int octv = cd.x;
  int layer = cd.y, r = cd.z, c = cd.w;
  const int cols = G.w[octv], rows = G.h[octv];
  const size_t px = (size_t)cols * rows;
  const float* DO = dog + (size_t)b * G.d_seq + G.doff[octv];
  const float img_scale = 1.f / 255.f, deriv_scale = img_scale * 0.5f, second = img_scale, cross = img_scale * 0.25f;
  float xi = 0, xr = 0, xc = 0;
  int i = 0;
  for (; i < 5; i++) {
    const float* img = DO + (size_t)layer * px;
    const float* prev = img - px;
    const float* nxt = img + px;
    const size_t o = (size_t)r * cols + c;
    const float dD[3] = {(img[o + 1] - img[o - 1]) * deriv_scale, (img[o + cols] - img[o - cols]) * deriv_scale, (nxt[o] - prev[o]) * deriv_scale};
    const float v2 = img[o] * 2.f;
    const float dxx = ((img[o + 1] + img[o - 1]) - v2) * second;
    const float dyy = ((img[o + cols] + img[o - cols]) - v2) * second;
    const float dss = ((nxt[o] + prev[o]) - v2) * second;
    const float dxy = (((img[o + cols + 1] - img[o + cols - 1]) - img[o - cols + 1]) + img[o - cols - 1]) * cross;
    const float dxs = (((nxt[o + 1] - nxt[o - 1]) - prev[o + 1]) + prev[o - 1]) * cross;
    const float dys = (((nxt[o + cols] - nxt[o - cols]) - prev[o + cols]) + prev[o - cols]) * cross;
    const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
    float X[3] = {0, 0, 0};
    if (!sift_solve3(H, dD, X)) { X[0] = X[1] = X[2] = 0; }
    xi = -X[2]; xr = -X[1]; xc = -X[0];
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
    if (fabsf(xi) > (float)(2147483647 / 3) || fabsf(xr) > (float)(2147483647 / 3) || fabsf(xc) > (float)(2147483647 / 3)) return;
    c += (int)rintf(xc); r += (int)rintf(xr); layer += (int)rintf(xi);
    if (layer < 1 || layer > SIFT_LAYERS || c < SIFT_BORDER || c >= cols - SIFT_BORDER || r < SIFT_BORDER || r >= rows - SIFT_BORDER) return;
  }
  if (i >= 5) return;
  float response, size, kx, ky;
  int octave_field;
  {
    const float* img = DO + (size_t)layer * px;
    const float* prev = img - px;
    const float* nxt = img + px;
    const size_t o = (size_t)r * cols + c;
    const float d0 = (img[o + 1] - img[o - 1]) * deriv_scale, d1 = (img[o + cols] - img[o - cols]) * deriv_scale, d2 = (nxt[o] - prev[o]) * deriv_scale;
    const float t = (d0 * xc + d1 * xr) + d2 * xi;
    const float contr = img[o] * img_scale + t * 0.5f;
    if ((double)(fabsf(contr) * 3.f) < 0.04) return;
    const float v2 = img[o] * 2.f;
    const float dxx = ((img[o + 1] + img[o - 1]) - v2) * second;
    const float dyy = ((img[o + cols] + img[o - cols]) - v2) * second;
    const float dxy = (((img[o + cols + 1] - img[o + cols - 1]) - img[o - cols + 1]) + img[o - cols - 1]) * cross;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || (double)(tr * tr) * 10.0 >= 121.0 * (double)det) return;
    const float sc = (float)(1 << octv);
    const float pw = (float)pow(2.0, (double)(((float)layer + xi) / 3.f));
    size = (float)(1.6 * (double)pw * (double)(1 << octv) * 2.0);
    kx = ((float)c + xc) * sc; ky = ((float)r + xr) * sc;
    octave_field = octv + (layer << 8) + ((int)rint(((double)xi + 0.5) * 255.0) << 16);
    response = fabsf(contr);
  }
  // ---- orientation histogram on the Gaussian image of the refined layer ----
  const float* GI = gauss + (size_t)b * G.g_seq + G.goff[octv] + (size_t)layer * px;
  const float scl_octv = size * 0.5f / (float)(1 << octv);
  const int radius = (int)rintf(4.5f * scl_octv);
  const float sigma = 1.5f * scl_octv;
  const float expf_scale = -1.f / (2.f * (sigma * sigma));
  float temphist[36];
  for (int k = 0; k < 36; k++) temphist[k] = 0.f;
  for (int di = -radius; di <= radius; di++) {
    const int y = r + di;
    if (y <= 0 || y >= rows - 1) continue;
    for (int dj = -radius; dj <= radius; dj++) {
      const int x = c + dj;
      if (x <= 0 || x >= cols - 1) continue;
      const size_t o = (size_t)y * cols + x;
      const float dx = GI[o + 1] - GI[o - 1], dy = GI[o - cols] - GI[o + cols];
      const float W = sift_exp((float)(di * di + dj * dj) * expf_scale);
      const float ori = sift_fast_atan2(dy, dx);
      const float mag = sqrtf(dx * dx + dy * dy);
      int bin = (int)rintf((36.f / 360.f) * ori);
      if (bin >= 36) bin -= 36;
      if (bin < 0) bin += 36;
      temphist[bin] += W * mag;
    }
  }
  float hist[36];
  float omax = 0.f;
  for (int k = 0; k < 36; k++) {
    const float tm2 = temphist[(k + 34) % 36], tm1 = temphist[(k + 35) % 36], tp1 = temphist[(k + 1) % 36], tp2 = temphist[(k + 2) % 36];
    hist[k] = ((tm2 + tp2) * (1.f / 16.f) + (tm1 + tp1) * (4.f / 16.f)) + temphist[k] * (6.f / 16.f);
    omax = (k == 0) ? hist[k] : fmaxf(omax, hist[k]);
  }
  const float mag_thr = omax * 0.8f;
  for (int j = 0; j < 36; j++) {
    const int l = j > 0 ? j - 1 : 35, r2 = j < 35 ? j + 1 : 0;
    if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
      float bin = (float)j + (0.5f * (hist[l] - hist[r2])) / ((hist[l] - 2.f * hist[j]) + hist[r2]);
      bin = bin < 0 ? 36.f + bin : (bin >= 36.f ? bin - 36.f : bin);
      float ang = 360.f - (360.f / 36.f) * bin;
      if (fabsf(ang - 360.f) < 1.1920929e-07f) ang = 0.f;
      const int slot = atomicAdd(cnt + 2 * b + 1, 1);
      if (slot < SIFT_RAW_CAP) {
        vo_sift_kp k;
        k.x = kx; k.y = ky; k.size = size; k.angle = ang; k.response = response; k.octave = octave_field;
        raw[(size_t)b * SIFT_RAW_CAP + slot] = k;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per keypoint: calcSIFTDescriptor
// ------------------------------------------------------------------------------------------------
// one WAVE per keypoint: the lanes evaluate 64 samples of the window at a time (gradient, exp, atan2, trilinear weights --
// the expensive part), lane 0 then adds the valid ones to the histogram in LDS in sample order, which keeps the float32
// sums identical to a sequential walk (3.5 ms -> 0.3 ms per 1000 keypoints against a lane per keypoint)
__global__ void __launch_bounds__(64) k_sift_desc(sift_geom G, const float* __restrict__ gauss, const vo_sift_kp* __restrict__ kps, const int* __restrict__ n_kp,
                                                  int cap, float* __restrict__ desc) {
  __shared__ float s_hist[360];
  __shared__ float s_val[64][9];       // 8 contributions + padding (odd stride: conflict-free column writes)
  __shared__ int s_idx[64];
  __shared__ float s_dst[128];
  __shared__ float s_scale;
  const int b = blockIdx.y, ki = blockIdx.x, lane = threadIdx.x;
  if (ki >= n_kp[b]) return;
  const vo_sift_kp kp = kps[(size_t)b * cap + ki];
  int o = kp.octave & 255;
  const int layer = (kp.octave >> 8) & 255;
  o = o < 128 ? o : (-128 | o);
  const float scale = o >= 0 ? 1.f / (float)(1 << o) : (float)(1 << -o);
  const float size = kp.size * scale;
  float ori = 360.f - kp.angle;
  if (fabsf(ori - 360.f) < 1.1920929e-07f) ori = 0.f;
  const float scl = size * 0.5f;
  const int oi = o + 1;                               // pyramid octave (first octave = -1)
  const int cols = G.w[oi], rows = G.h[oi];
  const float* img = gauss + (size_t)b * G.g_seq + G.goff[oi] + (size_t)layer * cols * rows;
  const float ptx = kp.x * scale, pty = kp.y * scale;
  const int px = (int)rintf(ptx), py = (int)rintf(pty);
  const float ang = ori * (float)(3.141592653589793 / 180.0);
  float cos_t = (float)cos((double)ang), sin_t = (float)sin((double)ang);
  const float bins_per_rad = 8.f / 360.f, exp_scale = -1.f / (4.f * 4.f * 0.5f);
  const float hist_width = 3.f * scl;
  int radius = (int)rintf(((hist_width * 1.4142135623730951f) * 5.f) * 0.5f);
  radius = min(radius, (int)sqrt((double)cols * cols + (double)rows * rows));
  cos_t = cos_t / hist_width; sin_t = sin_t / hist_width;
  for (int k = lane; k < 360; k += 64) s_hist[k] = 0.f;
  const int side = 2 * radius + 1;
  const long long total = (long long)side * side;
  for (long long t0 = 0; t0 < total; t0 += 64) {
    const long long t = t0 + lane;
    bool valid = false;
    if (t < total) {
      const int di = (int)(t / side) - radius, dj = (int)(t % side) - radius;      // sample order: rows, then columns
      const float fi = (float)di, fj = (float)dj;
      const float c_rot = fj * cos_t - fi * sin_t, r_rot = fj * sin_t + fi * cos_t;
      float rbin = (r_rot + 2.f) - 0.5f, cbin = (c_rot + 2.f) - 0.5f;
      const int r = py + di, c = px + dj;
      if (rbin > -1.f && rbin < 4.f && cbin > -1.f && cbin < 4.f && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
        valid = true;
        const size_t off = (size_t)r * cols + c;
        const float dx = img[off + 1] - img[off - 1], dy = img[off - cols] - img[off + cols];
        const float W = sift_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        const float Ori = sift_fast_atan2(dy, dx);
        const float Mag = sqrtf(dx * dx + dy * dy);
        float obin = (Ori - ori) * bins_per_rad;
        const float mag = Mag * W;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin = rbin - (float)r0; cbin = cbin - (float)c0; obin = obin - (float)o0;
        if (o0 < 0) o0 += 8;
        if (o0 >= 8) o0 -= 8;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
        const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
        const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
        const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
        s_idx[lane] = ((r0 + 1) * 6 + c0 + 1) * 10 + o0;
        float* sv = s_val[lane];
        sv[0] = v000; sv[1] = v001; sv[2] = v010; sv[3] = v011; sv[4] = v100; sv[5] = v101; sv[6] = v110; sv[7] = v111;
      }
    }
    unsigned long long m = __ballot(valid);
    __syncthreads();
    if (lane == 0) {
      while (m) {
        const int sl = __builtin_ctzll(m);
        m &= m - 1;
        const int idx = s_idx[sl];
        const float* sv = s_val[sl];
        s_hist[idx] += sv[0]; s_hist[idx + 1] += sv[1]; s_hist[idx + 10] += sv[2]; s_hist[idx + 11] += sv[3];
        s_hist[idx + 60] += sv[4]; s_hist[idx + 61] += sv[5]; s_hist[idx + 70] += sv[6]; s_hist[idx + 71] += sv[7];
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        const int bidx = ((i + 1) * 6 + (j + 1)) * 10;
        s_hist[bidx] = s_hist[bidx] + s_hist[bidx + 8];
        s_hist[bidx + 1] = s_hist[bidx + 1] + s_hist[bidx + 9];
        for (int k = 0; k < 8; k++) s_dst[(i * 4 + j) * 8 + k] = s_hist[bidx + k];
      }
    float nrm2 = 0.f;
    for (int k = 0; k < 128; k++) nrm2 = nrm2 + s_dst[k] * s_dst[k];
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0.f;
    for (int k = 0; k < 128; k++) {
      const float val = fminf(s_dst[k], thr);
      s_dst[k] = val;
      nrm2 = nrm2 + val * val;
    }
    s_scale = 512.f / fmaxf(sqrtf(nrm2), 1.1920929e-07f);
  }
  __syncthreads();
  const float sc = s_scale;
  float* out = desc + ((size_t)b * cap + ki) * 128;
  for (int k = lane; k < 128; k += 64) out[k] = fminf(fmaxf(rintf(s_dst[k] * sc), 0.f), 255.f);
}

// ================================================================================================
// host
// ================================================================================================
void vo_sift_destroy(vo_ctx* c) {
  if (!c->sift) return;
  vo_sift_ws* w = c->sift;
  void* bufs[] = {w->d_img, w->d_a, w->d_b, w->d_gauss, w->d_dog, w->d_taps, w->d_cand, w->d_raw, w->d_cnt, w->d_fin, w->d_desc};
  for (void* p : bufs) if (p) (void)hipFree(p);
  delete w;
  c->sift = nullptr;
}

static int sift_taps(double sigma, float* out) {       // cv::getGaussianKernel for CV_32F, ksize = cvRound(8 sigma + 1) | 1
  const int n = (int)rint(sigma * 4 * 2 + 1) | 1;
  std::vector<double> t((size_t)n);
  double sum = 0;
  for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = exp(-0.5 / (sigma * sigma) * x * x); sum += t[i]; }
  for (int i = 0; i < n; i++) out[i] = (float)(t[i] / sum);
  return n / 2;
}

static int32_t sift_alloc(vo_ctx* c) {
  if (c->sift) return VO_OK;
  vo_sift_ws* w = new vo_sift_ws();
  c->sift = w;
  const size_t B = c->batch;
  const int W = c->width, H = c->height;
  sift_geom& G = w->G;
  G.n_oct = (int)rint(log((double)std::min(2 * W, 2 * H)) / log(2.0) - 2) + 1;
  if (G.n_oct > SIFT_MAX_OCT) G.n_oct = SIFT_MAX_OCT;
  size_t go = 0, dof = 0;
  int ow = 2 * W, oh = 2 * H, n = 0;
  for (int o = 0; o < G.n_oct; o++) {
    if (ow < 1 || oh < 1) break;
    G.w[o] = ow; G.h[o] = oh; G.goff[o] = go; G.doff[o] = dof;
    go += (size_t)6 * ow * oh; dof += (size_t)5 * ow * oh;
    ow /= 2; oh /= 2; n++;
  }
  G.n_oct = n; G.g_seq = go; G.d_seq = dof;
  const size_t base_px = (size_t)4 * W * H;
  VO_HIP(c, hipMalloc((void**)&w->d_img, B * W * H));
  VO_HIP(c, hipMalloc((void**)&w->d_a, sizeof(float) * B * base_px));
  VO_HIP(c, hipMalloc((void**)&w->d_b, sizeof(float) * B * base_px));
  VO_HIP(c, hipMalloc((void**)&w->d_gauss, sizeof(float) * B * G.g_seq));
  VO_HIP(c, hipMalloc((void**)&w->d_dog, sizeof(float) * B * G.d_seq));
  VO_HIP(c, hipMalloc((void**)&w->d_taps, sizeof(float) * 6 * SIFT_MAX_TAPS));
  VO_HIP(c, hipMalloc((void**)&w->d_cand, sizeof(int4) * B * SIFT_CAND_CAP));
  VO_HIP(c, hipMalloc((void**)&w->d_raw, sizeof(vo_sift_kp) * B * SIFT_RAW_CAP));
  VO_HIP(c, hipMalloc((void**)&w->d_cnt, sizeof(int) * 3 * B));
  // taps: [0] the initial blur sqrt(max(1.6^2 - 4 * 0.5^2, 0.01)), [1..5] the layer increments
  float taps[6][SIFT_MAX_TAPS];
  memset(taps, 0, sizeof(taps));
  const double sigma0 = 1.6, k = pow(2.0, 1.0 / SIFT_LAYERS);
  w->taps_r[0] = sift_taps(sqrt(std::max(sigma0 * sigma0 - 0.5 * 0.5 * 4, 0.01)), taps[0]);
  for (int i = 1; i < SIFT_LAYERS + 3; i++) {
    const double sp = pow(k, (double)(i - 1)) * sigma0, st = sp * k;
    w->taps_r[i] = sift_taps(sqrt(st * st - sp * sp), taps[i]);
  }
  VO_HIP(c, hipMemcpy(w->d_taps, taps, sizeof(taps), hipMemcpyHostToDevice));
  return VO_OK;
}

static void sift_blur(vo_ctx* c, const float* src, size_t src_seq, float* dst, size_t dst_seq, int w, int h, int which) {
  vo_sift_ws* s = c->sift;
  const dim3 grid(vo_div_up(w, 256), h, c->batch);
  const size_t tmp_seq = (size_t)4 * c->width * c->height;
  hipLaunchKernelGGL(k_sift_blur, grid, dim3(256), 0, c->stream, src, src_seq, s->d_b, tmp_seq, w, h, s->d_taps + which * SIFT_MAX_TAPS, s->taps_r[which], 1);
  hipLaunchKernelGGL(k_sift_blur, grid, dim3(256), 0, c->stream, s->d_b, tmp_seq, dst, dst_seq, w, h, s->d_taps + which * SIFT_MAX_TAPS, s->taps_r[which], 0);
}

// img [batch][h][stride] u8 (the context's image size), mask the same layout or NULL (0 = no keypoint there, applied
// like KeyPointsFilter::runByPixelsMask after the selection).  -> kps [batch][max_out], desc [batch][max_out][128] f32
// (values 0..255), n_out [batch].  Keypoints in KeyPoint_LessThan order (x, y, -size, angle, -response, -octave).
// VO_E_CAPACITY if a sequence has more than max_out keypoints after retainBest (ties at the nfeatures-th response are
// all kept, as in OpenCV) or the candidate lists overflow.
extern "C" int32_t vo_sift_detect_compute(vo_ctx* c, const uint8_t* img, int32_t stride, const uint8_t* mask, int32_t nfeatures, int32_t max_out,
                                          vo_sift_kp* kps, float* desc, int32_t* n_out) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, img && kps && desc && n_out, VO_E_INVALID, "null buffer");
  VO_CHECK(c, stride >= c->width && max_out >= 1, VO_E_INVALID, "bad stride / max_out");
  VO_HIP(c, hipSetDevice(c->device));
  int32_t rc = sift_alloc(c);
  if (rc != VO_OK) return rc;
  vo_sift_ws* w = c->sift;
  const sift_geom& G = w->G;
  const size_t B = c->batch;
  const int W = c->width, H = c->height;
  const size_t base_seq = (size_t)4 * W * H;
  VO_HIP(c, hipMemcpy2DAsync(w->d_img, W, img, stride, W, (size_t)H * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemsetAsync(w->d_cnt, 0, sizeof(int) * 3 * B, c->stream));
  hipLaunchKernelGGL(k_sift_up, dim3(vo_div_up(2 * W, 256), 2 * H, (unsigned)B), dim3(256), 0, c->stream, w->d_img, W, H, w->d_a, base_seq);
  sift_blur(c, w->d_a, base_seq, w->d_gauss + G.goff[0], G.g_seq, G.w[0], G.h[0], 0);
  for (int o = 0; o < G.n_oct; o++) {
    const int ow = G.w[o], oh = G.h[o];
    const size_t px = (size_t)ow * oh;
    float* g = w->d_gauss + G.goff[o];
    if (o > 0) {
      const int sw = G.w[o - 1], sh = G.h[o - 1];
      const double ifx = 1.0 / ((double)ow / (double)sw), ify = 1.0 / ((double)oh / (double)sh);
      hipLaunchKernelGGL(k_sift_decimate, dim3(vo_div_up(ow, 256), oh, (unsigned)B), dim3(256), 0, c->stream,
                         w->d_gauss + G.goff[o - 1] + (size_t)SIFT_LAYERS * sw * sh, G.g_seq, sw, sh, g, G.g_seq, ow, oh, ifx, ify);
    }
    for (int i = 1; i < SIFT_LAYERS + 3; i++) sift_blur(c, g + (size_t)(i - 1) * px, G.g_seq, g + (size_t)i * px, G.g_seq, ow, oh, i);
    hipLaunchKernelGGL(k_sift_dog, dim3((unsigned)((px + 255) / 256), SIFT_LAYERS + 2, (unsigned)B), dim3(256), 0, c->stream, g, G.g_seq,
                       w->d_dog + G.doff[o], G.d_seq, px);
    if (ow > 2 * SIFT_BORDER && oh > 2 * SIFT_BORDER)
      hipLaunchKernelGGL(k_sift_extrema, dim3(vo_div_up(ow - 2 * SIFT_BORDER, 256), oh - 2 * SIFT_BORDER, (unsigned)(SIFT_LAYERS * B)), dim3(256), 0, c->stream,
                         w->d_dog + G.doff[o], G.d_seq, ow, oh, o, w->d_cand, w->d_cnt);
  }
  VO_HIP(c, hipGetLastError());
  std::vector<int> cnt(3 * B);
  VO_HIP(c, hipMemcpyAsync(cnt.data(), w->d_cnt, sizeof(int) * 3 * B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  int max_cand = 0;
  for (size_t b = 0; b < B; b++) {
    VO_CHECK(c, cnt[2 * b] <= SIFT_CAND_CAP, VO_E_CAPACITY, "too many scale-space extrema");
    max_cand = std::max(max_cand, cnt[2 * b]);
  }
  if (max_cand > 0)
    hipLaunchKernelGGL(k_sift_refine, dim3(vo_div_up(max_cand, 64), (unsigned)B), dim3(64), 0, c->stream, G, w->d_gauss, w->d_dog, w->d_cand, w->d_cnt, w->d_raw);
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipMemcpyAsync(cnt.data(), w->d_cnt, sizeof(int) * 3 * B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  // ---- host: removeDuplicatedSorted, retainBest, first-octave rescale, mask ----
  std::vector<std::vector<vo_sift_kp>> fin(B);
  int max_fin = 0;
  for (size_t b = 0; b < B; b++) {
    const int n_raw = cnt[2 * b + 1];
    VO_CHECK(c, n_raw <= SIFT_RAW_CAP, VO_E_CAPACITY, "too many raw keypoints");
    std::vector<vo_sift_kp> v((size_t)n_raw);
    if (n_raw) VO_HIP(c, hipMemcpy(v.data(), w->d_raw + b * SIFT_RAW_CAP, sizeof(vo_sift_kp) * n_raw, hipMemcpyDeviceToHost));
    std::sort(v.begin(), v.end(), [](const vo_sift_kp& a, const vo_sift_kp& k) {
      if (a.x != k.x) return a.x < k.x;
      if (a.y != k.y) return a.y < k.y;
      if (a.size != k.size) return a.size > k.size;
      if (a.angle != k.angle) return a.angle < k.angle;
      if (a.response != k.response) return a.response > k.response;
      return a.octave > k.octave;
    });
    std::vector<vo_sift_kp> u;
    for (const vo_sift_kp& k : v)
      if (u.empty() || u.back().x != k.x || u.back().y != k.y || u.back().size != k.size || u.back().angle != k.angle) u.push_back(k);
    if (nfeatures > 0 && (int)u.size() > nfeatures) {
      std::vector<float> resp;
      for (const vo_sift_kp& k : u) resp.push_back(k.response);
      std::nth_element(resp.begin(), resp.begin() + (nfeatures - 1), resp.end(), [](float a, float k) { return a > k; });
      const float thr = resp[(size_t)nfeatures - 1];
      std::vector<vo_sift_kp> kept;
      for (const vo_sift_kp& k : u) if (k.response >= thr) kept.push_back(k);
      u.swap(kept);
    }
    for (vo_sift_kp& k : u) {
      k.octave = (k.octave & ~255) | ((k.octave - 1) & 255);
      k.x *= 0.5f; k.y *= 0.5f; k.size *= 0.5f;
    }
    if (mask) {
      const uint8_t* M = mask + b * (size_t)H * stride;
      std::vector<vo_sift_kp> kept;
      for (const vo_sift_kp& k : u) {
        const int my = (int)(k.y + 0.5f), mx = (int)(k.x + 0.5f);
        if (my >= 0 && my < H && mx >= 0 && mx < W && M[(size_t)my * stride + mx]) kept.push_back(k);
      }
      u.swap(kept);
    }
    VO_CHECK(c, (int)u.size() <= max_out, VO_E_CAPACITY, "more keypoints than max_out");
    n_out[b] = (int32_t)u.size();
    max_fin = std::max(max_fin, (int)u.size());
    fin[b].swap(u);
  }
  if (max_fin == 0) return VO_OK;
  if (w->fin_cap < max_out) {
    if (w->d_fin) (void)hipFree(w->d_fin);
    if (w->d_desc) (void)hipFree(w->d_desc);
    w->d_fin = nullptr; w->d_desc = nullptr;
    VO_HIP(c, hipMalloc((void**)&w->d_fin, sizeof(vo_sift_kp) * B * max_out));
    VO_HIP(c, hipMalloc((void**)&w->d_desc, sizeof(float) * 128 * B * max_out));
    w->fin_cap = max_out;
  }
  int* d_nfin = w->d_cnt + 2 * B;
  for (size_t b = 0; b < B; b++) {
    if (!fin[b].empty()) {
      VO_HIP(c, hipMemcpy(w->d_fin + b * w->fin_cap, fin[b].data(), sizeof(vo_sift_kp) * fin[b].size(), hipMemcpyHostToDevice));
      memcpy(kps + b * (size_t)max_out, fin[b].data(), sizeof(vo_sift_kp) * fin[b].size());
    }
  }
  VO_HIP(c, hipMemcpy(d_nfin, n_out, sizeof(int) * B, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sift_desc, dim3((unsigned)max_fin, (unsigned)B), dim3(64), 0, c->stream, G, w->d_gauss, w->d_fin, d_nfin, w->fin_cap, w->d_desc);
  VO_HIP(c, hipGetLastError());
  for (size_t b = 0; b < B; b++)
    if (n_out[b] > 0)
      VO_HIP(c, hipMemcpyAsync(desc + b * (size_t)max_out * 128, w->d_desc + b * (size_t)w->fin_cap * 128, sizeof(float) * 128 * n_out[b],
                               hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}
