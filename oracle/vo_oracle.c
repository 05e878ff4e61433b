/*
 * vo_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the arithmetic the reference's hot path delegates to
 * OpenCV 4.4.0 (pinned by /root/reference/setup/conda_env.yml:57,78,87; the
 * library is NOT vendored under /root/reference and is not installed in this
 * image).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libvo_mi355x.so) never links or calls it.
 *
 * PARITY STATUS AT THE OPENCV BOUNDARY: *** parity unpinned ***
 *   The reference holds no tests, golden vectors or fixtures for these calls and
 *   cv2 cannot be imported here, so this restatement follows the published
 *   OpenCV-4.4 algorithms and is validated against analytic
 *   ground truth (known warps / known 3-D points), not against OpenCV outputs.
 *
 * Reference call sites restated here:
 *   cv2.calcOpticalFlowPyrLK   /root/reference/src/extractor/extractor.py:44-45,65-66
 *        (winSize 31x31, maxLevel 3, criteria (EPS|COUNT, 30, 0.03): extractor.py:16-19)
 *   cv2.goodFeaturesToTrack    /root/reference/src/extractor/extractor.py:111
 *        (maxCorners 1000, quality 0.03, minDistance, blockSize 31: extractor.py:21-24)
 *   cv2.circle(mask, ...)      /root/reference/src/extractor/extractor.py:104-107
 *   cv2.triangulatePoints      /root/reference/src/extractor/extractor.py:270
 *   cv2.bilateralFilter        /root/reference/src/loader/loader.py:16-20,86   (d 5, sigmaColor 1.5, sigmaSpace 1.5;
 *        SURVEY.md 8f "next" row 2: the loader's pre-filter, immediately before the frame enters the path)
 *
 * Stated deviations from OpenCV (both are exact-arithmetic refinements, made so
 * that the GPU path can be compared BIT-EXACTLY with this oracle):
 *   (1) KLT: the 2x2 normal matrix and mismatch vector are accumulated exactly in
 *       64-bit integers (acc_mode=1) instead of float accumulators.  OpenCV's own
 *       scalar and SIMD paths already differ from each other in float summation
 *       order; acc_mode=0 reproduces the scalar float order for comparison.
 *   (2) Shi-Tomasi: the 31x31 structure-tensor sums are accumulated exactly in
 *       int32 (Sobel outputs are integers before the scale) and rounded to float
 *       once (exact_int=1); exact_int=0 follows OpenCV's float running sums.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VO_MAX_LEVELS 8

static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

static inline int pix101(const uint8_t* img, int w, int h, int x, int y) {
  return img[(size_t)reflect101(y, h) * w + reflect101(x, w)];
}

/* ------------------------------------------------------------------------- */
/* pyrDown: 5x5 separable [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101   */
/* (imgproc/pyramids.cpp, uchar path)               */
/* ------------------------------------------------------------------------- */
void vo_oracle_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) {
  static const int k5[5] = {1, 4, 6, 4, 1};
  int dw = (w + 1) / 2, dh = (h + 1) / 2;
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++) {
      int sum = 0;
      for (int j = -2; j <= 2; j++) {
        int row = 0;
        for (int i = -2; i <= 2; i++) row += k5[i + 2] * pix101(src, w, h, 2 * x + i, 2 * y + j);
        sum += k5[j + 2] * row;
      }
      dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
    }
}

/* ------------------------------------------------------------------------- */
/* Scharr derivative, int16 interleaved (Ix, Iy), un-normalised (x32),         */
/* reflect-101 at the image border (video/lkpyramid.cpp calcSharrDeriv)        */
/* ------------------------------------------------------------------------- */
void vo_oracle_scharr(const uint8_t* src, int w, int h, int16_t* dst) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int t0m, t0p, t1m, t1c, t1p;
      /* vertical smooth  t0 = 3*(up+down) + 10*mid ; vertical diff t1 = down - up */
#define COL_T0(xx) ((pix101(src, w, h, (xx), y - 1) + pix101(src, w, h, (xx), y + 1)) * 3 + pix101(src, w, h, (xx), y) * 10)
#define COL_T1(xx) (pix101(src, w, h, (xx), y + 1) - pix101(src, w, h, (xx), y - 1))
      t0m = COL_T0(x - 1); t0p = COL_T0(x + 1);
      t1m = COL_T1(x - 1); t1c = COL_T1(x); t1p = COL_T1(x + 1);
#undef COL_T0
#undef COL_T1
      dst[((size_t)y * w + x) * 2 + 0] = (int16_t)(t0p - t0m);
      dst[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
    }
}

/* number of pyramid levels actually used (buildOpticalFlowPyramid truncation) */
int vo_oracle_pyr_levels(int w, int h, int win, int max_level) {
  int lv = 0;
  while (lv < max_level) {
    int nw = (w + 1) / 2, nh = (h + 1) / 2;
    if (nw <= win || nh <= win) break;
    w = nw; h = nh; lv++;
  }
  return lv; /* highest level index */
}

/* ------------------------------------------------------------------------- */
/* calcOpticalFlowPyrLK (video/lkpyramid.cpp, LKTrackerInvoker), flags = 0     */
/* ------------------------------------------------------------------------- */
typedef struct {
  int w, h;
  uint8_t* img;
  int16_t* der; /* only for prev */
} level_t;

static inline int deriv0(const int16_t* d, int w, int h, int x, int y, int c) {
  if (x < 0 || x >= w || y < 0 || y >= h) return 0; /* BORDER_CONSTANT padding */
  return d[((size_t)y * w + x) * 2 + c];
}

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline void lk_weights(float a, float b, int* iw00, int* iw01, int* iw10, int* iw11) {
  *iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
  *iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
  *iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
  *iw11 = (1 << W_BITS) - *iw00 - *iw01 - *iw10;
}

/*
 * im0, im1 : uint8 H x W (C-contiguous);  p0 : n x 2 float32
 * out: p1 (n x 2), status (n), err (n), iters (n x (max_level+1), may be NULL:
 *      iterations executed per level, -1 if the level was skipped for the point)
 * acc_mode: 1 exact int64 accumulation, 0 float accumulators in raster order
 * returns the highest pyramid level used.
 */
int vo_oracle_klt(const uint8_t* im0, const uint8_t* im1, int w, int h, const float* p0, int n,
                  int win, int max_level, int max_count, double eps, float min_eig_thr,
                  int acc_mode, float* p1, uint8_t* status, float* err, int32_t* iters) {
  level_t L0[VO_MAX_LEVELS], L1[VO_MAX_LEVELS];
  if (max_level >= VO_MAX_LEVELS) max_level = VO_MAX_LEVELS - 1;
  if (max_count < 0) max_count = 0;
  if (max_count > 100) max_count = 100;
  if (eps < 0) eps = 0;
  if (eps > 10) eps = 10;
  eps *= eps;
  int top = vo_oracle_pyr_levels(w, h, win, max_level);
  for (int l = 0; l <= top; l++) {
    int lw = l ? (L0[l - 1].w + 1) / 2 : w, lh = l ? (L0[l - 1].h + 1) / 2 : h;
    L0[l].w = L1[l].w = lw; L0[l].h = L1[l].h = lh;
    L0[l].img = (uint8_t*)malloc((size_t)lw * lh);
    L1[l].img = (uint8_t*)malloc((size_t)lw * lh);
    if (l == 0) { memcpy(L0[0].img, im0, (size_t)w * h); memcpy(L1[0].img, im1, (size_t)w * h); }
    else {
      vo_oracle_pyr_down(L0[l - 1].img, L0[l - 1].w, L0[l - 1].h, L0[l].img);
      vo_oracle_pyr_down(L1[l - 1].img, L1[l - 1].w, L1[l - 1].h, L1[l].img);
    }
    L0[l].der = (int16_t*)malloc((size_t)lw * lh * 2 * sizeof(int16_t));
    L1[l].der = NULL;
    vo_oracle_scharr(L0[l].img, lw, lh, L0[l].der);
  }
  for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0.f; }
  if (iters) for (int i = 0; i < n * (max_level + 1); i++) iters[i] = -1;

  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (win - 1) * 0.5f;
  int16_t* Iw = (int16_t*)malloc((size_t)win * win * sizeof(int16_t));
  int16_t* dIw = (int16_t*)malloc((size_t)win * win * 2 * sizeof(int16_t));

  for (int level = top; level >= 0; level--) {
    const level_t* I = &L0[level];
    const level_t* J = &L1[level];
    const int cols = I->w, rows = I->h;
    for (int pt = 0; pt < n; pt++) {
      float prevx = p0[2 * pt] * (float)(1. / (1 << level));
      float prevy = p0[2 * pt + 1] * (float)(1. / (1 << level));
      float nextx, nexty;
      if (level == top) { nextx = prevx; nexty = prevy; }
      else { nextx = p1[2 * pt] * 2.f; nexty = p1[2 * pt + 1] * 2.f; }
      p1[2 * pt] = nextx; p1[2 * pt + 1] = nexty;

      prevx -= half; prevy -= half;
      int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
      if (ipx < -win || ipx >= cols || ipy < -win || ipy >= rows) {
        if (level == 0) { status[pt] = 0; err[pt] = 0.f; }
        continue;
      }
      float a = prevx - ipx, b = prevy - ipy;
      int iw00, iw01, iw10, iw11;
      lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);

      float fA11 = 0, fA12 = 0, fA22 = 0;
      int64_t iA11 = 0, iA12 = 0, iA22 = 0;
      for (int y = 0; y < win; y++)
        for (int x = 0; x < win; x++) {
          int sx = ipx + x, sy = ipy + y;
          int ival = DESCALE(pix101(I->img, cols, rows, sx, sy) * iw00 + pix101(I->img, cols, rows, sx + 1, sy) * iw01 +
                             pix101(I->img, cols, rows, sx, sy + 1) * iw10 + pix101(I->img, cols, rows, sx + 1, sy + 1) * iw11,
                             W_BITS - 5);
          int ixval = DESCALE(deriv0(I->der, cols, rows, sx, sy, 0) * iw00 + deriv0(I->der, cols, rows, sx + 1, sy, 0) * iw01 +
                              deriv0(I->der, cols, rows, sx, sy + 1, 0) * iw10 + deriv0(I->der, cols, rows, sx + 1, sy + 1, 0) * iw11,
                              W_BITS);
          int iyval = DESCALE(deriv0(I->der, cols, rows, sx, sy, 1) * iw00 + deriv0(I->der, cols, rows, sx + 1, sy, 1) * iw01 +
                              deriv0(I->der, cols, rows, sx, sy + 1, 1) * iw10 + deriv0(I->der, cols, rows, sx + 1, sy + 1, 1) * iw11,
                              W_BITS);
          Iw[y * win + x] = (int16_t)ival;
          dIw[(y * win + x) * 2] = (int16_t)ixval;
          dIw[(y * win + x) * 2 + 1] = (int16_t)iyval;
          fA11 += (float)(ixval * ixval); fA12 += (float)(ixval * iyval); fA22 += (float)(iyval * iyval);
          iA11 += (int64_t)ixval * ixval; iA12 += (int64_t)ixval * iyval; iA22 += (int64_t)iyval * iyval;
        }
      float A11, A12, A22;
      if (acc_mode) { A11 = (float)iA11 * FLT_SCALE; A12 = (float)iA12 * FLT_SCALE; A22 = (float)iA22 * FLT_SCALE; }
      else { A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE; }

      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
      if (minEig < min_eig_thr || D < 1.1920929e-07f /* FLT_EPSILON */) {
        if (level == 0) status[pt] = 0;
        continue;
      }
      D = 1.f / D;

      nextx -= half; nexty -= half;
      float pdx = 0, pdy = 0;
      int j;
      for (j = 0; j < max_count; j++) {
        int inx = (int)floorf(nextx), iny = (int)floorf(nexty);
        if (inx < -win || inx >= cols || iny < -win || iny >= rows) {
          if (level == 0) status[pt] = 0;
          break;
        }
        a = nextx - inx; b = nexty - iny;
        lk_weights(a, b, &iw00, &iw01, &iw10, &iw11);
        float fb1 = 0, fb2 = 0;
        int64_t ib1 = 0, ib2 = 0;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            int sx = inx + x, sy = iny + y;
            int diff = DESCALE(pix101(J->img, cols, rows, sx, sy) * iw00 + pix101(J->img, cols, rows, sx + 1, sy) * iw01 +
                               pix101(J->img, cols, rows, sx, sy + 1) * iw10 + pix101(J->img, cols, rows, sx + 1, sy + 1) * iw11,
                               W_BITS - 5) - Iw[y * win + x];
            int gx = dIw[(y * win + x) * 2], gy = dIw[(y * win + x) * 2 + 1];
            fb1 += (float)(diff * gx); fb2 += (float)(diff * gy);
            ib1 += (int64_t)diff * gx; ib2 += (int64_t)diff * gy;
          }
        float b1, b2;
        if (acc_mode) { b1 = (float)ib1 * FLT_SCALE; b2 = (float)ib2 * FLT_SCALE; }
        else { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
        float dx = (A12 * b2 - A22 * b1) * D;
        float dy = (A12 * b1 - A11 * b2) * D;
        nextx += dx; nexty += dy;
        p1[2 * pt] = nextx + half; p1[2 * pt + 1] = nexty + half;
        if ((double)dx * dx + (double)dy * dy <= eps) { j++; break; }
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
          p1[2 * pt] -= dx * 0.5f; p1[2 * pt + 1] -= dy * 0.5f;
          j++;
          break;
        }
        pdx = dx; pdy = dy;
      }
      if (iters) iters[pt * (max_level + 1) + level] = j;

      if (status[pt] && level == 0) {
        float nx = p1[2 * pt] - half, ny = p1[2 * pt + 1] - half;
        int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (inx < -win || inx >= cols || iny < -win || iny >= rows) { status[pt] = 0; continue; }
        float aa = nx - inx, bb = ny - iny;
        lk_weights(aa, bb, &iw00, &iw01, &iw10, &iw11);
        float ferr = 0.f;
        int64_t ierr = 0;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            int sx = inx + x, sy = iny + y;
            int diff = DESCALE(pix101(J->img, cols, rows, sx, sy) * iw00 + pix101(J->img, cols, rows, sx + 1, sy) * iw01 +
                               pix101(J->img, cols, rows, sx, sy + 1) * iw10 + pix101(J->img, cols, rows, sx + 1, sy + 1) * iw11,
                               W_BITS - 5) - Iw[y * win + x];
            ferr += fabsf((float)diff);
            ierr += diff < 0 ? -diff : diff;
          }
        err[pt] = (acc_mode ? (float)ierr : ferr) * 1.f / (float)(32 * win * win);
      }
    }
  }
  free(Iw); free(dIw);
  for (int l = 0; l <= top; l++) { free(L0[l].img); free(L1[l].img); free(L0[l].der); }
  return top;
}

/* ------------------------------------------------------------------------- */
/* cv2.circle(mask, c, r, color, -1)  (imgproc/drawing.cpp Circle, fill)       */
/* ------------------------------------------------------------------------- */
static void hline(uint8_t* img, int w, int h, int y, int x0, int x1, uint8_t color) {
  if ((unsigned)y >= (unsigned)h) return;
  if (x0 < 0) x0 = 0;
  if (x1 > w - 1) x1 = w - 1;
  for (int x = x0; x <= x1; x++) img[(size_t)y * w + x] = color;
}

void vo_oracle_circle(uint8_t* img, int w, int h, int cx, int cy, int radius, int color) {
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    int mask;
    hline(img, w, h, cy - dy, cx - dx, cx + dx, (uint8_t)color);
    hline(img, w, h, cy + dy, cx - dx, cx + dx, (uint8_t)color);
    hline(img, w, h, cy - dx, cx - dy, cx + dy, (uint8_t)color);
    hline(img, w, h, cy + dx, cx - dy, cx + dy, (uint8_t)color);
    dy++;
    err += plus;
    plus += 2;
    mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

/* half-width of the filled disc at row offset |dy| (or -1 if the row is empty);
 * used by tests to check the GPU rasteriser's row table. */
void vo_oracle_circle_rows(int radius, int* halfw /* radius+1 entries */) {
  for (int i = 0; i <= radius; i++) halfw[i] = -1;
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    int mask;
    if (dx > halfw[dy]) halfw[dy] = dx;
    if (dy > halfw[dx]) halfw[dx] = dy;
    dy++;
    err += plus;
    plus += 2;
    mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

/* ------------------------------------------------------------------------- */
/* cornerMinEigenVal(img, blockSize, ksize=3) (imgproc/corner.cpp)             */
/* ------------------------------------------------------------------------- */
/* harris != 0: cornerHarris(img, blockSize, 3, k) -- the response goodFeaturesToTrack(useHarrisDetector=True, k) ranks (the reference
 * leaves it off, extractor.py:21-24).  Same Sobel scale and box sums (cornerEigenValsVecs), then calcHarris
 * (scalar form): a = cov_xx, b = cov_xy, c = cov_yy WITHOUT the halves, dst = (float)(a * c - b * b - k * (a + c) * (a + c)) with float
 * a, b, c and double k -- the first difference is float arithmetic, the k term double.
 * NB this is the SCALAR tail of calcHarris.  A cv2 built with SIMD (every stock wheel) runs all but the last width % lanes pixels of a row
 * through the vector form -- float k, (a * c - b * b) - (kf * (a + c)) * (a + c) entirely in float -- which differs from this form by ~1 ulp and
 * depends on the build's vector width (4 / 8 / 16 lanes).  Harris parity is therefore pinned to THIS restatement, not to cv2 bit for bit; the
 * reference never turns the option on.                                                                                            */
static inline float corner_value(float sxx, float sxy, float syy, int harris, double k) {
  if (harris) { const float a = sxx, b = sxy, c = syy; return (float)((double)(a * c - b * b) - k * (double)(a + c) * (double)(a + c)); }
  const float a = sxx * 0.5f, b = sxy, c = syy * 0.5f;
  return (a + c) - sqrtf((a - c) * (a - c) + b * b);
}
void vo_oracle_corner_response(const uint8_t* img, int w, int h, int block, float* eig, int exact_int, int harris, double harris_k);
void vo_oracle_min_eig(const uint8_t* img, int w, int h, int block, float* eig, int exact_int) {
  vo_oracle_corner_response(img, w, h, block, eig, exact_int, 0, 0.04);
}
void vo_oracle_corner_response(const uint8_t* img, int w, int h, int block, float* eig, int exact_int, int harris, double harris_k) {
  const double scale_d = 1.0 / ((double)(1 << 2) * block * 255.0);
  const float sf = (float)scale_d;
  const int r = block / 2; /* anchor = centre; block odd */
  size_t np = (size_t)w * h;
  if (exact_int) {
    int32_t* pxx = (int32_t*)malloc(np * sizeof(int32_t));
    int32_t* pxy = (int32_t*)malloc(np * sizeof(int32_t));
    int32_t* pyy = (int32_t*)malloc(np * sizeof(int32_t));
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
#define P(dx_, dy_) pix101(img, w, h, x + (dx_), y + (dy_))
        int dx = (P(1, -1) - P(-1, -1)) + 2 * (P(1, 0) - P(-1, 0)) + (P(1, 1) - P(-1, 1));
        int dy = (P(-1, 1) - P(-1, -1)) + 2 * (P(0, 1) - P(0, -1)) + (P(1, 1) - P(1, -1));
#undef P
        pxx[(size_t)y * w + x] = dx * dx;
        pxy[(size_t)y * w + x] = dx * dy;
        pyy[(size_t)y * w + x] = dy * dy;
      }
    /* separable exact box sum with reflect-101 on the PRODUCT images */
    int32_t* hxx = (int32_t*)malloc(np * sizeof(int32_t));
    int32_t* hxy = (int32_t*)malloc(np * sizeof(int32_t));
    int32_t* hyy = (int32_t*)malloc(np * sizeof(int32_t));
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int32_t a = 0, b = 0, c = 0;
        for (int i = -r; i < block - r; i++) {
          size_t q = (size_t)y * w + reflect101(x + i, w);
          a += pxx[q]; b += pxy[q]; c += pyy[q];
        }
        hxx[(size_t)y * w + x] = a; hxy[(size_t)y * w + x] = b; hyy[(size_t)y * w + x] = c;
      }
    const float s2 = sf * sf;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int32_t sa = 0, sb = 0, sc = 0;
        for (int j = -r; j < block - r; j++) {
          size_t q = (size_t)reflect101(y + j, h) * w + x;
          sa += hxx[q]; sb += hxy[q]; sc += hyy[q];
        }
        eig[(size_t)y * w + x] = corner_value((float)sa * s2, (float)sb * s2, (float)sc * s2, harris, harris_k);
      }
    free(pxx); free(pxy); free(pyy); free(hxx); free(hxy); free(hyy);
  } else {
    /* float path in OpenCV's order: Sobel (scale folded into the smoothing
     * kernel), cov = products, boxFilter(normalize=false) as RowSum + ColumnSum
     * sliding sums in float. */
    float* cov = (float*)malloc(np * 3 * sizeof(float));
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
#define P(dx_, dy_) pix101(img, w, h, x + (dx_), y + (dy_))
        /* Dx: row filter [-1 0 1] (exact), column filter [1 2 1]*scale */
        float r0 = (float)(P(1, -1) - P(-1, -1)), r1 = (float)(P(1, 0) - P(-1, 0)), r2 = (float)(P(1, 1) - P(-1, 1));
        float dx = (2.f * sf) * r1 + sf * (r0 + r2);
        /* Dy: row filter [1 2 1]*scale (float), column filter [-1 0 1] */
        float s0 = (2.f * sf) * (float)P(0, -1) + sf * ((float)P(-1, -1) + (float)P(1, -1));
        float s2r = (2.f * sf) * (float)P(0, 1) + sf * ((float)P(-1, 1) + (float)P(1, 1));
        float dy = s2r - s0;
#undef P
        float* c = cov + ((size_t)y * w + x) * 3;
        c[0] = dx * dx; c[1] = dx * dy; c[2] = dy * dy;
      }
    float* rs = (float*)malloc(np * 3 * sizeof(float));
    for (int y = 0; y < h; y++)
      for (int ch = 0; ch < 3; ch++) {
        float s = 0;
        for (int i = 0; i < block; i++) s += cov[((size_t)y * w + reflect101(i - r, w)) * 3 + ch];
        rs[((size_t)y * w) * 3 + ch] = s;
        for (int x = 0; x < w - 1; x++) {
          s += cov[((size_t)y * w + reflect101(x + block - r, w)) * 3 + ch] - cov[((size_t)y * w + reflect101(x - r, w)) * 3 + ch];
          rs[((size_t)y * w + x + 1) * 3 + ch] = s;
        }
      }
    float* sum = (float*)calloc((size_t)w * 3, sizeof(float));
    for (int j = 0; j < block - 1; j++) {
      const float* rowp = rs + (size_t)reflect101(j - r, h) * w * 3;
      for (int i = 0; i < w * 3; i++) sum[i] += rowp[i];
    }
    for (int y = 0; y < h; y++) {
      const float* sp = rs + (size_t)reflect101(y + block - 1 - r, h) * w * 3;
      const float* sm = rs + (size_t)reflect101(y - r, h) * w * 3;
      for (int x = 0; x < w; x++) {
        float v[3];
        for (int ch = 0; ch < 3; ch++) {
          float s0 = sum[x * 3 + ch] + sp[x * 3 + ch];
          v[ch] = s0;
          sum[x * 3 + ch] = s0 - sm[x * 3 + ch];
        }
        eig[(size_t)y * w + x] = corner_value(v[0], v[1], v[2], harris, harris_k);
      }
    }
    free(cov); free(rs); free(sum);
  }
}

/* ------------------------------------------------------------------------- */
/* goodFeaturesToTrack (imgproc/featureselect.cpp), useHarris = false          */
/* ------------------------------------------------------------------------- */
typedef struct { float v; int idx; } cand_t;
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return (a->idx > b->idx) ? -1 : (a->idx < b->idx) ? 1 : 0; /* higher address first */
}

/*
 * mask may be NULL.  out_xy: max_corners x 2 float (integer-valued x, y).
 * eig_buf (optional, w*h floats) receives the min-eigenvalue map.
 * n_cand_out (optional) receives the number of NMS candidates before selection.
 * returns the number of corners.
 */
int vo_oracle_good_features(const uint8_t* img, const uint8_t* mask, int w, int h, int max_corners,
                            double quality, double min_distance, int block, int exact_int,
                            float* out_xy, float* eig_buf, int* n_cand_out, int use_harris, double harris_k) {
  size_t np = (size_t)w * h;
  float* eig = eig_buf ? eig_buf : (float*)malloc(np * sizeof(float));
  vo_oracle_corner_response(img, w, h, block, eig, exact_int, use_harris, harris_k);
  double maxVal = 0; /* minMaxLoc over mask != 0 */
  int any = 0;
  for (size_t i = 0; i < np; i++)
    if (!mask || mask[i]) { if (!any || eig[i] > maxVal) { maxVal = eig[i]; any = 1; } }
  float thr = (float)(maxVal * quality);
  float* te = (float*)malloc(np * sizeof(float));
  for (size_t i = 0; i < np; i++) te[i] = eig[i] > thr ? eig[i] : 0.f;
  cand_t* cands = (cand_t*)malloc(np * sizeof(cand_t));
  int nc = 0;
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      float v = te[(size_t)y * w + x];
      if (v == 0.f) continue;
      float m = v;
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++) { float q = te[(size_t)(y + j) * w + x + i]; if (q > m) m = q; }
      if (v == m && (!mask || mask[(size_t)y * w + x])) { cands[nc].v = v; cands[nc].idx = y * w + x; nc++; }
    }
  if (n_cand_out) *n_cand_out = nc;
  qsort(cands, nc, sizeof(cand_t), cand_cmp);
  int ncorners = 0;
  if (min_distance >= 1) {
    const int cell = (int)lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    /* grid cells as linked lists */
    int* head = (int*)malloc((size_t)gw * gh * sizeof(int));
    int* next = (int*)malloc((size_t)(max_corners > 0 ? max_corners : nc) * sizeof(int) + sizeof(int));
    for (int i = 0; i < gw * gh; i++) head[i] = -1;
    const double md2 = min_distance * min_distance; /* OpenCV: minDistance *= minDistance (double) */
    for (int i = 0; i < nc; i++) {
      int y = cands[i].idx / w, x = cands[i].idx - y * w;
      int xc = x / cell, yc = y / cell;
      int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1;
      int x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
      int good = 1;
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++)
          for (int q = head[yy * gw + xx]; q >= 0; q = next[q]) {
            float ddx = (float)x - out_xy[2 * q], ddy = (float)y - out_xy[2 * q + 1];
            if ((double)(ddx * ddx + ddy * ddy) < md2) { good = 0; break; }
          }
      if (good) {
        out_xy[2 * ncorners] = (float)x; out_xy[2 * ncorners + 1] = (float)y;
        next[ncorners] = head[yc * gw + xc]; head[yc * gw + xc] = ncorners;
        ncorners++;
        if (max_corners > 0 && ncorners == max_corners) break;
      }
    }
    free(head); free(next);
  } else {
    for (int i = 0; i < nc; i++) {
      int y = cands[i].idx / w, x = cands[i].idx - y * w;
      out_xy[2 * ncorners] = (float)x; out_xy[2 * ncorners + 1] = (float)y;
      ncorners++;
      if (max_corners > 0 && ncorners == max_corners) break;
    }
  }
  free(cands); free(te);
  if (!eig_buf) free(eig);
  return ncorners;
}

/* ------------------------------------------------------------------------- */
/* triangulatePoints (calib3d/triangulate.cpp): per point 4x4 A (double),      */
/* SVD, last right-singular vector.  One-sided (Hestenes) Jacobi in double.    */
/* ------------------------------------------------------------------------- */
static void svd4_smallest(const double A[4][4], double v_out[4]) {
  double U[4][4], V[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { U[i][j] = A[i][j]; V[i][j] = (i == j); }
  for (int sweep = 0; sweep < 60; sweep++) {
    int changed = 0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 4; k++) { al += U[k][p] * U[k][p]; be += U[k][q] * U[k][q]; ga += U[k][p] * U[k][q]; }
        if (fabs(ga) <= 2.220446049250313e-16 * sqrt(al * be)) continue;
        changed = 1;
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 4; k++) {
          double up = U[k][p], uq = U[k][q];
          U[k][p] = c * up - s * uq; U[k][q] = s * up + c * uq;
          double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
        }
      }
    if (!changed) break;
  }
  int best = 0; double bn = 0;
  for (int j = 0; j < 4; j++) {
    double nn = 0;
    for (int k = 0; k < 4; k++) nn += U[k][j] * U[k][j];
    if (j == 0 || nn < bn) { bn = nn; best = j; }
  }
  for (int k = 0; k < 4; k++) v_out[k] = V[k][best];
}

/* P0, P1: 3x4 float32 row-major; uv0, uv1: n x 2 float32; X4: 4 x n float32 (OpenCV layout) */
void vo_oracle_triangulate(const float* P0, const float* P1, const float* uv0, const float* uv1, int n, float* X4) {
  for (int i = 0; i < n; i++) {
    double A[4][4], v[4];
    for (int view = 0; view < 2; view++) {
      const float* P = view ? P1 : P0;
      double x = view ? uv1[2 * i] : uv0[2 * i], y = view ? uv1[2 * i + 1] : uv0[2 * i + 1];
      for (int k = 0; k < 4; k++) {
        A[view * 2 + 0][k] = x * (double)P[8 + k] - (double)P[k];
        A[view * 2 + 1][k] = y * (double)P[8 + k] - (double)P[4 + k];
      }
    }
    svd4_smallest(A, v);
    for (int k = 0; k < 4; k++) X4[(size_t)k * n + i] = (float)v[k];
  }
}

/* ------------------------------------------------------------------------------------------------
 * cv2.bilateralFilter(src, d, sigmaColor, sigmaSpace) for 8-bit single-channel images, borderType
 * BORDER_DEFAULT (reflect-101) -- OpenCV 4.4 imgproc/bilateral_filter.dispatch.cpp (bilateralFilter_8u) and the
 * scalar form of BilateralFilter_8u_Invoker: radius = d/2 (d > 0) else round(1.5 sigmaSpace), at least 1; the taps
 * are the offsets (i, j) of the (2 radius + 1)^2 square with sqrt(i^2 + j^2) <= radius, rows outer / columns inner;
 * space_weight[k] = (float)exp(r^2 * -0.5 / sigmaSpace^2), color_weight[a] = (float)exp(a^2 * -0.5 / sigmaColor^2)
 * (double exp, rounded to float); per pixel, in tap order and in float:
 *     w = space_weight[k] * color_weight[|val - val0|];  sum += val * w;  wsum += w;
 * dst = (uchar) cvRound(sum / wsum)  (round half to even).
 * (OpenCV's SIMD rows use v_muladd, i.e. an FMA where the build has one; this restatement is the scalar,
 * unfused order -- parity at this boundary is unpinned like the other OpenCV calls.)
 * ------------------------------------------------------------------------------------------------ */
int vo_oracle_bilateral_taps(int d, double sigma_color, double sigma_space, int* ofs_xy /* 2 x max taps */,
                             float* space_w, float* color_w /* 256 */) {
  if (sigma_color <= 0) sigma_color = 1;
  if (sigma_space <= 0) sigma_space = 1;
  const double gc = -0.5 / (sigma_color * sigma_color), gs = -0.5 / (sigma_space * sigma_space);
  int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
  if (radius < 1) radius = 1;
  for (int i = 0; i < 256; i++) color_w[i] = (float)exp((double)i * i * gc);
  int maxk = 0;
  for (int i = -radius; i <= radius; i++)
    for (int j = -radius; j <= radius; j++) {
      const double r = sqrt((double)i * i + (double)j * j);
      if (r > radius) continue;
      space_w[maxk] = (float)exp(r * r * gs);
      ofs_xy[2 * maxk] = j; ofs_xy[2 * maxk + 1] = i;
      maxk++;
    }
  return maxk;
}

int vo_oracle_bilateral(const uint8_t* src, int w, int h, int d, double sigma_color, double sigma_space, uint8_t* dst) {
  int radius = d <= 0 ? (int)lrint((sigma_space <= 0 ? 1 : sigma_space) * 1.5) : d / 2;
  if (radius < 1) radius = 1;
  const int side = 2 * radius + 1;
  int* ofs = (int*)malloc(sizeof(int) * 2 * side * side);
  float* sw = (float*)malloc(sizeof(float) * side * side);
  float cw[256];
  if (!ofs || !sw) { free(ofs); free(sw); return -1; }
  const int maxk = vo_oracle_bilateral_taps(d, sigma_color, sigma_space, ofs, sw, cw);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int val0 = src[(size_t)y * w + x];
      float sum = 0.f, wsum = 0.f;
      for (int k = 0; k < maxk; k++) {
        const int val = pix101(src, w, h, x + ofs[2 * k], y + ofs[2 * k + 1]);
        const float wt = sw[k] * cw[abs(val - val0)];
        sum += (float)val * wt;
        wsum += wt;
      }
      dst[(size_t)y * w + x] = (uint8_t)lrintf(sum / wsum);
    }
  free(ofs); free(sw);
  return maxk;
}
