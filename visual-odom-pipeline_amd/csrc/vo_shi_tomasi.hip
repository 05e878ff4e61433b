// Shi-Tomasi re-detection on the current frame: exclusion mask + goodFeaturesToTrack, all on device.
//
// Replaces /root/reference/src/extractor/extractor.py:103-112:
//     mask = 255; for kp: cv2.circle(mask, int32(kp.uv), mask_radius, 0, -1)
//     cv2.goodFeaturesToTrack(img, mask=mask, maxCorners=1000, qualityLevel=0.03, minDistance=7, blockSize=31)
// Algorithm = OpenCV 4.4 imgproc/featureselect.cpp + corner.cpp + drawing.cpp.
// The 31x31 structure-tensor sums are exact int32 (Sobel outputs are integers before the scale), so
// the eigenvalue map, the candidate ranking and the selected corners are bit-identical to
// oracle/vo_oracle.c (exact_int = 1).
//
// Launches (one stream, no host round trip):
//   k_st_mask_init  : mask = 255 (or the caller's mask), scalars zeroed
//   k_st_discs      : filled midpoint circles of the tracked keypoints into the mask
//   k_st_eig_fused  : Sobel products, 31 x 31 box sums, min-eigenvalue map + masked maxima in one pass (block size 31;
//                     other block sizes, the Harris response, tiny images or vo_tuning.st_two_kernels: k_st_sobel_hsum -> 3 int32 planes -> k_st_vsum_eig)
//   k_st_nms        : threshold + 3x3 non-max suppression + mask -> compacted (value, index) keys
//   k_st_select     : ONE workgroup: bitonic sort of the keys in LDS (value desc, index desc) and the
//                     greedy min-distance selection done as parallel fixed-point rounds (a candidate is
//                     accepted iff every higher-ranked candidate within min_distance is rejected) --
//                     equivalent to OpenCV's sequential grid scan, output in rank order, first maxCorners.
#include "vo_internal.h"

#include <type_traits>

#define ST_CAND_CAP 16384        // keys sorted in LDS (128 KB of the CU's 160 KB)
#define ST_GLOBAL_CAP (1 << 18)  // NMS candidates kept per sequence in HBM; above ST_CAND_CAP the selection works on the
                                 // ST_CAND_CAP strongest (radix select) and is exact whenever it fills max_corners
#define ST_CAND_STRIDE (ST_GLOBAL_CAP + ST_CAND_CAP)   // raw list (conservative threshold) | staging of one rank-ordered chunk
#define ST_OUT_CAP 4096
#define ST_MAX_RADIUS 31

struct vo_st_ws {
  uint8_t* d_mask = nullptr;
  uint8_t* d_user_mask = nullptr;
  int32_t* d_h = nullptr;          // 3 planes W*H: hxx, hxy, hyy
  float* d_eig = nullptr;
  uint32_t* d_scalars = nullptr;   // [0] max eig bits, [1] n candidates, [2] n out (int), [3] rounds
  unsigned long long* d_cand = nullptr;
  uint32_t* d_nraw = nullptr;      // [batch] raw candidate counters (appended to by the NMS stage, re-armed by k_st_select)
  bool mask_clean = false;         // the mask is all 255 (k_st_eig_fused restored it): the next resident launch needs no k_st_mask_init
  bool eig_valid = false;          // the last launch stored the eigenvalue map (vo_shi_tomasi_read)
  float* d_blockmax = nullptr;     // per-workgroup masked maxima of the eigenvalue pass
  float* d_out = nullptr;          // ST_OUT_CAP x 2
  float* d_pts = nullptr;          // uploaded cur_pts (non-resident call)
  int n_blockmax = 0;
  int last_max_corners = 0;
};

struct disc_rows { int hw[ST_MAX_RADIUS + 1]; };

__device__ __forceinline__ int st_reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

// grid (x blocks, rows, batch); per-sequence strides: img_seq_px pixels, plane = W * H elements
#define ST_HS_RUN 5                      // consecutive outputs per thread in the row-sum pass (odd stride: conflict-free LDS)
#define ST_HS_COLS (256 * ST_HS_RUN)     // columns per workgroup
__global__ void __launch_bounds__(256) k_st_sobel_hsum(const uint8_t* __restrict__ img, size_t img_seq_px, int pitch, int W, int H, int r,
                                                       int32_t* __restrict__ hbase, uint8_t* __restrict__ mask,
                                                       const uint8_t* __restrict__ user_mask,
                                                       uint32_t* __restrict__ scalars, size_t slab_seq) {
  // products of one row segment (+ the box radius on both sides), then SLIDING row sums: a thread owns ST_HS_RUN
  // consecutive outputs, so it reads 2 r + 1 + 2 (RUN - 1) products per plane instead of RUN (2 r + 1); the sums go
  // back through LDS so that the global stores are coalesced
  __shared__ int32_t sxx[ST_HS_COLS + 2 * 15 + 2], sxy[ST_HS_COLS + 2 * 15 + 2], syy[ST_HS_COLS + 2 * 15 + 2];
  const int y = blockIdx.y, x0 = blockIdx.x * ST_HS_COLS, t = threadIdx.x;
  const int bseq = blockIdx.z;
  const size_t np = (size_t)W * H;
  img += (size_t)bseq * img_seq_px;
  int32_t* hxx = hbase + (size_t)bseq * 3 * np; int32_t* hxy = hxx + np; int32_t* hyy = hxy + np;
  mask += (size_t)bseq * np;
  if (user_mask) user_mask += (size_t)bseq * np;
  scalars = vo_seq(scalars, slab_seq, bseq);
  if (blockIdx.x == 0 && blockIdx.y == 0 && t < 4) scalars[t] = 0;
  const int ncol = min(ST_HS_COLS, W - x0);          // outputs of this workgroup
  const int span = ncol + 2 * r;
  for (int i = t; i < span; i += 256) {
    const int xs = st_reflect101(x0 - r + i, W);   // box filter reflects the PRODUCT image
    const uint8_t* p = img + (size_t)(y + VO_PAD) * pitch + (xs + VO_PAD);
    const int a00 = p[-pitch - 1], a01 = p[-pitch], a02 = p[-pitch + 1];
    const int a10 = p[-1], a12 = p[1];
    const int a20 = p[pitch - 1], a21 = p[pitch], a22 = p[pitch + 1];
    const int dx = (a02 - a00) + 2 * (a12 - a10) + (a22 - a20);
    const int dy = (a20 - a00) + 2 * (a21 - a01) + (a22 - a02);
    sxx[i] = dx * dx; sxy[i] = dx * dy; syy[i] = dy * dy;
  }
  __syncthreads();
  int32_t oa[ST_HS_RUN], ob[ST_HS_RUN], oc[ST_HS_RUN];
  const int c0 = t * ST_HS_RUN;                      // first output column (local) of this thread
  if (c0 < ncol) {
    int32_t a = 0, b = 0, c = 0;
    for (int i = 0; i <= 2 * r; i++) { a += sxx[c0 + i]; b += sxy[c0 + i]; c += syy[c0 + i]; }
    oa[0] = a; ob[0] = b; oc[0] = c;
#pragma unroll
    for (int k = 1; k < ST_HS_RUN; k++) {
      // columns past the end of the row read stale LDS; those outputs are never stored
      a += sxx[c0 + 2 * r + k] - sxx[c0 + k - 1]; b += sxy[c0 + 2 * r + k] - sxy[c0 + k - 1]; c += syy[c0 + 2 * r + k] - syy[c0 + k - 1];
      oa[k] = a; ob[k] = b; oc[k] = c;
    }
  }
  __syncthreads();                                   // every product has been consumed: reuse the arrays for the sums
  if (c0 < ncol) {
#pragma unroll
    for (int k = 0; k < ST_HS_RUN; k++) { sxx[c0 + k] = oa[k]; sxy[c0 + k] = ob[k]; syy[c0 + k] = oc[k]; }
  }
  __syncthreads();
  for (int i = t; i < ncol; i += 256) {
    const size_t o = (size_t)y * W + x0 + i;
    hxx[o] = sxx[i]; hxy[o] = sxy[i]; hyy[o] = syy[i];
    mask[o] = user_mask ? user_mask[o] : (uint8_t)255;
  }
}

// grid (blocks, batch); pts of sequence b at + b * pts_seq bytes
__global__ void __launch_bounds__(256) k_st_discs(const float* __restrict__ pts, size_t pts_seq, int n, int radius, disc_rows rows,
                                                  uint8_t* __restrict__ mask, int W, int H, const int32_t* __restrict__ counts) {
  // (16 lanes per disc row -- lane = column, so that a store instruction covers four 15-byte row segments instead of one byte in each of 64 rows --
  //  was measured in round 4: 54.8 us instead of 38.4 per 32 x 2 048 discs; sixteen times the waves cost more than the scattered stores.
  //  What pays is fewer store instructions per lane: see the span stores below)
  const int nrows = 2 * radius + 1;
  pts = vo_seq(pts, pts_seq, blockIdx.y);
  mask += (size_t)blockIdx.y * W * H;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * nrows) return;
  const int k = gid / nrows, dy = gid - k * nrows - radius;
  if (counts && k >= counts[blockIdx.y]) return;
  const int cx = (int)pts[2 * k], cy = (int)pts[2 * k + 1];   // np.int32(): truncation toward zero
  const int y = cy + dy;
  if (y < 0 || y >= H) return;
  const int hw = rows.hw[dy < 0 ? -dy : dy];
  if (hw < 0) return;
  int xa = cx - hw, xb = cx + hw;
  if (xa < 0) xa = 0;
  if (xb > W - 1) xb = W - 1;
  // the span [xa, xb] as OVERLAPPING unaligned wide stores of zeros (a disc row of radius 7 is 15 bytes: two 8-byte stores instead of 15
  // byte stores, each of which touched a different cache line in every lane of the wave: 38 us per 32 x 2 048 discs)
  uint8_t* const p = mask + (size_t)y * W + xa;
  const int len = xb - xa + 1;
  if (len <= 0) return;
  const unsigned long long z8 = 0ull; const uint32_t z4 = 0u; const unsigned short z2 = 0;
  if (len >= 8) {
    for (int o = 0; o + 8 <= len; o += 8) __builtin_memcpy(p + o, &z8, 8);
    __builtin_memcpy(p + len - 8, &z8, 8);
  } else if (len >= 4) {
    __builtin_memcpy(p, &z4, 4); __builtin_memcpy(p + len - 4, &z4, 4);
  } else if (len >= 2) {
    __builtin_memcpy(p, &z2, 2); __builtin_memcpy(p + len - 2, &z2, 2);
  } else {
    p[0] = 0;
  }
}

#define ST_RG 8       // output rows per load group in the vertical pass
#define ST_GROUPS 6   // groups per thread: a thread walks 48 rows with running sums, so the 31 start-up rows are read once
                      // per 48 outputs (7.8 loads per pixel instead of 16.9 at 8 rows; the pass is HBM / MALL bound)
#define ST_ROWS (ST_RG * ST_GROUPS)
// harris != 0: the response is OpenCV's calcHarris instead of calcMinEigenVal (imgproc/corner.cpp; goodFeaturesToTrack(useHarrisDetector=True,
// k), the option the reference's parameter dict extractor.py:21-24 leaves off): a c - b^2 - k (a + c)^2 on the box sums WITHOUT the halves,
// evaluated as the scalar C++ expression is -- a c - b^2 in float, the k term in double (k is a double), one rounding to float
__global__ void __launch_bounds__(256) k_st_vsum_eig(const int32_t* __restrict__ hbase, const uint8_t* __restrict__ mask,
                                                     int W, int H, int r, float s2, float* __restrict__ eig,
                                                     float* __restrict__ blockmax, int harris, double harris_k) {
  __shared__ float s_m[4];
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y0 = blockIdx.y * ST_ROWS;
  const int bseq = blockIdx.z;
  const size_t np = (size_t)W * H;
  const int32_t* hxx = hbase + (size_t)bseq * 3 * np; const int32_t* hxy = hxx + np; const int32_t* hyy = hxy + np;
  mask += (size_t)bseq * np; eig += (size_t)bseq * np;
  blockmax += (size_t)bseq * gridDim.x * gridDim.y;
  float lmax = 0.f;
  if (x < W) {
    int32_t sa = 0, sb = 0, sc = 0;
    int j = -r;
    for (; j + 3 <= r; j += 4) {   // 12 independent loads in flight per step
      const size_t o0 = (size_t)st_reflect101(y0 + j, H) * W + x, o1 = (size_t)st_reflect101(y0 + j + 1, H) * W + x;
      const size_t o2 = (size_t)st_reflect101(y0 + j + 2, H) * W + x, o3 = (size_t)st_reflect101(y0 + j + 3, H) * W + x;
      const int32_t a0 = hxx[o0], a1 = hxx[o1], a2 = hxx[o2], a3 = hxx[o3];
      const int32_t b0 = hxy[o0], b1 = hxy[o1], b2 = hxy[o2], b3 = hxy[o3];
      const int32_t c0 = hyy[o0], c1 = hyy[o1], c2 = hyy[o2], c3 = hyy[o3];
      sa += (a0 + a1) + (a2 + a3); sb += (b0 + b1) + (b2 + b3); sc += (c0 + c1) + (c2 + c3);
    }
    for (; j <= r; j++) {
      const size_t o = (size_t)st_reflect101(y0 + j, H) * W + x;
      sa += hxx[o]; sb += hxy[o]; sc += hyy[o];
    }
    // sliding window: the 6 loads per row are independent of the running sums -> a group's loads are issued up front
    for (int g = 0; g < ST_GROUPS; g++) {
      const int yb = y0 + g * ST_RG;
      if (yb >= H) break;
      int32_t da[ST_RG], db[ST_RG], dc[ST_RG];
      uint8_t mk[ST_RG];
#pragma unroll
      for (int k = 0; k < ST_RG; k++) {
        const int y = min(yb + k, H - 1);
        mk[k] = mask[(size_t)y * W + x];
        if (g == 0 && k == 0) { da[0] = db[0] = dc[0] = 0; continue; }
        const size_t on = (size_t)st_reflect101(y + r, H) * W + x, oo = (size_t)st_reflect101(y - r - 1, H) * W + x;
        da[k] = hxx[on] - hxx[oo]; db[k] = hxy[on] - hxy[oo]; dc[k] = hyy[on] - hyy[oo];
      }
#pragma unroll
      for (int k = 0; k < ST_RG; k++) {
        const int y = yb + k;
        sa += da[k]; sb += db[k]; sc += dc[k];
        if (y < H) {
          float e;
          if (harris) {
            const float a = (float)sa * s2, b = (float)sb * s2, c = (float)sc * s2;
            e = (float)((double)(a * c - b * b) - harris_k * (double)(a + c) * (double)(a + c));
          } else {
            const float a = ((float)sa * s2) * 0.5f, b = (float)sb * s2, c = ((float)sc * s2) * 0.5f;
            e = (a + c) - sqrtf((a - c) * (a - c) + b * b);
          }
          eig[(size_t)y * W + x] = e;
          if (mk[k] && e > lmax) lmax = e;
        }
      }
    }
  }
  // block max -> one float per block (the next kernel reduces them; max is order independent)
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = lmax;
  __syncthreads();
  if (threadIdx.x == 0) blockmax[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// ------------------------------------------------------------------------------------------------
// k_st_eig_fused : Sobel products, 31 x 31 box sums and the min-eigenvalue map in ONE pass over the image.
// The two-kernel form above moves 12 B/px of row sums through HBM twice (0.19 GB written + 0.62 GB read per batched
// launch); this one reads 1 B/px and writes 4.  A workgroup owns 256 - 2 R product columns x `rb` output rows and walks
// down the band: a thread = one product column (reflected like the box filter reflects the product image); the 3-row
// Sobel window rolls through registers (3 byte loads per row), the VERTICAL box sum is a running sum over a ring of the
// last 2 R + 1 products held in registers (ring index = unrolled loop counter), the HORIZONTAL box sum of the running
// sums is a wave prefix scan (DPP) + one LDS exchange: S(x) = P(x + R) - P(x - R - 1), wave totals added where the
// window straddles two waves.  All sums are uint32 modulo 2^32 -- the prefix overflows, the window sum (< 2^31) does not --
// so the result is bit-identical to the two-kernel form and to the oracle.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t st_ld_u32_any(const uint8_t* p) {     // unaligned dword load (one global_load_dword)
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// correctly rounded sqrtf for x = 0 or x >= 2^-100 (here: sums of squares of differences of quantised products, 0 or >= 1e-19):
// v_sqrt_f32 is within one ulp; the two fused residuals pick the neighbour when it is the nearest -- the compiler's own sequence
// for sqrtf without its denormal pre-scaling and class test (9 instead of 18 instructions; x = 0 falls through: the lower
// "neighbour" is a NaN, the comparisons fail)
__device__ __forceinline__ float st_sqrt_rn(float x) {
  float y = __builtin_amdgcn_sqrtf(x);
  const float ym = __int_as_float(__float_as_int(y) - 1), yp = __int_as_float(__float_as_int(y) + 1);
  const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
  y = (rm <= 0.f) ? ym : y;
  y = (rp > 0.f) ? yp : y;
  return y;
}

#define ST_DPP_ADD(v, ctrl, rmask, bc) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, bc)
// inclusive wave prefix sums of three values; the three chains are interleaved so that each DPP read is two VALU
// instructions behind the write it depends on (no hazard s_nops)
__device__ __forceinline__ void st_scan64x3(unsigned& a, unsigned& b, unsigned& c) {
  ST_DPP_ADD(a, 0x111, 0xf, true); ST_DPP_ADD(b, 0x111, 0xf, true); ST_DPP_ADD(c, 0x111, 0xf, true);      // row_shr:1
  ST_DPP_ADD(a, 0x112, 0xf, true); ST_DPP_ADD(b, 0x112, 0xf, true); ST_DPP_ADD(c, 0x112, 0xf, true);      // row_shr:2
  ST_DPP_ADD(a, 0x114, 0xf, true); ST_DPP_ADD(b, 0x114, 0xf, true); ST_DPP_ADD(c, 0x114, 0xf, true);      // row_shr:4
  ST_DPP_ADD(a, 0x118, 0xf, true); ST_DPP_ADD(b, 0x118, 0xf, true); ST_DPP_ADD(c, 0x118, 0xf, true);      // row_shr:8
  ST_DPP_ADD(a, 0x142, 0xa, false); ST_DPP_ADD(b, 0x142, 0xa, false); ST_DPP_ADD(c, 0x142, 0xa, false);   // row_bcast:15 -> rows 1, 3
  ST_DPP_ADD(a, 0x143, 0xc, false); ST_DPP_ADD(b, 0x143, 0xc, false); ST_DPP_ADD(c, 0x143, 0xc, false);   // row_bcast:31 -> rows 2, 3
}

__global__ void __launch_bounds__(256) k_st_mask_init(uint8_t* __restrict__ mask, const uint8_t* __restrict__ user_mask, size_t np,
                                                      uint32_t* __restrict__ scalars, size_t slab_seq) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;      // 16 bytes per thread (np * batch offsets keep 4-byte alignment only)
  const int bseq = blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x < 4) vo_seq(scalars, slab_seq, bseq)[threadIdx.x] = 0;
  mask += (size_t)bseq * np;
  if (i >= np) return;
  if (!user_mask && ((reinterpret_cast<uintptr_t>(mask + i) & 3) == 0) && i + 16 <= np) {
    uint32_t* m4 = reinterpret_cast<uint32_t*>(mask + i);
    m4[0] = 0xFFFFFFFFu; m4[1] = 0xFFFFFFFFu; m4[2] = 0xFFFFFFFFu; m4[3] = 0xFFFFFFFFu;
    return;
  }
  if (user_mask) user_mask += (size_t)bseq * np;
  for (int k = 0; k < 16; k++)
    if (i + k < np) mask[i + k] = user_mask ? user_mask[i + k] : (uint8_t)255;
}

// k_st_eig_fused also does the 3 x 3 non-maximum suppression of goodFeaturesToTrack (dilate + compare, featureselect.cpp) and
// appends the surviving pixels to the candidate list, so the eigenvalue map is never re-read (k_st_nms read it back: 107 MB per
// batched launch) and need not be WRITTEN at all unless somebody asks for it (`eig` may be null).
//   * a workgroup owns OUTC = 256 - 2 R - 2 output columns and `rb` output rows; it computes one eigenvalue column / row more
//     on every side (halo), so every 3 x 3 neighbourhood of its outputs is its own;
//   * a thread keeps the last three eigenvalues of its column; the column-wise maxima go through LDS once per row
//     (double-buffered, the existing barrier of the row orders it) and the test of row y runs one row later;
//   * the quality threshold needs the GLOBAL masked maximum, unknown during the pass: candidates are kept against a RUNNING
//     maximum (this column and, spreading one column per row, its neighbours) -- a lower bound, so the list is a superset;
//     k_st_select drops the entries below the true threshold before ranking;
//   * candidates go out with one atomic per wave and row (ballot + prefix), keys (value bits << 32 | pixel index) are unique, the
//     ranking sorts them, so the append order does not matter;
//   * `restore_mask`: the exclusion mask is consumed exactly once per pixel here; writing 255 back leaves it clean for the next
//     frame's discs (no separate k_st_mask_init launch on the resident path).
template <int R>
__global__ void __launch_bounds__(256, 4) k_st_eig_fused(const uint8_t* __restrict__ img, size_t img_seq_px, int pitch, int W, int H, int rb, float s2,
                                                      uint8_t* __restrict__ mask, float* __restrict__ eig, float* __restrict__ blockmax,
                                                      double quality, unsigned long long* __restrict__ cand, uint32_t* __restrict__ nraw,
                                                      int restore_mask, int do_nms, int remap) {
  constexpr int D = 2 * R + 1, OUTC = 256 - 2 * R - 2;
  __shared__ uint4 s_p[2][257];                                       // [256]: zeros (the subtrahend of windows that start at the workgroup's / a wave's edge)
  __shared__ float2 s_c[2][256];                                      // (max of the column's three rows, running masked maximum)
  __shared__ float s_m[4];
  // candidates are collected in LDS (one LDS atomic per wave and row) and flushed with ONE global atomic when the list could
  // overflow on the next row and at the end: a returning global atomic per wave and row stalled the row loop (3.6x slower)
  constexpr int LCAP = 2048;
  __shared__ unsigned long long s_keys[LCAP];
  __shared__ unsigned int s_cnt, s_gbase;
  __shared__ unsigned int s_flag[2];                                  // "empty the list before this row's appends", by row parity
  const int t = threadIdx.x, lane = t & 63;
  if (t == 0) s_cnt = 0;
  if (t < 2) { s_p[t][256] = make_uint4(0u, 0u, 0u, 0u); s_flag[t] = 0u; }
  s_c[0][t] = make_float2(0.f, 0.f); s_c[1][t] = make_float2(0.f, 0.f);    // the first row's (unconditional) neighbour reads
  // (band, sequence) assignment: a sequence's bands run on the XCD that built its pyramid (vo_xcd_assign); blocks of a band adjacent
  int bxy, bseq;
  vo_xcd_assign((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, remap, bxy, bseq);
  const int bandy = bxy / (int)gridDim.x, bandx = bxy - bandy * (int)gridDim.x;
  const int x0 = bandx * OUTC, y0 = bandy * rb;
  const int rows_out = min(rb, H - y0);
  const int total = rows_out + 2 + 2 * R;                             // product rows y0 - 1 - R .. y0 + rows_out + R
  const int xs = st_reflect101(x0 - 1 - R + t, W);                    // product column of this thread
  // image rows and mask bytes come through buffer loads: descriptor base + a per-thread column offset that never changes (vector
  // register) + the row offset (scalar register, scalar arithmetic) -- no vector address arithmetic per row
  const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img + (size_t)bseq * img_seq_px), 0, -1, 0x00020000);
  const int xe = x0 - 1 + t - R;                                      // eigenvalue column of this thread (halo: x0 - 1 and x0 + OUTC)
  const bool evalid = (t >= R) && (t < 256 - R) && xe >= 0 && xe < W; // an eigenvalue of the image is formed here
  const bool outt = evalid && (t > R) && (t < 255 - R);               // ... and it is one of this workgroup's outputs
  const int xo_c = outt ? xe : 0;
  const size_t np = (size_t)W * H;
  mask += (size_t)bseq * np;
  const __amdgpu_buffer_rsrc_t r_mask = __builtin_amdgcn_make_buffer_rsrc(mask, 0, -1, 0x00020000);
  if (eig) eig += (size_t)bseq * np;
  cand += (size_t)bseq * ST_CAND_STRIDE; nraw += bseq;
  // the ring keeps the gradients (dx | dy << 16, |.| <= 1020), not the three products: 31 registers instead of 93 -- the kernel is
  // bound by latency, its time goes with 1 / (waves per SIMD) (measured 170 us at 2, 118 at 3), and the ring decides the register count
  uint32_t ring[D];
#pragma unroll
  for (int u = 0; u < D; u++) ring[u] = 0;
  unsigned V0 = 0, V1 = 0, V2 = 0;
  float lmax = 0.f;                                                   // masked maximum of the outputs of this column (exact part of the global maximum)
  float rmax = 0.f;                                                   // running lower bound of the global maximum used for the candidate threshold
  float e1 = 0.f, e2 = 0.f;                                           // eigenvalues of the two previous rows of this column
  float pv = 0.f, pcm = 0.f; bool pok = false; int py = 0;            // pending 3 x 3 test: centre value, max of the rows above / below, validity, row
  uint8_t mk1 = 0;
  // the three image rows of a product row arrive as unaligned dwords (bytes xs - 1 .. xs + 2), fetched PF product rows
  // ahead: with one round trip to L2 per row and three waves per SIMD the kernel was bound by that latency
  constexpr int PF = 3;
  uint32_t q[PF][3];
  auto fetch = [&](int k, uint32_t (&w)[3]) {
    int rp = y0 - 1 - R + min(k, total - 1);                        // product row (uniform); one reflection is enough: overshoot <= R + 1 < H
    rp = rp < 0 ? -rp : (rp >= H ? 2 * (H - 1) - rp : rp);
    const int o1 = (rp + VO_PAD) * pitch + (VO_PAD - 1);            // padded row of product row rp, one byte left of the thread's column
    w[0] = __builtin_amdgcn_raw_buffer_load_b32(r_img, xs, o1 - pitch, 0);
    w[1] = __builtin_amdgcn_raw_buffer_load_b32(r_img, xs, o1, 0);
    w[2] = __builtin_amdgcn_raw_buffer_load_b32(r_img, xs, o1 + pitch, 0);
  };
  // the pending test of row `py`: neighbours' column maxima of the slot written one row ago.  Branch-free up to the ballot (the
  // LDS reads use clamped indices; `pok` is false wherever they would be meaningless).  The threshold here only thins the list --
  // k_st_select applies the exact one -- so a float product rounded DOWN replaces the float64 product of the reference formula.
  // LDS slots of the horizontal window sum S(x) = P(x + R) - P(x - R - 1) (+ the total of the wave the window starts in)
  const int ia = min(t + R, 255);
  const int ib_z = (t - R - 1 >= 0) ? t - R - 1 : 256;
  const int it_z = (t - R - 1 >= 0 && (ia >> 6) != ((t - R - 1) >> 6)) ? ((t - R - 1) | 63) : 256;
  const float q_lo = (float)(quality * (1.0 - 1e-6));
  const int tl = max(t - 1, 0), tr = min(t + 1, 255);
  auto test_pending = [&](int slot) {
    const float2 l = s_c[slot][tl], r = s_c[slot][tr];
    rmax = fmaxf(rmax, fmaxf(l.y, r.y));
    const float nmax = fmaxf(fmaxf(l.x, r.x), pcm);
    const bool c = pok && (pv > rmax * q_lo) && (pv >= nmax);            // pv > threshold >= 0 excludes the zeros as well
    const unsigned long long bal = __ballot(c);
    if (bal) {
      unsigned int basep = 0;
      if (lane == 0) basep = atomicAdd(&s_cnt, (unsigned int)__popcll(bal));     // LDS
      basep = (unsigned int)__builtin_amdgcn_readfirstlane((int)basep);
      if (c) s_keys[basep + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull))] =
               ((unsigned long long)__float_as_uint(pv) << 32) | (unsigned long long)(uint32_t)((size_t)py * W + xe);
    }
  };
  // Emptying the list.  The decision must be the same in every wave, but a count read after a barrier is not: a faster wave may
  // already have appended this row's candidates (the earlier form read it there: waves could disagree when the list was nearly
  // full).  So thread 0 publishes the decision for the NEXT row (s_flag, double-buffered by row parity; read on the other side of the
  // row's barrier).  The count it saw is at most one row behind the count at the flush, and one more row is appended before the
  // next chance: the list is emptied when it holds more than LCAP - 2 * 256 entries.
  auto flush_now = [&]() {
    __syncthreads();                                                   // every append of the previous row is in
    const unsigned int n = s_cnt;
    if (t == 0 && n) s_gbase = atomicAdd(nraw, n);
    __syncthreads();
    const unsigned int gb = s_gbase;
    for (unsigned int i = t; i < n; i += 256) if (gb + i < ST_GLOBAL_CAP) cand[gb + i] = s_keys[i];
    __syncthreads();
    if (t == 0) s_cnt = 0;
    __syncthreads();
  };
#pragma unroll
  for (int i = 0; i < PF; i++) fetch(i, q[i]);
  auto mask_row = [&](int ke) { return min(max(y0 - 1 + ke, 0), H - 1); };   // image row of eigenvalue row index ke (clamped: halo rows of the image border)
  uint8_t mk_next = __builtin_amdgcn_raw_buffer_load_b8(r_mask, xo_c, mask_row(0) * W, 0);
  uint8_t mk_post = 0;                                                // mask byte of the row post() handles
  // post(k): everything of eigenvalue row k - 2 R that needs the OTHER threads' prefix sums (written before the last barrier): the
  // pending 3 x 3 test of the row before, the eigenvalue, maxima, the column maxima for the next test.  It runs one iteration late,
  // next to the Sobel / ring / scan work of row k + 1 (two independent dependency chains between two barriers).
  auto post = [&](int k) {
    if (do_nms) {
      if (s_flag[k & 1]) flush_now();
      test_pending((k - 1) & 1);                                      // row ye - 2 against the maxima written one row ago
      if (t == 0) s_flag[(k + 1) & 1] = (s_cnt > (unsigned int)(LCAP - 512)) ? 1u : 0u;
    }
    const uint4* buf = s_p[k & 1];
    const int ke = k - 2 * R, ye = y0 - 1 + ke;                       // eigenvalue row (-1 and H are halo rows of nothing)
    const uint8_t mk = mk_post;
    float e0;
    {
      // every thread forms a value (slot 256 holds zeros: no selects); the ones outside the eigenvalue columns drop it
      const uint4 A = buf[ia], Bv = buf[ib_z], T = buf[it_z];          // T: total of the wave the window starts in
      const int sa = (int)(A.x - Bv.x + T.x), sb = (int)(A.y - Bv.y + T.y), sc = (int)(A.z - Bv.z + T.z);
      const float a = ((float)sa * s2) * 0.5f, b = (float)sb * s2, c = ((float)sc * s2) * 0.5f;
      e0 = evalid ? (a + c) - st_sqrt_rn((a - c) * (a - c) + b * b) : 0.f;
    }
    const bool own_row = (ke >= 1) && (ke <= rows_out);               // ye is one of this workgroup's output rows
    const bool mine = outt && own_row;
    lmax = (mine && mk) ? fmaxf(lmax, e0) : lmax;
    if (mine) {
      const size_t o = (size_t)ye * W + xe;
      if (eig) eig[o] = e0;
      if (restore_mask && mk != 255) mask[o] = 255;
    }
    rmax = fmaxf(rmax, lmax);
    // publish: max of the column over rows ye - 2 .. ye, and the running maximum
    const float cm = fmaxf(e2, e0);
    if (do_nms) s_c[k & 1][t] = make_float2(fmaxf(cm, e1), rmax);
    // the centre row ye - 1 becomes the pending test (decided by the next post, when the neighbours' maxima are visible)
    pv = e1; pcm = cm; py = ye - 1;
    pok = outt && (ke >= 2) && (ke - 1 <= rows_out) && mk1 && (py >= 1) && (py < H - 1) && (xe >= 1) && (xe < W - 1);
    e2 = e1; e1 = e0; mk1 = (outt && own_row) ? mk : (uint8_t)0;
  };
  for (int kb = 0; kb < total; kb += D) {
#pragma unroll
    for (int u = 0; u < D; u++) {
      const int k = kb + u;
      if (k < total) {
        const uint32_t w0 = q[u % PF][0], w1 = q[u % PF][1], w2 = q[u % PF][2];
        // the mask byte of a row's output is requested one iteration ahead: waiting for it then leaves the prefetched
        // image rows in flight (loads return in order).  Unconditional, clamped address: a load under a divergent branch
        // would be waited for at the end of the branch.
        const uint8_t mk = mk_next;
        mk_next = __builtin_amdgcn_raw_buffer_load_b8(r_mask, xo_c, mask_row(max(k + 1 - 2 * R, 0)) * W, 0);
        // the slot just consumed receives the row PF ahead: the prefetch queue is indexed by the unrolled counter (no register
        // shuffling per row); D % PF rows of phase are taken out once per trip of the outer loop, below
        fetch(k + PF, q[u % PF]);
        // Sobel as byte dot products: bytes (left, centre, right) of the three rows
        const int dx = (int)(__builtin_amdgcn_udot4(w2, 0x00010000u, __builtin_amdgcn_udot4(w1, 0x00020000u, __builtin_amdgcn_udot4(w0, 0x00010000u, 0u, false), false), false) -
                             __builtin_amdgcn_udot4(w2, 0x00000001u, __builtin_amdgcn_udot4(w1, 0x00000002u, __builtin_amdgcn_udot4(w0, 0x00000001u, 0u, false), false), false));
        const int dy = (int)(__builtin_amdgcn_udot4(w2, 0x00010201u, 0u, false) - __builtin_amdgcn_udot4(w0, 0x00010201u, 0u, false));
        {
          // V += (products of the entering row) - (products of the leaving row), as three 2-term dot products of 16-bit pairs:
          //   (dx, dxo) . (dx, -dxo),  (dx, dxo) . (dy, -dyo),  (dy, dyo) . (dy, -dyo)
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          const uint32_t nw = __builtin_amdgcn_perm((uint32_t)dy, (uint32_t)dx, 0x05040100u), od = ring[u];
          ring[u] = nw;
          const s16x2 px = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(od, nw, 0x05040100u));      // (dx, dxo)
          const s16x2 py = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(od, nw, 0x07060302u));      // (dy, dyo)
          const s16x2 sgn = {1, -1};
          const s16x2 nx = px * sgn, ny = py * sgn;
          V0 = (unsigned)__builtin_amdgcn_sdot2(px, nx, (int)V0, false);
          V1 = (unsigned)__builtin_amdgcn_sdot2(px, ny, (int)V1, false);
          V2 = (unsigned)__builtin_amdgcn_sdot2(py, ny, (int)V2, false);
        }
        if (k >= 2 * R) {
          unsigned P0 = V0, P1 = V1, P2 = V2;
          st_scan64x3(P0, P1, P2);
          s_p[k & 1][t] = make_uint4(P0, P1, P2, 0u);
          if (k > 2 * R) post(k - 1);                                  // reads the OTHER buffer, filled before the last barrier
          mk_post = mk;
          __syncthreads();
        }
      }
    }
    if (D % PF != 0) {                                               // row kb + D sits in slot D % PF: rotate it to slot 0
      uint32_t tmp[PF][3];
#pragma unroll
      for (int i = 0; i < PF; i++) { tmp[i][0] = q[(i + D) % PF][0]; tmp[i][1] = q[(i + D) % PF][1]; tmp[i][2] = q[(i + D) % PF][2]; }
#pragma unroll
      for (int i = 0; i < PF; i++) { q[i][0] = tmp[i][0]; q[i][1] = tmp[i][1]; q[i][2] = tmp[i][2]; }
    }
  }
  post(total - 1);
  __syncthreads();
  if (do_nms) {
    if (s_flag[total & 1]) flush_now();
    test_pending((total - 1) & 1);
    flush_now();
  }
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
  if ((t & 63) == 0) s_m[t >> 6] = lmax;
  __syncthreads();
  if (t == 0) blockmax[(size_t)bseq * gridDim.x * gridDim.y + bxy] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

#define ST_NMS_ROWS 8
__global__ void __launch_bounds__(256) k_st_nms(const float* __restrict__ eig, const uint8_t* __restrict__ mask, int W,
                                                int H, double quality, const float* __restrict__ blockmax, int n_blockmax,
                                                unsigned long long* __restrict__ cand, uint32_t* __restrict__ scalars, size_t slab_seq,
                                                uint32_t* __restrict__ nraw) {
  __shared__ float s_m[4];
  __shared__ unsigned int s_cnt, s_base;
  __shared__ uint16_t s_list[256 * ST_NMS_ROWS];     // tile positions (row << 8 | column); EVERY pixel of the tile can qualify: the test is
                                                     // v >= neighbours, a flat plateau passes everywhere
  __shared__ float s_tile[ST_NMS_ROWS + 2][260];
  const int tid = threadIdx.x;
  {
    const int bseq = blockIdx.z;
    const size_t np = (size_t)W * H;
    eig += (size_t)bseq * np; mask += (size_t)bseq * np; blockmax += (size_t)bseq * n_blockmax;
    cand += (size_t)bseq * ST_CAND_STRIDE; scalars = vo_seq(scalars, slab_seq, bseq); nraw += bseq;
  }
  // ---- global masked maximum (minMaxLoc) from the per-block maxima ----
  float m = 0.f;
  for (int i = tid; i < n_blockmax; i += 256) m = fmaxf(m, blockmax[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) s_m[tid >> 6] = m;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const float maxv = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) scalars[0] = __float_as_uint(maxv);
  const float thr = (float)((double)maxv * quality);
  // ---- stage the (ST_NMS_ROWS + 2) x 258 eigenvalue tile through LDS: every load is issued up front and coalesced,
  // the 3x3 tests then read LDS (the direct form chained a dependent global load behind a branch per row) ----
  const int x0 = blockIdx.x * 256, y0 = blockIdx.y * ST_NMS_ROWS;      // tile origin = pixel (x0, y0); outputs at +1
  {
    float v[ST_NMS_ROWS + 2], vh[ST_NMS_ROWS + 2];
    const int xc = x0 + tid, xh = x0 + 256 + tid;                       // halo columns 256, 257 by threads 0, 1
#pragma unroll
    for (int r = 0; r < ST_NMS_ROWS + 2; r++) {
      const int yy = y0 + r;
      v[r] = (yy < H && xc < W) ? eig[(size_t)yy * W + xc] : 0.f;
      vh[r] = (tid < 2 && yy < H && xh < W) ? eig[(size_t)yy * W + xh] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < ST_NMS_ROWS + 2; r++) {
      s_tile[r][tid] = v[r];
      if (tid < 2) s_tile[r][256 + tid] = vh[r];
    }
  }
  const int x = x0 + tid + 1;
  uint8_t mk[ST_NMS_ROWS];
#pragma unroll
  for (int ry = 0; ry < ST_NMS_ROWS; ry++) {
    const int y = y0 + ry + 1;
    mk[ry] = (x < W - 1 && y < H - 1) ? mask[(size_t)y * W + x] : (uint8_t)0;
  }
  __syncthreads();
  if (x < W - 1) {
#pragma unroll
    for (int ry = 0; ry < ST_NMS_ROWS; ry++) {
      const int y = y0 + ry + 1;
      if (y >= H - 1) break;
      const float v = s_tile[ry + 1][tid + 1];
      if (!(v > thr) || v == 0.f || !mk[ry]) continue;
      const float* e0 = &s_tile[ry][tid];
      const float* e1 = &s_tile[ry + 1][tid];
      const float* e2 = &s_tile[ry + 2][tid];
      const bool ismax = e0[0] <= v && e0[1] <= v && e0[2] <= v && e1[0] <= v && e1[2] <= v &&
                         e2[0] <= v && e2[1] <= v && e2[2] <= v;
      if (!ismax) continue;
      const unsigned int pos = atomicAdd(&s_cnt, 1u);        // LDS
      s_list[pos] = (uint16_t)((ry << 8) | tid);
    }
  }
  __syncthreads();
  const unsigned int cnt = s_cnt;
  if (tid == 0 && cnt) s_base = atomicAdd(nraw, cnt);   // one global atomic per workgroup
  __syncthreads();
  for (unsigned int i = tid; i < cnt; i += 256) {
    const unsigned int pos = s_base + i;
    const int ry = s_list[i] >> 8, tx = s_list[i] & 255;
    const float v = s_tile[ry + 1][tx + 1];
    const size_t o = (size_t)(y0 + ry + 1) * W + (x0 + tx + 1);
    if (pos < ST_GLOBAL_CAP) cand[pos] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(uint32_t)o;
  }
}

// wave-uniform values that come out of LDS / global memory arrive in vector registers; pinning them into scalar registers keeps
// the loop-carried state of k_st_select (counts, bounds, the 64-bit rank limits) out of the 128-VGPR budget of a 1024-thread workgroup
__device__ __forceinline__ uint32_t st_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long st_uniform64(unsigned long long v) {
  return ((unsigned long long)st_uniform((uint32_t)(v >> 32)) << 32) | st_uniform((uint32_t)v);
}

// LDS map of k_st_select for `cap` entries: exchange buffer of the sort: keys u64 [n2 <= cap] at 0.
// selection phase (keys dead): xy u32 [cap] at 0 | cell heads u32 [8192] | next u16 [cap] | state u8 [cap]   (144 KB at cap = 16 384, 60 KB at 4 096)
#define ST_MAX_CELLS 8192
static inline size_t st_sel_lds(int cap) { const size_t a = 8 * (size_t)cap, b = 7 * (size_t)cap + 4 * ST_MAX_CELLS; return a > b ? a : b; }
#define ST_SEL_LDS st_sel_lds(ST_CAND_CAP)
#define ST_CAP_CLOSED_LOOP 4096
#define ST_KP_MAX (ST_CAND_CAP / 1024)

// value of lane (l ^ J) for J = 1 ... 32 without the LDS pipe's address path: DPP quad permutes / row rotation, one swizzle, the two
// permlane swaps
template <int J>
__device__ __forceinline__ uint32_t st_xor_lane(uint32_t x, int lane) {
  if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);        // quad_perm [1, 0, 3, 2]
  if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);        // quad_perm [2, 3, 0, 1]
  if (J == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x101F);                         // bit mode: and 0x1f, or 0, xor 4
  if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);       // row_ror:8
  if (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); return (lane & 16) ? r[0] : r[1]; }
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return (lane & 32) ? r[0] : r[1];
}

template <int KP, int J>
__device__ __forceinline__ void st_wave_substep(unsigned long long (&kr)[KP], int ibase, int lane, int k) {
#pragma unroll
  for (int q = 0; q < KP; q++) {
    const int i = ibase + q * 64;
    const unsigned long long mine = kr[q];
    const unsigned long long other = ((unsigned long long)st_xor_lane<J>((uint32_t)(mine >> 32), lane) << 32) | st_xor_lane<J>((uint32_t)mine, lane);
    const bool take_max = (((i & k) == 0) == ((lane & J) == 0));
    kr[q] = take_max ? (mine > other ? mine : other) : (mine < other ? mine : other);
  }
}

// Bitonic sort (descending) of n2 = 1024 * KP keys held in REGISTERS: lane (wave w, lane l) owns keys
// i = w * 64 KP + q * 64 + l, q < KP.  Compare-exchange distances j < 64 are cross-lane shuffles, 64 <= j < 64 KP
// are pure register swaps, only j >= 64 KP (between waves) goes through LDS -- LDS latency (~100 cycles per
// dependent access) is what bounds an all-LDS bitonic sort of this size.  What bounds THIS one is vector issue: 2 048 keys x 66 steps x
// ~12 instructions on 64-bit keys = 38 k cycles (42 k with ds_bpermute exchanges).  Packing a small sort onto 4 waves x 8 keys (3 cross-wave
// steps instead of 10) was measured and is slower, 60 k: one wave per SIMD cannot hide the exchange latencies.
// n2: the network's size for this call, a power of two in [64, 1024 KP] that covers the real keys (the slots behind them hold 0 and sort among
// themselves): a 256-key chunk of the closed loop takes 36 compare-exchange steps, 3 of them across waves, instead of 55 / 10
template <int KP>
__device__ __forceinline__ void st_sort_regs(unsigned long long (&kr)[KP], unsigned long long* keys, int tid, int n2) {
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int SEG = KP * 64;
  const int ibase = wave * SEG + lane;
#pragma unroll 1
  for (int k = 2; k <= n2; k <<= 1) {
#pragma unroll 1
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= SEG) {
        // partner lives in another wave: exchange through LDS
#pragma unroll
        for (int q = 0; q < KP; q++) keys[ibase + q * 64] = kr[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KP; q++) {
          const int i = ibase + q * 64;
          const unsigned long long other = keys[i ^ j];
          const bool take_max = (((i & k) == 0) == ((i & j) == 0));
          const unsigned long long mine = kr[q];
          kr[q] = take_max ? (mine > other ? mine : other) : (mine < other ? mine : other);
        }
        __syncthreads();
      } else if (j >= 64) {
        const int jq = j >> 6;
#pragma unroll
        for (int q = 0; q < KP; q++) {
          if ((q & jq) == 0 && (q | jq) < KP) {
            const int i = ibase + q * 64;
            const bool desc = ((i & k) == 0);
            const unsigned long long a = kr[q], b = kr[q | jq];
            const bool sw = desc ? (a < b) : (a > b);
            kr[q] = sw ? b : a; kr[q | jq] = sw ? a : b;
          }
        }
      } else {
        // partner in the same wave: steps j, j / 2, ... 1 in one go, the lane-xor exchanges as VALU / swizzle instructions with compile-time
        // patterns (a run-time __shfl_xor of a 64-bit key is two ds_bpermute + their address arithmetic per key and step)
        if (j >= 32) st_wave_substep<KP, 32>(kr, ibase, lane, k);
        if (j >= 16) st_wave_substep<KP, 16>(kr, ibase, lane, k);
        if (j >= 8) st_wave_substep<KP, 8>(kr, ibase, lane, k);
        if (j >= 4) st_wave_substep<KP, 4>(kr, ibase, lane, k);
        if (j >= 2) st_wave_substep<KP, 2>(kr, ibase, lane, k);
        st_wave_substep<KP, 1>(kr, ibase, lane, k);
        break;
      }
    }
  }
}

// sorts the keys of cand[0 .. n_load) that exceed `floor_key` (the others count as absent: key 0 sorts last); n of them exist
template <int KP>
__device__ __forceinline__ void st_sort_dispatch(const unsigned long long* __restrict__ cand, int n_load, unsigned long long floor_key, int n,
                                                 unsigned long long* keys, uint32_t* xy, int W, int tid) {
  unsigned long long kr[KP];
  const int ibase = (tid >> 6) * (KP * 64) + (tid & 63);
#pragma unroll
  for (int q = 0; q < KP; q++) {
    const int i = ibase + q * 64;
    const unsigned long long k = (i < n_load) ? cand[i] : 0ull;
    kr[q] = (k > floor_key) ? k : 0ull;
  }
  int n2 = 64;
  while (n2 < n_load) n2 <<= 1;                    // (uniform; n_load <= 1024 KP)
  st_sort_regs<KP>(kr, keys, tid, n2);
  __syncthreads();   // the exchange buffer aliases xy
  // keys -> packed (x, y) in rank order
#pragma unroll
  for (int q = 0; q < KP; q++) {
    const int i = ibase + q * 64;
    if (i < n) {
      const int idx = (int)(uint32_t)kr[q];
      const int y = idx / W, x = idx - y * W;
      xy[i] = (uint32_t)x | ((uint32_t)y << 16);
    }
  }
}

template <int cap>       // entries the LDS structures hold: a template parameter so that the LDS offsets stay instruction immediates
__global__ void __launch_bounds__(1024) k_st_select(unsigned long long* __restrict__ cand,
                                                    uint32_t* __restrict__ scalars, int W, int H, int cell, int gw, int gh,
                                                    double md2, int use_dist, int max_corners, float* __restrict__ out,
                                                    size_t slab_seq, unsigned long long* __restrict__ dbg,
                                                    const float* __restrict__ blockmax, int n_blockmax, double quality,
                                                    uint32_t* __restrict__ nraw, const int32_t* __restrict__ limit_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cand += (size_t)blockIdx.x * ST_CAND_STRIDE;          // one workgroup per sequence
  scalars = vo_seq(scalars, slab_seq, blockIdx.x); out = vo_seq(out, slab_seq, blockIdx.x);
  if (blockIdx.x != 0) dbg = nullptr;
  __shared__ int s_flags[3];
  __shared__ int s_scan[1024];
  // cap: entries the LDS structures of this launch hold (st_sel_lds(cap) bytes): ST_CAND_CAP = 16 384 (144 KB: a CU to itself) for a
  // caller that may want every corner, 4 096 (60 KB) for the closed loop, whose device-side corner limit keeps the chunks short -- with 144 KB
  // the workgroup had to wait until a whole CU's LDS was free of the bundle adjustment's workgroups running beside it
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* xy = reinterpret_cast<uint32_t*>(smem);
  uint32_t* heads = reinterpret_cast<uint32_t*>(smem + 4 * (size_t)cap);
  uint16_t* nxt = reinterpret_cast<uint16_t*>(smem + 4 * (size_t)cap + 4 * ST_MAX_CELLS);
  uint8_t* state = smem + 6 * (size_t)cap + 4 * ST_MAX_CELLS;   // plain LDS bytes (a volatile pointer here degrades to FLAT sc0 sc1 accesses)
  const int tid = threadIdx.x;
  VO_STAMP(dbg, 0);
  // ---- global masked maximum (minMaxLoc) from the per-workgroup maxima, the quality threshold, and the candidates that pass it:
  //      the NMS stage appended against a running lower bound of the maximum (a superset, in arbitrary order) ----
  __shared__ float s_mx[16];
  __shared__ uint32_t s_nvalid;
  blockmax += (size_t)blockIdx.x * n_blockmax; nraw += blockIdx.x;
  {
    float m = 0.f;
    for (int i = tid; i < n_blockmax; i += 1024) m = fmaxf(m, blockmax[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) s_mx[tid >> 6] = m;
    if (tid == 0) s_nvalid = 0;
  }
  __syncthreads();
  float maxv = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) maxv = fmaxf(maxv, s_mx[i]);
  const float thr = __uint_as_float(st_uniform(__float_as_uint((float)((double)maxv * quality))));
  const uint32_t n_raw = st_uniform(*nraw);
  if (n_raw > ST_GLOBAL_CAP) {                            // host reports VO_E_CAPACITY
    __syncthreads();
    if (tid == 0) { scalars[0] = __float_as_uint(maxv); scalars[1] = n_raw; scalars[2] = 0xFFFFFFFFu; *nraw = 0; }
    return;
  }
  // candidates that pass the true threshold: positive floats order like their bit patterns, so "value > thr" is one 64-bit
  // compare of the key against (bits(thr) << 32 | ~0); the list is neither copied nor compacted, every later stage applies it
  const unsigned long long thr_key = ((unsigned long long)__float_as_uint(thr) << 32) | 0xFFFFFFFFull;
  {
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < n_raw; i += 1024) cnt += (cand[i] > thr_key) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0 && cnt) atomicAdd(&s_nvalid, cnt);
    __syncthreads();
  }
  const uint32_t ncand = st_uniform(s_nvalid);
  if (tid == 0) { scalars[0] = __float_as_uint(maxv); scalars[1] = ncand; *nraw = 0; }   // the raw counter is re-armed for the next launch
  // ---- OpenCV scans the candidates in rank order (value desc, index desc) and accepts one iff no accepted corner lies closer
  //      than min_distance, until max_corners are out.  The LDS structures hold ST_CAND_CAP entries, so the list is consumed in
  //      rank-ordered CHUNKS: the corners accepted so far (they outrank everything that follows) + the strongest remaining keys.
  //      One chunk is the whole list in practice; more are needed only when > ST_CAND_CAP maxima pass the threshold AND the
  //      strongest of them do not fill max_corners (flat or noisy images with a large min_distance). ----
  uint32_t* hist = reinterpret_cast<uint32_t*>(s_scan);
  __shared__ unsigned long long s_prefix;
  __shared__ uint32_t s_need, s_fill, s_take;
  int limit = (max_corners > 0) ? min(max_corners, ST_OUT_CAP) : ST_OUT_CAP;
  // limit_dev (closed loop): the caller can use at most that many corners of this sequence (the free slots of its table + 1, typically a few
  // dozen).  The scan is in rank order, so the first `limit` accepted corners depend only on the strongest candidates: the list is consumed
  // in SHORT rank-ordered chunks (the radix select below) instead of sorting all of it -- the same corners, exactly.
  int kcap = cap;
  if (limit_dev) {
    const int ld = (int)st_uniform((uint32_t)max(limit_dev[blockIdx.x], 0));
    limit = min(limit, ld);
    kcap = min(cap, max(256, 8 * limit));
  }
  unsigned long long* const src = cand;                   // raw list (entries <= thr_key do not count)
  unsigned long long* const top_buf = cand + ST_GLOBAL_CAP;   // chunk staging (behind the raw list)
  unsigned long long upper = ~0ull;                       // keys >= upper have been consumed by earlier chunks
  uint32_t remaining = ncand;
  int n_acc = 0;                                          // corners accepted so far: out[0 .. n_acc) in rank order
  uint32_t rounds_total = 0;
#pragma unroll 1
  for (int pass = 0;; pass++) {
    const int K = min(cap - n_acc, kcap);
    const unsigned long long* chunk = src;
    int n_new = (int)remaining, n_load = (int)n_raw;
    unsigned long long next_upper = 0;
    if (remaining > (uint32_t)K) {
      // the K strongest keys below `upper`: keys are unique (value bits | pixel index), so the K-th largest is found exactly by
      // an 8-pass radix select (256-bin LDS histogram of the next byte among the keys that share the prefix)
      // It stops as soon as the WHOLE bin of the K-th key still fits the sort's size class (K <= taken <= k_room): the chunk is then the keys
      // down to that bin's lower bound -- a rank-ordered prefix of the list like any other, a few more keys than asked for -- after two or three
      // passes instead of eight (eigenvalues between 3 % and 100 % of the maximum spread over ~640 bins of the top 16 bits).
      int k_room = 512;                                   // (the sort network's size classes: 2 K at least, so that a whole bin has room)
      while (k_room < 2 * K) k_room <<= 1;
      k_room = min(k_room, cap - n_acc);
      // Every valid key lies in (thr_key, max_key]: the bits above the highest bit in which those two differ are common to all of them and
      // carry no information (with the byte-aligned digits of the first form the whole first pass -- sign and seven exponent bits -- put every
      // key of the list into one or two bins, 1 600 LDS atomics on the same address, and the second pass did the work).  The digits start at
      // that bit: the first histogram spreads the keys over the whole value range [quality x max, max].
      const unsigned long long max_key = ((unsigned long long)__float_as_uint(maxv) << 32) | 0xFFFFFFFFull;
      const int top = 63 - __builtin_clzll((thr_key ^ max_key) | 0xFFull);      // >= 7; (a list with one distinct value: the low byte)
      unsigned long long prefix = (top >= 63) ? 0ull : (max_key >> (top + 1)) << (top + 1);
      uint32_t need = (uint32_t)K, taken = (uint32_t)K;
      for (int hi = top; hi >= 0; hi -= 8) {                // digit = bits hi ... max(hi - 7, 0)
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int sh = max(hi - 7, 0), nbits = hi - sh + 1;
        const uint32_t dmask = (1u << nbits) - 1u;
        for (uint32_t i = tid; i < n_raw; i += 1024) {
          const unsigned long long k = src[i];
          if (k > thr_key && k < upper && (hi >= 63 || (k >> (hi + 1)) == (prefix >> (hi + 1)))) atomicAdd(&hist[(uint32_t)(k >> sh) & dmask], 1u);
        }
        __syncthreads();
        // the bin that holds the need-th largest key, counting from bin 255 down (bin 0 takes what is left): one wave, four bins per lane,
        // a suffix sum over the lanes, then at most three steps inside the lane that holds it (one thread walking 255 dependent LDS reads
        // cost 25 k cycles per pass: fine while this path was rare, not for the closed loop's short chunks)
        if (tid < 64) {
          const int l = tid;
          const uint32_t h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
          const uint32_t s4 = hist[4 * l] + h1 + h2 + h3;
          uint32_t x = s4;                                   // -> bins 4 l ... 255
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_down((int)x, o); if (l + o < 64) x += y; }
          const uint32_t above = x - s4;
          const bool hit = above < need && need <= x;
          const bool none = __ballot(hit) == 0ull;           // fewer keys than asked for: everything down to bin 0
          if (hit || (none && l == 0)) {
            uint32_t nd = need - above;
            int d = 4 * l + 3;
            if (nd > h3) { nd -= h3; d--; if (nd > h2) { nd -= h2; d--; if (nd > h1) { nd -= h1; d--; } } }
            s_prefix = prefix | ((unsigned long long)d << sh);
            s_need = nd;
            s_fill = 0;
            // keys above this bin (K - need from the earlier passes + the bins above it in this one) + the whole bin
            s_take = ((uint32_t)K - need) + (need - nd) + hist[d];
          }
        }
        __syncthreads();
        prefix = st_uniform64(s_prefix); need = st_uniform(s_need);
        const uint32_t whole = st_uniform(s_take);
        if (whole <= (uint32_t)k_room) { taken = whole; break; }     // (uniform)
      }
      for (uint32_t i0 = 0; i0 < n_raw; i0 += 1024) {        // (uniform trip count: the ballot below needs every lane)
        const uint32_t i = i0 + tid;
        const unsigned long long k = (i < n_raw) ? src[i] : 0ull;
        const bool in = k >= prefix && k > thr_key && k < upper;      // (a whole bin may reach below the threshold)
        const unsigned long long bal = __ballot(in);
        if (bal) {                                                    // one LDS atomic per wave instead of one per key on the same counter
          uint32_t base = 0;
          if ((tid & 63) == 0) base = atomicAdd(&s_fill, (uint32_t)__popcll(bal));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          const uint32_t pos = base + (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
          if (in && pos < (uint32_t)cap) top_buf[pos] = k;
        }
      }
      __syncthreads();
      chunk = top_buf; n_new = (int)taken; n_load = (int)taken; next_upper = prefix;
    } else if (pass > 0) {
      if (tid == 0) s_fill = 0;
      __syncthreads();
      for (uint32_t i = tid; i < n_raw; i += 1024) {
        const unsigned long long k = src[i];
        if (k > thr_key && k < upper) top_buf[atomicAdd(&s_fill, 1u)] = k;
      }
      __syncthreads();
      chunk = top_buf; n_load = n_new;
    } else if (n_raw > (uint32_t)cap) {
      // first and only chunk, but the raw list is longer than the sort holds: compact the valid keys (<= ST_CAND_CAP of them)
      if (tid == 0) s_fill = 0;
      __syncthreads();
      for (uint32_t i = tid; i < n_raw; i += 1024) {
        const unsigned long long k = src[i];
        if (k > thr_key) top_buf[atomicAdd(&s_fill, 1u)] = k;
      }
      __syncthreads();
      chunk = top_buf; n_load = n_new;
    }
    const int n = n_acc + n_new;
    // ---- sort the chunk by (value desc, index desc) = OpenCV's greaterThanPtr order; result: xy[n_acc ..] in rank order ----
    if (n_load <= 1024) st_sort_dispatch<1>(chunk, n_load, thr_key, n_new, keys, xy + n_acc, W, tid);
    else if (n_load <= 2048) st_sort_dispatch<2>(chunk, n_load, thr_key, n_new, keys, xy + n_acc, W, tid);
    else if (n_load <= 4096) st_sort_dispatch<4>(chunk, n_load, thr_key, n_new, keys, xy + n_acc, W, tid);
    else if (n_load <= 8192) st_sort_dispatch<8>(chunk, n_load, thr_key, n_new, keys, xy + n_acc, W, tid);
    else st_sort_dispatch<16>(chunk, n_load, thr_key, n_new, keys, xy + n_acc, W, tid);
    __syncthreads();
    for (int i = tid; i < n_acc; i += 1024) xy[i] = (uint32_t)out[2 * i] | ((uint32_t)out[2 * i + 1] << 16);   // earlier chunks' corners
    for (int i = tid; i < n; i += 1024) state[i] = (i < n_acc || !use_dist) ? 1 : 0;
    for (int i = tid; i < gw * gh; i += 1024) heads[i] = 0xFFFFu;
    if (tid < 3) s_flags[tid] = 0;
    __syncthreads();
    if (pass == 0) VO_STAMP(dbg, 1);   // sort done
    // The conflict lists live in registers, one 64-bit word per own candidate: the section is specialised by the number of candidates per
    // thread the chunk really has (1, 2, 4, 8 or 16 -- a full-size frame has 1 500 ... 4 000 candidates), otherwise the 16-entry arrays of
    // the largest case set the register pressure of every launch (90 spilled registers in a 1024-thread workgroup)
    auto select_rounds = [&](auto kp_tag) {
      constexpr int KPC = decltype(kp_tag)::value;
      // ---- grid of linked lists (acceleration structure only: any cell size >= min_distance gives the same result) ----
      for (int i = tid; i < n; i += 1024) {
        const uint32_t p = xy[i];
        const int x = p & 0xFFFF, y = p >> 16;
        nxt[i] = (uint16_t)atomicExch(&heads[(y / cell) * gw + (x / cell)], (uint32_t)i);
      }
      __syncthreads();
      if (pass == 0) VO_STAMP(dbg, 2);   // grid built
      // ---- conflict lists: for each own candidate the (<= 4) higher-ranked candidates closer than min_distance,
      //      found by ONE walk over the 3x3 cells (9 independent head reads, then the short chains) and kept in
      //      registers, so that the selection rounds below touch one LDS byte per conflict ----
      unsigned long long nb[KPC];
      uint32_t over = 0, undec = 0;   // bit q: list overflowed / still undecided
#pragma unroll
      for (int q = 0; q < KPC; q++) {
        nb[q] = ~0ull;
        const int i = tid + q * 1024;
        if (i >= n_acc && i < n) {
          undec |= 1u << q;
          const uint32_t p = xy[i];
          const int x = p & 0xFFFF, y = p >> 16;
          const int xc = x / cell, yc = y / cell;
          uint32_t hd[9];
#pragma unroll
          for (int c9 = 0; c9 < 9; c9++) {
            const int xx = xc + (c9 % 3) - 1, yy = yc + (c9 / 3) - 1;
            hd[c9] = (xx >= 0 && xx < gw && yy >= 0 && yy < gh) ? heads[yy * gw + xx] : 0xFFFFu;
          }
          int cnt = 0;
          unsigned long long list = ~0ull;
#pragma unroll
          for (int c9 = 0; c9 < 9; c9++)
            for (uint32_t qn = hd[c9]; qn != 0xFFFFu; qn = nxt[qn]) {
              if ((int)qn >= i) continue;   // only higher-ranked candidates matter
              const uint32_t pq = xy[qn];
              const int ddx = x - (int)(pq & 0xFFFF), ddy = y - (int)(pq >> 16);
              if ((double)(ddx * ddx + ddy * ddy) < md2) {
                if (cnt < 4) list = (list << 16) | (unsigned long long)qn;
                cnt++;
              }
            }
          nb[q] = list;
          if (cnt > 4) over |= 1u << q;
        }
      }
      // ---- greedy min-distance selection as monotone parallel rounds: a candidate is accepted iff every
      //      higher-ranked candidate closer than min_distance is rejected (== OpenCV's sequential scan) ----
      for (int round = 0; round <= n; round++) {
        if (tid == 0) s_flags[(round + 1) % 3] = 0;   // re-arm the flag last read two rounds ago
#pragma unroll
        for (int q = 0; q < KPC; q++) {
          if (!((undec >> q) & 1)) continue;
          const int i = tid + q * 1024;
          bool any_acc = false, any_und = false;
          if (!((over >> q) & 1)) {
            const unsigned long long list = nb[q];
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const uint32_t qn = (uint32_t)(list >> (16 * t)) & 0xFFFFu;
              if (qn != 0xFFFFu) {
                const uint8_t sq = state[qn];
                any_acc |= (sq == 1); any_und |= (sq == 0);
              }
            }
          } else {
            // rare: more than 4 conflicts -> walk the grid again
            const uint32_t p = xy[i];
            const int x = p & 0xFFFF, y = p >> 16;
            const int xc = x / cell, yc = y / cell;
            const int x1 = max(xc - 1, 0), y1 = max(yc - 1, 0), x2 = min(xc + 1, gw - 1), y2 = min(yc + 1, gh - 1);
            for (int yy = y1; yy <= y2; yy++)
              for (int xx = x1; xx <= x2; xx++)
                for (uint32_t qn = heads[yy * gw + xx]; qn != 0xFFFFu; qn = nxt[qn]) {
                  if ((int)qn >= i) continue;
                  const uint32_t pq = xy[qn];
                  const int ddx = x - (int)(pq & 0xFFFF), ddy = y - (int)(pq >> 16);
                  if ((double)(ddx * ddx + ddy * ddy) < md2) {
                    const uint8_t sq = state[qn];
                    any_acc |= (sq == 1); any_und |= (sq == 0);
                  }
                }
          }
          if (any_acc) { state[i] = 2; undec &= ~(1u << q); }
          else if (!any_und) { state[i] = 1; undec &= ~(1u << q); }
        }
        if (undec) s_flags[round % 3] = 1;
        __syncthreads();
        if (round == 0 && pass == 0) VO_STAMP(dbg, 5);
        if (!s_flags[round % 3]) { rounds_total += (uint32_t)round + 1; break; }
      }
    };
    if (use_dist) {
      if (n <= 1024) select_rounds(std::integral_constant<int, 1>{});
      else if (n <= 2048) select_rounds(std::integral_constant<int, 2>{});
      else if (n <= 4096) select_rounds(std::integral_constant<int, 4>{});
      else if (n <= 8192) select_rounds(std::integral_constant<int, 8>{});
      else select_rounds(std::integral_constant<int, ST_KP_MAX>{});
    }
    if (pass == 0) VO_STAMP(dbg, 3);   // rounds done
    // ---- ordered compaction of the chunk's accepted candidates (rank order) behind the earlier ones, up to max_corners ----
    const int per = (n_new + 1023) / 1024;
    const int b0 = n_acc + min(tid * per, n_new), b1 = min(b0 + per, n);
    int cnt = 0;
    for (int i = b0; i < b1; i++) cnt += (state[i] == 1);
    int incl = cnt;                  // inclusive scan inside the wave, the 16 wave totals through LDS: two barriers instead of twenty
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if ((tid & 63) >= o) incl += y; }
    __syncthreads();                 // hist (the radix select) aliases s_scan
    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    int woff = 0, wtot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const int v = s_scan[w]; if (w < (tid >> 6)) woff += v; wtot += v; }
    int pos = n_acc + woff + incl - cnt;
    const int total = n_acc + (int)st_uniform((uint32_t)wtot);
    for (int i = b0; i < b1; i++)
      if (state[i] == 1) {
        if (pos < limit) {
          const uint32_t p = xy[i];
          out[2 * pos] = (float)(p & 0xFFFF); out[2 * pos + 1] = (float)(p >> 16);
        }
        pos++;
      }
    n_acc = min(total, limit);
    remaining -= (uint32_t)n_new;
    __syncthreads();                 // out[] and s_scan are re-read / re-used by the next chunk
    if (n_acc >= limit || remaining == 0) break;
    upper = next_upper;
  }
  if (tid == 0) { scalars[2] = (uint32_t)n_acc; scalars[3] = rounds_total; if (dbg) dbg[6] = (unsigned long long)rounds_total; }
  VO_STAMP(dbg, 4);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void vo_st_destroy(vo_ctx* c) {
  if (!c->st) return;
  vo_st_ws* s = c->st;
  void* bufs[] = {s->d_mask, s->d_user_mask, s->d_h, s->d_eig, s->d_cand, s->d_nraw, s->d_blockmax, s->d_pts};   // scalars / out live in the ctx slab
  for (void* b : bufs) if (b) (void)hipFree(b);
  delete s;
  c->st = nullptr;
}

static int32_t st_init(vo_ctx* c) {
  if (c->st) return VO_OK;
  vo_st_ws* s = new vo_st_ws();
  c->st = s;
  const size_t np = (size_t)c->width * c->height, B = (size_t)c->batch;
  VO_HIP(c, hipMalloc((void**)&s->d_mask, np * B));
  VO_HIP(c, hipMalloc((void**)&s->d_user_mask, np * B));
  VO_HIP(c, hipMalloc((void**)&s->d_eig, np * sizeof(float) * B));
  s->d_scalars = vo_slab<uint32_t>(c, c->off_st_scalars);
  VO_HIP(c, hipMalloc((void**)&s->d_cand, sizeof(unsigned long long) * ST_CAND_STRIDE * B));
  VO_HIP(c, hipMalloc((void**)&s->d_nraw, sizeof(uint32_t) * B));
  VO_HIP(c, hipMemsetAsync(s->d_nraw, 0, sizeof(uint32_t) * B, c->stream));
  s->n_blockmax = vo_div_up(c->width, 256 - 32) * c->height;                     // upper bound over both eigenvalue paths (1-row bands)
  VO_HIP(c, hipMalloc((void**)&s->d_blockmax, sizeof(float) * (size_t)s->n_blockmax * B));
  s->d_out = vo_slab<float>(c, c->off_st_out);
  VO_HIP(c, hipMalloc((void**)&s->d_pts, sizeof(float) * 2 * (size_t)c->max_pts * B));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_st_select<ST_CAP_CLOSED_LOOP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)st_sel_lds(ST_CAP_CLOSED_LOOP)));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_st_select<ST_CAND_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ST_SEL_LDS));
  return VO_OK;
}

// filled-circle row table: the integer midpoint loop of imgproc/drawing.cpp Circle()
static void circle_rows(int radius, disc_rows* rows) {
  for (int i = 0; i <= ST_MAX_RADIUS; i++) rows->hw[i] = -1;
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    if (dx > rows->hw[dy]) rows->hw[dy] = dx;
    if (dy > rows->hw[dx]) rows->hw[dx] = dy;
    dy++;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
}

bool vo_st_ready(const vo_ctx* c) { return c->st != nullptr; }
int vo_st_last_max_corners(const vo_ctx* c) { return c->st ? c->st->last_max_corners : 0; }
// host-side flags a launch sets at ENQUEUE time (also under stream capture): a capture that fails must put them back, or the next
// resident launch skips k_st_mask_init on a mask that still holds the previous frame's discs
int vo_st_flags_save(const vo_ctx* c) { return c->st ? (c->st->mask_clean ? 1 : 0) | (c->st->eig_valid ? 2 : 0) : -1; }
void vo_st_flags_restore(vo_ctx* c, int saved) { if (c->st && saved >= 0) { c->st->mask_clean = (saved & 1) != 0; c->st->eig_valid = (saved & 2) != 0; } }
int vo_st_launch_state(const vo_ctx* c) { return c->st ? (c->st->mask_clean ? 1 : 0) | (c->tune.st_keep_eig ? 2 : 0) : 0; }   // decides which kernels a resident launch enqueues
// all allocations a launch with these parameters needs (called outside any graph capture)
int32_t vo_st_prepare(vo_ctx* c, const vo_st_params* prm) {
  int32_t r = st_init(c);
  if (r != VO_OK) return r;
  vo_st_ws* s = c->st;
  const bool fused = !c->tune.st_two_kernels && prm && prm->block_size == 31 && c->height > 31 && c->width > 31;
  if (!fused && !s->d_h) VO_HIP(c, hipMalloc((void**)&s->d_h, (size_t)c->width * c->height * 3 * sizeof(int32_t) * c->batch));
  return VO_OK;
}

extern "C" int32_t vo_st_default_params(vo_st_params* p) {
  if (!p) return VO_E_INVALID;
  p->max_corners = 1000; p->block_size = 31; p->quality_level = 0.03; p->min_distance = 7.0;
  p->use_harris = 0; p->_pad = 0; p->harris_k = 0.04;
  return VO_OK;
}

// pts: sequence-0 pointer of the exclusion-disc centres, pts_seq: byte stride between sequences
// keep: store the eigenvalue map and leave the exclusion mask in place (vo_shi_tomasi_read); otherwise the fused kernel writes
// no map and hands the mask back clean, which saves the k_st_mask_init launch of the next call
static int32_t st_launch(vo_ctx* c, const float* d_pts, size_t pts_seq, int n_cur, int mask_radius, const uint8_t* d_user_mask,
                         const vo_st_params* prm, const int32_t* counts, bool keep, const int32_t* limit_dev = nullptr) {
  VO_CHECK(c, c->n_pushed >= 1, VO_E_STATE, "no frame pushed");
  VO_CHECK(c, prm->block_size >= 1 && prm->block_size <= 31 && (prm->block_size & 1), VO_E_INVALID,
           "block_size must be odd, <= 31");
  VO_CHECK(c, mask_radius >= 0 && mask_radius <= ST_MAX_RADIUS, VO_E_INVALID, "mask_radius must be 0..31");
  VO_CHECK(c, prm->max_corners <= ST_OUT_CAP, VO_E_CAPACITY, "max_corners exceeds 4096");
  vo_st_ws* s = c->st;
  vo_prof_scope prof(c, VO_PROF_ST);
  const int W = c->width, H = c->height, r = prm->block_size / 2, B = c->batch;
  const vo_frame& F = c->fr[c->cur];
  const double scale_d = 1.0 / ((double)(1 << 2) * prm->block_size * 255.0);
  const float sf = (float)scale_d;
  const float s2 = sf * sf;
  const bool fused = !c->tune.st_two_kernels && r == 15 && H > 31 && W > 31 && !prm->use_harris;    // (single border reflection per row inside the kernel; the Harris
                                                                                     //  response -- an option the reference never enables -- takes the two-kernel form)
  int n_blockmax;
  if (fused) {
    if (keep || d_user_mask || !s->mask_clean)
      hipLaunchKernelGGL(k_st_mask_init, dim3(vo_div_up((int)(((size_t)W * H + 15) / 16), 256), B), dim3(256), 0, c->stream, s->d_mask, d_user_mask,
                         (size_t)W * H, s->d_scalars, c->slab_seq);
  } else {
    if (!s->d_h) VO_HIP(c, hipMalloc((void**)&s->d_h, (size_t)W * H * 3 * sizeof(int32_t) * B));
    hipLaunchKernelGGL(k_st_sobel_hsum, dim3(vo_div_up(W, ST_HS_COLS), H, B), dim3(256), 0, c->stream, F.img[0], c->lvl_px[0],
                       c->lv[0].pitch, W, H, r, s->d_h, s->d_mask, d_user_mask, s->d_scalars, c->slab_seq);
  }
  if (n_cur > 0) {
    disc_rows rows;
    circle_rows(mask_radius, &rows);
    const int total = n_cur * (2 * mask_radius + 1);
    hipLaunchKernelGGL(k_st_discs, dim3(vo_div_up(total, 256), B), dim3(256), 0, c->stream, d_pts, pts_seq, n_cur, mask_radius,
                       rows, s->d_mask, W, H, counts);
  }
  if (fused) {
    // rows per band: few bands keep the 30-row start-up small; with few sequences in flight more, shorter bands fill the GPU
    const int gx = vo_div_up(W, 256 - 32);
    // (5 workgroups fit a CU: 95 registers, 28.7 KB of LDS.)  Bands of at least 12 rows; among the band counts the one with the least
    // rounds x rows per workgroup (rb + the 30-row start-up), where rounds = workgroups / 1 280 slots, rounded up: a launch of 1.2 rounds
    // takes two.  KITTI frames, 6 column blocks: a batch of 32 -> 6 bands of 63 rows (1 152 workgroups, one round); a batch of 256 ->
    // 4 bands of 94 rows (6 144 workgroups, 5 rounds x 124 rows; ONE band of 376 rows -- what "as many bands as fit the slots" came to there --
    // is 2 rounds x 406: `k_st_eig_fused` 748 -> 590 us per launch, tools/st_bands.sh)
    int gyw = 1;
    {
      const int gy_max = H / 12 > 1 ? H / 12 : 1;
      long long best = -1;
      for (int g = 1; g <= gy_max; g++) {
        const int rbg = vo_div_up(H, g), gyg = vo_div_up(H, rbg);
        const long long rounds = ((long long)gx * gyg * B + 1279) / 1280, cost = rounds * (rbg + 30);
        if (best < 0 || cost < best) { best = cost; gyw = g; }
      }
    }
    int rb = vo_div_up(H, gyw);
    if (c->tune.st_band_rows > 0) rb = c->tune.st_band_rows < H ? c->tune.st_band_rows : H;
    const int gy = vo_div_up(H, rb);
    n_blockmax = gx * gy;
    // vo_tuning.st_separate_nms: the eigenvalue map goes through HBM to the separate k_st_nms, as in round 1
    const bool nms_fused = !c->tune.st_separate_nms;
    const int xcd_remap = (!c->tune.xcd_remap_off && B % 8 == 0) ? 1 : 0;
    const bool restore = !keep && !d_user_mask && nms_fused;
    hipLaunchKernelGGL(k_st_eig_fused<15>, dim3(gx, gy, B), dim3(256), 0, c->stream, F.img[0], c->lvl_px[0], c->lv[0].pitch, W, H, rb, s2,
                       s->d_mask, (keep || !nms_fused) ? s->d_eig : nullptr, s->d_blockmax, prm->quality_level, s->d_cand, s->d_nraw,
                       restore ? 1 : 0, nms_fused ? 1 : 0, xcd_remap);
    if (!nms_fused)
      hipLaunchKernelGGL(k_st_nms, dim3(vo_div_up(W - 2, 256), vo_div_up(H - 2, ST_NMS_ROWS), B), dim3(256), 0, c->stream, s->d_eig,
                         s->d_mask, W, H, prm->quality_level, s->d_blockmax, n_blockmax, s->d_cand, s->d_scalars, c->slab_seq, s->d_nraw);
    s->mask_clean = restore; s->eig_valid = keep || !nms_fused;
  } else {
    n_blockmax = vo_div_up(W, 256) * vo_div_up(H, ST_ROWS);
    hipLaunchKernelGGL(k_st_vsum_eig, dim3(vo_div_up(W, 256), vo_div_up(H, ST_ROWS), B), dim3(256), 0, c->stream, s->d_h,
                       s->d_mask, W, H, r, s2, s->d_eig, s->d_blockmax, prm->use_harris ? 1 : 0, prm->harris_k);
    hipLaunchKernelGGL(k_st_nms, dim3(vo_div_up(W - 2, 256), vo_div_up(H - 2, ST_NMS_ROWS), B), dim3(256), 0, c->stream, s->d_eig,
                       s->d_mask, W, H, prm->quality_level, s->d_blockmax, n_blockmax, s->d_cand, s->d_scalars, c->slab_seq, s->d_nraw);
    s->mask_clean = false; s->eig_valid = true;
  }
  const int use_dist = prm->min_distance >= 1.0 ? 1 : 0;
  // grid cell: >= min_distance (3x3 neighbourhood then covers the exclusion radius), coarse enough to fit LDS
  int cell = use_dist ? (int)ceil(prm->min_distance) : 1;
  if (cell < 1) cell = 1;
  while (((W + cell - 1) / cell) * ((H + cell - 1) / cell) > ST_MAX_CELLS) cell++;
  const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
  const double md2 = prm->min_distance * prm->min_distance;
  // the short-chunk instance keeps min(cap - accepted, ...) keys per pass: with a corner limit near its capacity the last corners would come
  // one tiny chunk -- and one rescan of the raw list -- at a time, so limits above half of it take the large instance
  if (limit_dev && (prm->max_corners > 0 && prm->max_corners <= ST_CAP_CLOSED_LOOP / 2))
    hipLaunchKernelGGL(k_st_select<ST_CAP_CLOSED_LOOP>, dim3(B), dim3(1024), st_sel_lds(ST_CAP_CLOSED_LOOP), c->stream, s->d_cand,
                       s->d_scalars, W, H, cell, gw, gh, md2, use_dist, prm->max_corners, s->d_out, c->slab_seq, c->d_dbg,
                       s->d_blockmax, n_blockmax, prm->quality_level, s->d_nraw, limit_dev);
  else
    hipLaunchKernelGGL(k_st_select<ST_CAND_CAP>, dim3(B), dim3(1024), st_sel_lds(ST_CAND_CAP), c->stream, s->d_cand,
                       s->d_scalars, W, H, cell, gw, gh, md2, use_dist, prm->max_corners, s->d_out, c->slab_seq, c->d_dbg,
                       s->d_blockmax, n_blockmax, prm->quality_level, s->d_nraw, limit_dev);
  VO_HIP(c, hipGetLastError());
  s->last_max_corners = prm->max_corners;
  return VO_OK;
}

// out_pts [batch][max_corners][2], n_out [batch]
static int32_t st_fetch(vo_ctx* c, float* out_pts, int32_t* n_out) {
  vo_st_ws* s = c->st;
  const int B = c->batch;
  const int mc = s->last_max_corners > 0 ? s->last_max_corners : ST_OUT_CAP;
  std::vector<uint32_t> sc((size_t)4 * B);
  VO_HIP(c, hipMemcpy2DAsync(sc.data(), 16, c->d_slab + c->off_st_scalars, c->slab_seq, 16, B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(out_pts, sizeof(float) * 2 * (size_t)mc, c->d_slab + c->off_st_out, c->slab_seq,
                             sizeof(float) * 2 * (size_t)mc, B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  for (int b = 0; b < B; b++) {
    if (sc[4 * b + 2] == 0xFFFFFFFFu) { n_out[b] = 0; return vo_fail(c, VO_E_CAPACITY, "shi_tomasi: the strongest NMS candidates the selection holds (16384; 4096 with a device-side corner limit) did not yield max_corners corners, or more than 262144 candidates"); }
    n_out[b] = (int32_t)sc[4 * b + 2];
  }
  return VO_OK;
}

// cur_pts [batch][n_cur][2], mask [batch][h][w] (optional), out_pts [batch][max_corners][2], n_out [batch]
extern "C" int32_t vo_shi_tomasi(vo_ctx* c, const float* cur_pts, int32_t n_cur, int32_t mask_radius, const uint8_t* mask,
                                 const vo_st_params* prm, float* out_pts, int32_t* n_out) {
  if (!c) return VO_E_INVALID;
  vo_st_params def;
  if (!prm) { vo_st_default_params(&def); prm = &def; }
  VO_CHECK(c, out_pts && n_out, VO_E_INVALID, "null output");
  VO_CHECK(c, n_cur >= 0 && n_cur <= c->max_pts && (n_cur == 0 || cur_pts), VO_E_CAPACITY, "bad cur_pts");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = st_init(c);
  if (r != VO_OK) return r;
  vo_st_ws* s = c->st;
  const size_t pts_seq = sizeof(float) * 2 * (size_t)c->max_pts;
  if (n_cur > 0)
    VO_HIP(c, hipMemcpy2DAsync(s->d_pts, pts_seq, cur_pts, sizeof(float) * 2 * n_cur, sizeof(float) * 2 * n_cur, c->batch,
                               hipMemcpyHostToDevice, c->stream));
  if (mask) VO_HIP(c, hipMemcpyAsync(s->d_user_mask, mask, (size_t)c->width * c->height * c->batch, hipMemcpyHostToDevice, c->stream));
  r = st_launch(c, s->d_pts, pts_seq, n_cur, mask_radius, mask ? s->d_user_mask : nullptr, prm, nullptr, true);
  if (r != VO_OK) return r;
  return st_fetch(c, out_pts, n_out);
}

extern "C" int32_t vo_shi_tomasi_resident(vo_ctx* c, int32_t n_cur, int32_t mask_radius, const vo_st_params* prm) {
  if (!c) return VO_E_INVALID;
  return vo_shi_tomasi_resident_counts(c, n_cur, mask_radius, prm, c->d_pt_counts, nullptr);   // (counts: non-null only while a vo_tracks_* table is seeded)
}

int32_t vo_shi_tomasi_resident_counts(vo_ctx* c, int32_t n_cur, int32_t mask_radius, const vo_st_params* prm, const int32_t* d_counts,
                                      const int32_t* d_limit) {
  vo_st_params def;
  if (!prm) { vo_st_default_params(&def); prm = &def; }
  VO_CHECK(c, n_cur >= 0 && n_cur <= c->n_resident, VO_E_INVALID, "n_cur exceeds the resident point set");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = st_init(c);
  if (r != VO_OK) return r;
  const bool dev_limit = !c->tune.st_host_limit;                                                           // (vo_tuning: A/B, read per call -- tests compare both)
  return st_launch(c, vo_slab<const float>(c, vo_off_p(c)), c->slab_seq, n_cur, mask_radius, nullptr, prm, d_counts,
                   c->tune.st_keep_eig != 0, dev_limit ? d_limit : nullptr);
}

extern "C" int32_t vo_shi_tomasi_fetch(vo_ctx* c, float* out_pts, int32_t* n_out) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->st && out_pts && n_out, VO_E_STATE, "no shi_tomasi call to fetch");
  VO_HIP(c, hipSetDevice(c->device));
  return st_fetch(c, out_pts, n_out);
}

// eig_out [batch][h][w], mask_out [batch][h][w], n_candidates [batch]
extern "C" int32_t vo_shi_tomasi_read(vo_ctx* c, float* eig_out, uint8_t* mask_out, int32_t* n_candidates) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->st, VO_E_STATE, "no shi_tomasi call yet");
  VO_CHECK(c, c->st->eig_valid || (!eig_out && !mask_out), VO_E_STATE,
           "the last launch was a resident one: it keeps neither the eigenvalue map nor the mask (vo_tuning.st_keep_eig makes it)");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (c->stream2) VO_HIP(c, hipStreamSynchronize(c->stream2));
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
  const size_t np = (size_t)c->width * c->height * c->batch;
  if (eig_out) VO_HIP(c, hipMemcpy(eig_out, c->st->d_eig, np * sizeof(float), hipMemcpyDeviceToHost));
  if (mask_out) VO_HIP(c, hipMemcpy(mask_out, c->st->d_mask, np, hipMemcpyDeviceToHost));
  if (n_candidates) {
    std::vector<uint32_t> sc((size_t)4 * c->batch);
    VO_HIP(c, hipMemcpy2D(sc.data(), 16, c->d_slab + c->off_st_scalars, c->slab_seq, 16, c->batch, hipMemcpyDeviceToHost));
    for (int b = 0; b < c->batch; b++) n_candidates[b] = (int32_t)sc[4 * b + 1];
  }
  return VO_OK;
}
