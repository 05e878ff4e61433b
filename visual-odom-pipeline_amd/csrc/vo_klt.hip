// Pyramidal Lucas-Kanade tracker for gfx950: ONE WAVE (64 lanes) PER KEYPOINT, all pyramid levels
// in one launch, template in registers, exact integer arithmetic.
//
// Replaces cv2.calcOpticalFlowPyrLK(prev, cur, p0, None, winSize=(31,31), maxLevel=3,
// criteria=(EPS|COUNT, 30, 0.03)) at /root/reference/src/extractor/extractor.py:44-45,65-66.
// Algorithm = OpenCV 4.4 video/lkpyramid.cpp LKTrackerInvoker, with the 2x2
// normal matrix / mismatch vector summed EXACTLY in integers (oracle/vo_oracle.c acc_mode = 1), so
// results are bit-identical to the CPU oracle independent of summation order.
//
// Lane <-> window mapping (window <= 31x31, read footprint 32 rows x 34 bytes):
//   lane = r*16 + cp,  cp = 0..15 (column pair), r = 0..3;  step s = 0..7 covers window row 8r + s,
//   columns 2cp and 2cp+1.  One unaligned dword load per lane per step fetches the 3 bytes the two
//   bilinear footprints need from the top row; the bottom row of step s is the SAME lane's top row of
//   step s + 1 (already in a register, its byte gathers are shared), only step 7 takes it from the
//   lane 16 further up (row group r + 1, step 0: one ds_bpermute per iteration).  => 8 dword loads per
//   lane per LK iteration for a 1 KB window, served by L1/L2 (a pyramid level is <= 0.6 MB; HBM sees it once).
//   The 16 lanes of a row group read 34 CONSECUTIVE bytes, so a load instruction touches 4 rows = 4-8 cache lines and each
//   quad of lanes one line.  (With lane = cp*4 + r -- a quad = four different rows, chosen in round 1 for a one-instruction
//   DPP neighbour exchange -- the texture addresser issued 28 cache accesses per load instruction and was busy 91 % of the
//   kernel: the kernel was bound by it, not by the vector ALUs; profiles/r02_pmc_klt_*.txt.)
//   The template (I, Ix, Iy at 16 pixels per lane) lives in 24 VGPRs across all iterations.
#include "vo_internal.h"

#include <stdlib.h>
#include <string.h>

#define W_BITS 14

struct klt_level_args {
  const uint8_t* imgI;    // sequence 0; sequence b at + b * seq_px pixels
  const uint32_t* derI;   // (4 Ix | 4 Iy << 16) per pixel
  const uint8_t* imgJ;
  size_t seq_px;
  int w, h, pitch;
};

struct klt_args {
  klt_level_args lv[VO_MAX_LEVELS];
  size_t slab_seq;        // byte stride between the sequences' point / status / err arrays
  size_t iters_seq;       // int32 stride between the sequences' iteration tables
  int top, win, max_count, n, iters_stride;
  int xcd_remap;             // batch % 8 == 0: keep every sequence on ONE XCD (see k_klt_track)
  float min_eig_num;         // minEig < threshold  <=>  numerator < min_eig_num  (klt_min_eig_numerator: no division per level)
  float eps_lo, eps_hi;      // |delta|^2 in float below / above these decides the convergence test; in between: float64
  double eps2;
};

// Window loads go through buffer instructions: address = descriptor base (the level's image: scalar registers) + ONE per-lane
// offset that is fixed for the level (vector register) + the wave-uniform window origin advanced per row (scalar register).
// With global_load the compiler kept the row advance in vector registers: one v_add per image load and a 64-bit
// v_lshl_add_u64 more per derivative load -- 8 / 24 vector instructions per LK iteration / template in a kernel that is bound
// by vector-instruction issue (tools/vmem_probe.hip).  Unaligned dwords are fine (same rules as global_load).
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t klt_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);   // raw, 4 GB range, no swizzle
}

// every lane receives the value of the lane 16 further up (the same column pair of the next row group; lanes 48..63 wrap
// around to rows that only masked pixels use): ds_bpermute_b32, the LDS crossbar -- no vector-ALU or memory-pipe slot
__device__ __forceinline__ uint32_t row_next(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, (int)v);
}

// exact wave-wide sum of an int32 per lane whose total fits in int32; result uniform
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
  return __builtin_amdgcn_readlane(v, 63);
}

// exact wave-wide sum of arbitrary int32 per lane (the total needs up to 36 bits), returned as the float nearest
// to the exact integer: the two 16-bit-split partial sums are combined in float64 (exact, < 2^53) and rounded ONCE,
// which is bit-identical to converting the 64-bit integer sum to float (what the CPU oracle does) and far cheaper
// than the software int64 -> float conversion.
__device__ __forceinline__ float wave_sum_exact_f32(int v) {
  const int lo = wave_sum_i32(v & 0xFFFF);
  const int hi = wave_sum_i32(v >> 16);
  return (float)((double)hi * 65536.0 + (double)lo);
}

// Four (two) wave-wide int32 sums at once as a reduce-scatter: after two quad exchanges lane l holds the quad sum of value
// number l & 3, the remaining steps (row rotations, row swaps) are multiples of 4 lanes and keep that assignment -- 15 + 4
// instructions instead of 4 x 7.  Totals must fit int32 (callers pass 16-bit halves).
template <int CTRL>
__device__ __forceinline__ int klt_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

__device__ __forceinline__ int klt_rows_sum(int z) {            // sum over the 16 quads, class (lane & 3) preserved
  z += klt_dpp<0x124>(z);                                       // row_ror:4
  z += klt_dpp<0x128>(z);                                       // row_ror:8
  {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)z, (unsigned)z, false, false);
    z = (int)r[0] + (int)r[1];
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)z, (unsigned)z, false, false);
    z = (int)r[0] + (int)r[1];
  }
  return z;
}

__device__ __forceinline__ void wave_sum4_i32(int a, int b, int c, int d, int lane, int& sa, int& sb, int& sc, int& sd) {
  const bool odd = lane & 1, up = lane & 2;
  const int x = (odd ? b : a) + klt_dpp<0xB1>(odd ? a : b);     // quad_perm [1,0,3,2]: even lanes sum a, odd lanes sum b (pairs)
  const int y = (odd ? d : c) + klt_dpp<0xB1>(odd ? c : d);     //                      even lanes sum c, odd lanes sum d
  int z = (up ? y : x) + klt_dpp<0x4E>(up ? x : y);             // quad_perm [2,3,0,1]: lane & 3 = 0, 1, 2, 3 <-> a, b, c, d (quads)
  z = klt_rows_sum(z);
  sa = __builtin_amdgcn_readlane(z, 0); sb = __builtin_amdgcn_readlane(z, 1);
  sc = __builtin_amdgcn_readlane(z, 2); sd = __builtin_amdgcn_readlane(z, 3);
}

__device__ __forceinline__ void wave_sum2_i32(int a, int b, int lane, int& sa, int& sb) {
  const bool odd = lane & 1;
  int z = (odd ? b : a) + klt_dpp<0xB1>(odd ? a : b);           // even lanes sum a, odd lanes sum b (pairs)
  z += klt_dpp<0x4E>(z);                                        // quads
  z = klt_rows_sum(z);
  sa = __builtin_amdgcn_readlane(z, 0); sb = __builtin_amdgcn_readlane(z, 1);
}

// Two / three wave-wide sums whose QUAD partial sums still fit int32 (callers guarantee |per-lane value| < 2^29): the first
// two butterfly stages run on the full 32-bit values as a reduce-scatter (lane & 1 selects the value), and only the 16 quad
// sums are split into 16-bit halves -- the spare lane class of every quad carries the high halves, so ONE row reduction
// delivers low and high totals of both values (15 vector instructions instead of 21 for the four pre-split halves).
__device__ __forceinline__ void wave_sum2_wide(int a, int b, int lane, int& la, int& ha, int& lb, int& hb) {
  const bool odd = lane & 1, up = lane & 2;
  int z = (odd ? b : a) + klt_dpp<0xB1>(odd ? a : b);           // pairs: even lanes a, odd lanes b
  z += klt_dpp<0x4E>(z);                                        // quads: lanes 0, 2 hold a's quad sum, lanes 1, 3 b's
  int w = up ? (z >> 16) : (z & 0xFFFF);                        // lane & 3 = 0: lo a, 1: lo b, 2: hi a, 3: hi b
  w = klt_rows_sum(w);
  la = __builtin_amdgcn_readlane(w, 0); lb = __builtin_amdgcn_readlane(w, 1);
  ha = __builtin_amdgcn_readlane(w, 2); hb = __builtin_amdgcn_readlane(w, 3);
}

__device__ __forceinline__ void wave_sum3_wide(int a, int b, int c, int lane, int& la, int& ha, int& lb, int& hb, int& lc, int& hc) {
  const bool odd = lane & 1, up = lane & 2;
  const int x = (odd ? b : a) + klt_dpp<0xB1>(odd ? a : b);     // pairs: even lanes a, odd lanes b
  const int y = c + klt_dpp<0xB1>(c);                           // pairs of c in every lane
  const int z = (up ? y : x) + klt_dpp<0x4E>(up ? x : y);       // quads: lane & 3 = 0: a, 1: b, 2 and 3: c
  const int w0 = klt_rows_sum(((lane & 3) == 3) ? (z >> 16) : (z & 0xFFFF));   // lo a, lo b, lo c, hi c
  const int w1 = klt_rows_sum(z >> 16);                                        // hi a, hi b
  la = __builtin_amdgcn_readlane(w0, 0); lb = __builtin_amdgcn_readlane(w0, 1);
  lc = __builtin_amdgcn_readlane(w0, 2); hc = __builtin_amdgcn_readlane(w0, 3);
  ha = __builtin_amdgcn_readlane(w1, 0); hb = __builtin_amdgcn_readlane(w1, 1);
}

// float nearest to the exact integer hi * 2^16 + lo (lo < 2^20).  It fits int32 whenever |hi| < 2^14 -- always, except for
// gross mismatches -- and then ONE v_cvt_f32_i32 rounds it exactly like the float64 route (5 quarter-rate instructions).
__device__ __forceinline__ float klt_combine(int hi, int lo) {
  if ((unsigned)(hi + 16384) < 32768u) return (float)(hi * 65536 + lo);
  return (float)((double)hi * 65536.0 + (double)lo);
}

// cvRound(x) for 0 <= x <= 2^14: adding 1.5 * 2^23 leaves round-to-nearest-even(x) in the low mantissa bits (one float add and one
// integer subtract at full rate instead of v_rndne_f32 + v_cvt_i32_f32 at quarter rate)
__device__ __forceinline__ int klt_round(float x) { return __float_as_int(x + 12582912.f) - 0x4B400000; }

// the four bilinear weights as packed 16-bit pairs wt = iw00 | iw01 << 16, wb = iw10 | iw11 << 16.  The rounded values sit in
// the low 16 bits of the magic-number sums (weights <= 2^14), so the byte gathers take them from there directly
__device__ __forceinline__ void lk_weights(float a, float b, uint32_t& wt, uint32_t& wb) {
  const uint32_t r00 = (uint32_t)__float_as_int((1.f - a) * (1.f - b) * (float)(1 << W_BITS) + 12582912.f);
  const uint32_t r01 = (uint32_t)__float_as_int(a * (1.f - b) * (float)(1 << W_BITS) + 12582912.f);
  const uint32_t r10 = (uint32_t)__float_as_int((1.f - a) * b * (float)(1 << W_BITS) + 12582912.f);
  const uint32_t iw11 = (uint32_t)(1 << W_BITS) + 3u * 0x4B400000u - (r00 + r01 + r10);
  wt = __builtin_amdgcn_perm(r01, r00, 0x05040100u);
  wb = __builtin_amdgcn_perm(iw11, r10, 0x05040100u);
}

__device__ __forceinline__ float uniform_f(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

// D = a.lo * b.lo + a.hi * b.hi + c  (signed 16-bit halves) -> v_dot2c_i32_i16
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
// The same product with the accumulator taken from a THIRD operand (VOP3P v_dot2_i32_i16): hipcc selects the two-address
// v_dot2c form for the plain builtin and then needs a v_mov to preload every rounding constant / zero (16 per LK iteration);
// the clamp bit exists only in the three-address encoding, so asking for it selects that form.  No sum here comes near
// the int32 range (|taps| <= 255 * 2^14), so the saturation never acts and the value is the plain dot product.
__device__ __forceinline__ int dot2k(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, true);
}
// (lo16(a) | lo16(b) << 16) -> one v_perm_b32
__device__ __forceinline__ uint32_t pack_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// bytes (k, k+1) of a dword widened to two 16-bit halves
__device__ __forceinline__ uint32_t bytes01(uint32_t t) { return __builtin_amdgcn_perm(0u, t, 0x0c010c00u); }
__device__ __forceinline__ uint32_t bytes12(uint32_t t) { return __builtin_amdgcn_perm(0u, t, 0x0c020c01u); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_abs(uint32_t a) {
  const s16x2 v = __builtin_bit_cast(s16x2, a);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, -v));
}

// bilinear samples (5 fractional bits) of the two pixels a lane owns in one step, packed (v0 | v1 << 16).
// T/B: top / bottom row dwords (3 useful bytes each); wt = iw00 | iw01 << 16, wb = iw10 | iw11 << 16.
__device__ __forceinline__ uint32_t sample2(uint32_t T, uint32_t B, uint32_t wt, uint32_t wb) {
  const int s0 = dot2(bytes01(B), wb, dot2k(bytes01(T), wt, 1 << (W_BITS - 5 - 1)));
  const int s1 = dot2(bytes12(B), wb, dot2k(bytes12(T), wt, 1 << (W_BITS - 5 - 1)));
  // (s >> 9) of both sums packed: bytes 1..2 of each sum are s >> 8 (0 <= s < 2^24), one packed 16-bit shift finishes --
  // two instructions instead of two shifts and a pack
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 h = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm((uint32_t)s1, (uint32_t)s0, 0x06050201u));
  return __builtin_bit_cast(uint32_t, h >> (unsigned short)(W_BITS - 5 - 8));
}

// interpolated derivative of one pixel, times 2^16: top pair / bottom pair already gathered as (left | right << 16).  The
// derivative image holds 4 x Scharr (vo_frame.hip), so the sum is 4 (s + 2^13) and the value OpenCV keeps, (s + 2^13) >> 14, is its
// UPPER HALF: the byte gather that packs two pixels reads it from there (and zeroes masked pixels): one instruction per pixel pair
// instead of two shifts, a pack and a mask
__device__ __forceinline__ uint32_t deriv1(uint32_t top, uint32_t bot, uint32_t wt, uint32_t wb) {
  return (uint32_t)dot2(bot, wb, dot2k(top, wt, 1 << (W_BITS + 1)));
}

// WAVES = minimum waves per SIMD the register allocation must allow (6: 79 VGPRs, no scratch -- the default; 5: 81; 4 and 5 measured 3 % and 2 % slower)
template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_klt_track(klt_args A, const float* __restrict__ p0, float* __restrict__ p1,
                                                  uint8_t* __restrict__ status, float* __restrict__ err,
                                                  int32_t* __restrict__ iters, unsigned long long* __restrict__ dbg,
                                                  const int32_t* __restrict__ counts) {
  // Workgroups are dealt to the 8 XCDs round-robin in dispatch order and every XCD has its own 4 MB L2.  With the
  // plain (point, sequence) grid all XCDs work on the same sequence and each pulls its own copy of that pyramid
  // (3.7 MB) from HBM / MALL -- 8x the bytes.  Remapped, XCD k tracks sequences k, k + 8, ... on its own: one
  // sequence's pyramids fit its L2 and cross the fabric once.
  int pt = blockIdx.x, bseq = blockIdx.y;
  if (A.xcd_remap) {
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = id >> 3;
    bseq = (int)(id & 7u) + 8 * (int)(q / (unsigned)A.n);
    pt = (int)(q % (unsigned)A.n);
  }
  if (pt >= A.n) return;
  const int lane = threadIdx.x;
  if (iters) iters += (size_t)bseq * A.iters_seq;
  const bool dead = counts && pt >= counts[bseq];    // track table: this sequence has fewer live points
  // levels the call does not visit (above `top`, or every level of a dead slot) are reported as -1 by the wave itself
  // (a memset of the whole table used to precede every launch)
  if (iters && lane < A.iters_stride && (dead || lane > A.top)) iters[pt * A.iters_stride + lane] = -1;
  if (dead) return;
  p0 = vo_seq(p0, A.slab_seq, bseq); p1 = vo_seq(p1, A.slab_seq, bseq);
  status = vo_seq(status, A.slab_seq, bseq); err = vo_seq(err, A.slab_seq, bseq);
  unsigned long long* dbgk = (pt == A.n / 2 && bseq == 0 && dbg) ? dbg + 24 : nullptr;   // diagnostic stamps of one wave
  VO_STAMP(dbgk, 0);
  const int cp = lane & 15, r = lane >> 4;
  const int win = A.win;
  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);

  const float p0x = uniform_f(p0[2 * pt]), p0y = uniform_f(p0[2 * pt + 1]);
  float outx = 0.f, outy = 0.f;   // nextPts[pt]
  int st = 1;
  float errv = 0.f;

  // validity of the lane's two columns as 16-bit masks (lo = column 2cp, hi = column 2cp + 1)
  const uint32_t colmask = ((2 * cp < win) ? 0x0000FFFFu : 0u) | ((2 * cp + 1 < win) ? 0xFFFF0000u : 0u);
  const uint32_t colones = colmask & 0x00010001u;
  // v_perm selector "upper halves of (a, b)" with the constant-zero code 0x0c for the columns outside the window
  const uint32_t colsel = (0x07060302u & colmask) | (0x0c0c0c0cu & ~colmask);

  for (int level = A.top; level >= 0; level--) {
    klt_level_args L = A.lv[level];
    L.imgI += (size_t)bseq * L.seq_px; L.derI += (size_t)bseq * L.seq_px; L.imgJ += (size_t)bseq * L.seq_px;
    const float scale = __int_as_float((127 - level) << 23);       // 2^-level, exactly what 1.f / (float)(1 << level) gives (no division)
    float prevx = p0x * scale, prevy = p0y * scale;
    float nextx, nexty;
    if (level == A.top) { nextx = prevx; nexty = prevy; }
    else { nextx = outx * 2.f; nexty = outy * 2.f; }
    outx = nextx; outy = nexty;
    int n_it = -1;

    prevx -= half; prevy -= half;
    const float fpx = floorf(prevx), fpy = floorf(prevy);        // (float)(int)floorf(x) == floorf(x): the fraction needs no int -> float convert
    const int ipx = (int)fpx, ipy = (int)fpy;
    if (ipx < -win || ipx >= L.w || ipy < -win || ipy >= L.h) {
      if (level == 0) { st = 0; errv = 0.f; }
      if (iters && lane == 0) iters[pt * A.iters_stride + level] = n_it;
      continue;
    }
    const uint32_t lane_off = (uint32_t)(8 * r * L.pitch + 2 * cp);    // the lane's corner of the 32 x 34 footprint
    uint32_t wt, wb;
    lk_weights(prevx - fpx, prevy - fpy, wt, wb);

    // ---- template: packed pairs of I (5 frac bits), Ix, Iy for the lane's 16 pixels; exact A11, A12, A22 ----
    uint32_t tI[8], tX[8], tY[8];
    {
      uint32_t T[8], D0[8], D1[8], D2[8];
      // addresses = level base (scalar registers) + a 32-bit offset: the wave-uniform window origin, advanced per row on the
      // scalar unit, plus ONE per-lane offset that is fixed for the level (it was a chain of 64-bit vector adds per row)
      const uint32_t uo = (uint32_t)(ipy + VO_PAD) * (uint32_t)L.pitch + (uint32_t)(ipx + VO_PAD);
      const __amdgpu_buffer_rsrc_t rI = klt_rsrc(L.imgI), rD = klt_rsrc(L.derI);
#pragma unroll
      for (int s = 0; s < 8; s++) {
        const uint32_t o = uo + (uint32_t)s * (uint32_t)L.pitch;          // wave-uniform
        T[s] = __builtin_amdgcn_raw_buffer_load_b32(rI, (int)lane_off, (int)o, 0);
        // three consecutive pixels: one 12-byte load.  (8 bytes + the neighbour lane's first pixel through a DPP row shift was
        // measured: the same kernel time -- the data path is not priced per byte.)
        const u32x3 d3 = __builtin_amdgcn_raw_buffer_load_b96(rD, (int)(lane_off * 4u), (int)(o * 4u), 0);
        D0[s] = d3[0]; D1[s] = d3[1]; D2[s] = d3[2];
      }
      // row 8r + 8 = step 0 of row group r + 1 (lanes of r == 3 receive a row that only masked pixels use)
      const uint32_t T8 = row_next(T[0], lane), D08 = row_next(D0[0], lane), D18 = row_next(D1[0], lane), D28 = row_next(D2[0], lane);
      int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
      for (int s = 0; s < 8; s++) {
        const uint32_t B = (s < 7) ? T[(s + 1) & 7] : T8;
        const uint32_t E0 = (s < 7) ? D0[(s + 1) & 7] : D08;
        const uint32_t E1 = (s < 7) ? D1[(s + 1) & 7] : D18;
        const uint32_t E2 = (s < 7) ? D2[(s + 1) & 7] : D28;
        tI[s] = sample2(T[s], B, wt, wb);
        const uint32_t x0 = deriv1(pack_lo(D0[s], D1[s]), pack_lo(E0, E1), wt, wb);
        const uint32_t y0 = deriv1(pack_hi(D0[s], D1[s]), pack_hi(E0, E1), wt, wb);
        const uint32_t x1 = deriv1(pack_lo(D1[s], D2[s]), pack_lo(E1, E2), wt, wb);
        const uint32_t y1 = deriv1(pack_hi(D1[s], D2[s]), pack_hi(E1, E2), wt, wb);
        const uint32_t sel = (8 * r + s < win) ? colsel : 0x0c0c0c0cu;    // rows / columns outside the window contribute nothing
        const uint32_t xp = __builtin_amdgcn_perm(x1, x0, sel), yp = __builtin_amdgcn_perm(y1, y0, sel);
        tX[s] = xp; tY[s] = yp;
        // the first step starts the three sums from an inline zero (three-address form: no preload)
        a11 = s ? dot2(xp, xp, a11) : dot2k(xp, xp, 0);
        a12 = s ? dot2(xp, yp, a12) : dot2k(xp, yp, 0);
        a22 = s ? dot2(yp, yp, a22) : dot2k(yp, yp, 0);
      }
      float A11, A12, A22;
      {
        // per lane 16 products of two int16 derivatives (|Scharr| <= 4080): < 2^28.01, a quad's sum < 2^30.01
        int l11, h11, l12, h12, l22, h22;
        wave_sum3_wide(a11, a12, a22, lane, l11, h11, l12, h12, l22, h22);
        A11 = klt_combine(h11, l11) * FLT_SCALE; A12 = klt_combine(h12, l12) * FLT_SCALE; A22 = klt_combine(h22, l22) * FLT_SCALE;
      }
      if (level == A.top) VO_STAMP(dbgk, 1);   // first template
      float D = A11 * A22 - A12 * A12;
      // minEig = num / (2 win^2) < minEigThreshold, decided on the numerator (threshold pre-divided exactly on the host)
      const float num = A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
      if (num < A.min_eig_num || D < 1.1920929e-07f) {
        if (level == 0) st = 0;
        if (iters && lane == 0) iters[pt * A.iters_stride + level] = n_it;
        continue;
      }
      D = 1.f / D;

      nextx -= half; nexty -= half;
      const __amdgpu_buffer_rsrc_t rJ = klt_rsrc(L.imgJ);
      float pdx = 0.f, pdy = 0.f;
      int j = 0;
      for (; j < A.max_count; j++) {
        const float fnx = floorf(nextx), fny = floorf(nexty);
        const int inx = (int)fnx, iny = (int)fny;
        if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
          if (level == 0) st = 0;
          break;
        }
        uint32_t jt, jb;
        lk_weights(nextx - fnx, nexty - fny, jt, jb);
        uint32_t Tj[8];
        const uint32_t uj = (uint32_t)(iny + VO_PAD) * (uint32_t)L.pitch + (uint32_t)(inx + VO_PAD);
#pragma unroll
        for (int s = 0; s < 8; s++) Tj[s] = __builtin_amdgcn_raw_buffer_load_b32(rJ, (int)lane_off, (int)(uj + (uint32_t)s * (uint32_t)L.pitch), 0);
        const uint32_t Tj8 = row_next(Tj[0], lane);
        int b1 = 0, b2 = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) {
          const uint32_t B = (s < 7) ? Tj[(s + 1) & 7] : Tj8;
          const uint32_t d = pk_sub(sample2(Tj[s], B, jt, jb), tI[s]);   // (diff0 | diff1 << 16), |diff| <= 8160
          b1 = s ? dot2(d, tX[s], b1) : dot2k(d, tX[0], 0);
          b2 = s ? dot2(d, tY[s], b2) : dot2k(d, tY[0], 0);
        }
        // per lane 16 products |diff| <= 8160 (255 << 5) times |derivative| <= 4080: < 2^28.99, a quad's sum < 2^30.99
        int l1, h1, l2, h2;
        wave_sum2_wide(b1, b2, lane, l1, h1, l2, h2);
        const float fb1 = klt_combine(h1, l1) * FLT_SCALE;
        const float fb2 = klt_combine(h2, l2) * FLT_SCALE;
        const float dx = (A12 * fb2 - A22 * fb1) * D;
        const float dy = (A12 * fb1 - A11 * fb2) * D;
        nextx += dx; nexty += dy;
        outx = nextx + half; outy = nexty + half;
        // |delta|^2 <= eps^2 is OpenCV's float64 test; its float32 value is within 2^-22 of it, so only a value between the
        // two guard constants needs the float64 evaluation
        const float d2 = dx * dx + dy * dy;
        bool conv;
        if (d2 < A.eps_lo) conv = true;
        else if (d2 > A.eps_hi) conv = false;
        else conv = (double)dx * (double)dx + (double)dy * (double)dy <= A.eps2;
        if (conv) { j++; break; }
        // fabs((double)x) < 0.01 for a float x  <=>  fabsf(x) <= (float)0.01: 0.01 lies strictly between that float and the next
        if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
          outx -= dx * 0.5f; outy -= dy * 0.5f;
          j++;
          break;
        }
        pdx = dx; pdy = dy;
        if (level == A.top && j == 0) VO_STAMP(dbgk, 2);   // first LK iteration
      }
      n_it = j;
      if (level == A.top) VO_STAMP(dbgk, 3);   // top level done
      if (level == 1) VO_STAMP(dbgk, 4);       // levels top-1 .. 1 done
      if (iters && lane == 0) iters[pt * A.iters_stride + level] = n_it;

      if (st && level == 0) {
        const float nx = outx - half, ny = outy - half;
        const float fnx = floorf(nx), fny = floorf(ny);
        const int inx = (int)fnx, iny = (int)fny;
        if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
          st = 0;
        } else {
          uint32_t jt, jb;
          lk_weights(nx - fnx, ny - fny, jt, jb);
          uint32_t Tj[8];
          const uint32_t uj = (uint32_t)(iny + VO_PAD) * (uint32_t)L.pitch + (uint32_t)(inx + VO_PAD);
#pragma unroll
          for (int s = 0; s < 8; s++) Tj[s] = __builtin_amdgcn_raw_buffer_load_b32(rJ, (int)lane_off, (int)(uj + (uint32_t)s * (uint32_t)L.pitch), 0);
          const uint32_t Tj8 = row_next(Tj[0], lane);
          int e = 0;
#pragma unroll
          for (int s = 0; s < 8; s++) {
            const uint32_t B = (s < 7) ? Tj[(s + 1) & 7] : Tj8;
            const uint32_t d = pk_abs(pk_sub(sample2(Tj[s], B, jt, jb), tI[s]));
            const uint32_t ones = (8 * r + s < win) ? colones : 0u;
            e = s ? dot2(d, ones, e) : dot2k(d, ones, 0);
          }
          const int ierr = wave_sum_i32(e);
          errv = (float)ierr * 1.f / (float)(32 * win * win);
        }
      }
    }
  }
  VO_STAMP(dbgk, 5);
  if (lane == 0) {
    p1[2 * pt] = outx; p1[2 * pt + 1] = outy;
    status[pt] = (uint8_t)st;
    err[pt] = st ? errv : 0.f;
  }
}

#ifdef VO_EXPERIMENTS
// ================================================================================================
// k_klt_track2: TWO keypoints per wave (experiment: compiled with -DVO_EXPERIMENTS only, selected by vo_tuning.klt_pair; slower than k_klt_track).  Of the ~150 vector instructions of an LK iteration ~62 are the same
// for every lane (weights, solve, tests, the cross-lane sums' tails): one wave per keypoint pays them per keypoint.  Here the two halves of
// a wave own one keypoint each -- lane = h * 32 + r * 16 + cp, a lane covers window rows 16 r + s (s = 0..15), columns 2 cp and 2 cp + 1 --
// so the per-pixel work per keypoint is unchanged (twice the steps on half the lanes) and the lane-uniform work is issued once per PAIR.
// The price: both halves stay in a level's loop for max(it_a, it_b) iterations (tools/klt_pairing_study.py: +8 % iterations for neighbours
// in the point list, -11.5 % instructions per keypoint net), the window origin is per half, so it rides in the per-lane offset instead of a
// scalar register, and the template needs 48 registers.  Same integer sums, same float expressions: bit-identical results.
// ================================================================================================
__device__ __forceinline__ uint32_t row_next2(uint32_t v, int lane) {       // the lane 16 further up INSIDE the half
  return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & 32) | ((lane + 16) & 31)) << 2, (int)v);
}
// sum over the two 16-lane rows of a half, lane class (lane & 3) preserved; every lane of the half receives its class's total
__device__ __forceinline__ int klt_half_sum(int z) {
  z += klt_dpp<0x124>(z);                                       // row_ror:4
  z += klt_dpp<0x128>(z);                                       // row_ror:8
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)z, (unsigned)z, false, false);
  return (int)r[0] + (int)r[1];                                 // rows 0 + 1 and rows 2 + 3: the halves
}
template <int K> __device__ __forceinline__ int klt_class(int z) {          // value of lane class K of the quad in all four lanes
  return __builtin_amdgcn_update_dpp(0, z, K * 0x55 /* quad_perm [K, K, K, K] */, 0xf, 0xf, true);
}

template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_klt_track2(klt_args A, const float* __restrict__ p0, float* __restrict__ p1,
                                                   uint8_t* __restrict__ status, float* __restrict__ err,
                                                   int32_t* __restrict__ iters, const int32_t* __restrict__ counts) {
  const int npair = (A.n + 1) >> 1;
  int pr = blockIdx.x, bseq = blockIdx.y;
  if (A.xcd_remap) {
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = id >> 3;
    bseq = (int)(id & 7u) + 8 * (int)(q / (unsigned)npair);
    pr = (int)(q % (unsigned)npair);
  }
  if (pr >= npair) return;
  const int lane = threadIdx.x, hsel = lane >> 5, l32 = lane & 31;
  const int pt = 2 * pr + hsel;
  if (iters) iters += (size_t)bseq * A.iters_seq;
  const int nlive = counts ? min(counts[bseq], A.n) : A.n;
  const bool exists = pt < A.n, dead = pt >= nlive;
  if (iters && exists && l32 < A.iters_stride && (dead || l32 > A.top)) iters[pt * A.iters_stride + l32] = -1;
  if (2 * pr >= nlive) return;                          // both slots dead (wave-uniform)
  p0 = vo_seq(p0, A.slab_seq, bseq); p1 = vo_seq(p1, A.slab_seq, bseq);
  status = vo_seq(status, A.slab_seq, bseq); err = vo_seq(err, A.slab_seq, bseq);
  const int cp = lane & 15, r = (lane >> 4) & 1;
  const int win = A.win;
  const float half = (float)(win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (float)(1 << 20);
  const int ptc = dead ? 2 * pr : pt;                   // a dead half reads its partner's point (never written back)
  const float p0x = p0[2 * ptc], p0y = p0[2 * ptc + 1];
  float outx = 0.f, outy = 0.f;
  int st = 1;
  float errv = 0.f;
  const uint32_t colmask = ((2 * cp < win) ? 0x0000FFFFu : 0u) | ((2 * cp + 1 < win) ? 0xFFFF0000u : 0u);
  const uint32_t colones = colmask & 0x00010001u;
  const uint32_t colsel = (0x07060302u & colmask) | (0x0c0c0c0cu & ~colmask);

  for (int level = A.top; level >= 0; level--) {
    klt_level_args L = A.lv[level];
    L.imgI += (size_t)bseq * L.seq_px; L.derI += (size_t)bseq * L.seq_px; L.imgJ += (size_t)bseq * L.seq_px;
    const float scale = __int_as_float((127 - level) << 23);
    float prevx = p0x * scale, prevy = p0y * scale;
    float nextx, nexty;
    if (level == A.top) { nextx = prevx; nexty = prevy; }
    else { nextx = outx * 2.f; nexty = outy * 2.f; }
    outx = nextx; outy = nexty;
    int n_it = -1;
    prevx -= half; prevy -= half;
    const float fpx = floorf(prevx), fpy = floorf(prevy);
    const int ipx = (int)fpx, ipy = (int)fpy;
    bool ok = !dead && !(ipx < -win || ipx >= L.w || ipy < -win || ipy >= L.h);
    if (!ok && !dead && level == 0) { st = 0; errv = 0.f; }
    const uint32_t lane_off = (uint32_t)(16 * r * L.pitch + 2 * cp);
    uint32_t tI[16], tX[16], tY[16];
    float A11 = 0.f, A12 = 0.f, A22 = 0.f, D = 0.f;
    if (ok) {
      uint32_t wt, wb;
      lk_weights(prevx - fpx, prevy - fpy, wt, wb);
      const uint32_t vo = lane_off + (uint32_t)(ipy + VO_PAD) * (uint32_t)L.pitch + (uint32_t)(ipx + VO_PAD);      // per lane: the half's origin
      const __amdgpu_buffer_rsrc_t rI = klt_rsrc(L.imgI), rD = klt_rsrc(L.derI);
      int a11 = 0, a12 = 0, a22 = 0;
      uint32_t T0 = 0; u32x3 d0 = {0, 0, 0};           // row 0 of the lane (its row-group neighbour needs it as the bottom row of step 15)
      uint32_t Tc = 0; u32x3 dc = {0, 0, 0};           // the row the next batch starts from
      // four batches of 4 steps: 5 (4) rows of loads in flight, then their arithmetic (all 17 rows at once would need 68 registers; the
      // template's 48 stay live for the whole level)
#pragma unroll
      for (int bt = 0; bt < 4; bt++) {
        uint32_t T[5]; u32x3 Dv[5];
        if (bt == 0) {
          T[0] = __builtin_amdgcn_raw_buffer_load_b32(rI, (int)vo, 0, 0);
          Dv[0] = __builtin_amdgcn_raw_buffer_load_b96(rD, (int)(vo * 4u), 0, 0);
          T0 = T[0]; d0 = Dv[0];
        } else { T[0] = Tc; Dv[0] = dc; }
#pragma unroll
        for (int k = 1; k < 5; k++) {
          const int row = 4 * bt + k;
          if (row < 16) {
            const uint32_t o = (uint32_t)row * (uint32_t)L.pitch;         // wave-uniform row advance (scalar)
            T[k] = __builtin_amdgcn_raw_buffer_load_b32(rI, (int)vo, (int)o, 0);
            Dv[k] = __builtin_amdgcn_raw_buffer_load_b96(rD, (int)(vo * 4u), (int)(o * 4u), 0);
          } else {
            T[k] = row_next2(T0, lane);
            Dv[k][0] = row_next2(d0[0], lane); Dv[k][1] = row_next2(d0[1], lane); Dv[k][2] = row_next2(d0[2], lane);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int s2 = 4 * bt + k;
          tI[s2] = sample2(T[k], T[k + 1], wt, wb);
          const uint32_t x0 = deriv1(pack_lo(Dv[k][0], Dv[k][1]), pack_lo(Dv[k + 1][0], Dv[k + 1][1]), wt, wb);
          const uint32_t y0 = deriv1(pack_hi(Dv[k][0], Dv[k][1]), pack_hi(Dv[k + 1][0], Dv[k + 1][1]), wt, wb);
          const uint32_t x1 = deriv1(pack_lo(Dv[k][1], Dv[k][2]), pack_lo(Dv[k + 1][1], Dv[k + 1][2]), wt, wb);
          const uint32_t y1 = deriv1(pack_hi(Dv[k][1], Dv[k][2]), pack_hi(Dv[k + 1][1], Dv[k + 1][2]), wt, wb);
          const uint32_t sel = (16 * r + s2 < win) ? colsel : 0x0c0c0c0cu;
          const uint32_t xp = __builtin_amdgcn_perm(x1, x0, sel), yp = __builtin_amdgcn_perm(y1, y0, sel);
          tX[s2] = xp; tY[s2] = yp;
          a11 = s2 ? dot2(xp, xp, a11) : dot2k(xp, xp, 0);
          a12 = s2 ? dot2(xp, yp, a12) : dot2k(xp, yp, 0);
          a22 = s2 ? dot2(yp, yp, a22) : dot2k(yp, yp, 0);
        }
        Tc = T[4]; dc = Dv[4];
      }
      {
        // per lane 32 products of two int16 derivatives (|Scharr| <= 4080): a quad's sum <= 4 * 32 * 4080^2 < 2^31, so the pair and quad
        // stages run on the full values (reduce-scatter), only the quad sums are split into 16-bit halves for the two rows of the half
        const bool odd = lane & 1, up = lane & 2;
        const int x = (odd ? a12 : a11) + klt_dpp<0xB1>(odd ? a11 : a12);     // pairs: even lanes a11, odd lanes a12
        const int y = a22 + klt_dpp<0xB1>(a22);                               // pairs of a22 in every lane
        const int z = (up ? y : x) + klt_dpp<0x4E>(up ? x : y);               // quads: lane & 3 = 0: a11, 1: a12, 2 and 3: a22
        const int w0 = klt_half_sum(((lane & 3) == 3) ? (z >> 16) : (z & 0xFFFF));   // lo a11, lo a12, lo a22, hi a22
        const int w1 = klt_half_sum(z >> 16);                                        // hi a11, hi a12
        A11 = klt_combine(klt_class<0>(w1), klt_class<0>(w0)) * FLT_SCALE;
        A12 = klt_combine(klt_class<1>(w1), klt_class<1>(w0)) * FLT_SCALE;
        A22 = klt_combine(klt_class<3>(w0), klt_class<2>(w0)) * FLT_SCALE;
      }
      D = A11 * A22 - A12 * A12;
      const float num = A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
      if (num < A.min_eig_num || D < 1.1920929e-07f) {
        if (level == 0) st = 0;
        ok = false;
      }
      D = 1.f / D;
    }
    nextx -= half; nexty -= half;
    const __amdgpu_buffer_rsrc_t rJ = klt_rsrc(L.imgJ);
    float pdx = 0.f, pdy = 0.f;
    int j = 0;
    bool act = ok && A.max_count > 0;
    if (ok) n_it = 0;
    while (__any(act)) {
      if (act) {
        const float fnx = floorf(nextx), fny = floorf(nexty);
        const int inx = (int)fnx, iny = (int)fny;
        if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
          if (level == 0) st = 0;
          act = false;
        } else {
          uint32_t jt, jb;
          lk_weights(nextx - fnx, nexty - fny, jt, jb);
          const uint32_t vj = lane_off + (uint32_t)(iny + VO_PAD) * (uint32_t)L.pitch + (uint32_t)(inx + VO_PAD);
          uint32_t Tj[16];
#pragma unroll
          for (int s = 0; s < 16; s++) Tj[s] = __builtin_amdgcn_raw_buffer_load_b32(rJ, (int)vj, (int)((uint32_t)s * (uint32_t)L.pitch), 0);
          const uint32_t Tj16 = row_next2(Tj[0], lane);
          int b1 = 0, b2 = 0;
#pragma unroll
          for (int s = 0; s < 16; s++) {
            const uint32_t B = (s < 15) ? Tj[(s + 1) & 15] : Tj16;
            const uint32_t d = pk_sub(sample2(Tj[s], B, jt, jb), tI[s]);
            b1 = s ? dot2(d, tX[s], b1) : dot2k(d, tX[0], 0);
            b2 = s ? dot2(d, tY[s], b2) : dot2k(d, tY[0], 0);
          }
          // per lane 32 products |diff| <= 8160 times |derivative| <= 4080: < 2^29.99 -- a quad's sum would leave int32, so the values are
          // split into 16-bit halves first and the four halves go through one reduce-scatter (lane & 3 = 0: lo b1, 1: lo b2, 2: hi b1, 3: hi b2)
          const bool odd = lane & 1, up = lane & 2;
          const int lo1 = b1 & 0xFFFF, hi1 = b1 >> 16, lo2 = b2 & 0xFFFF, hi2 = b2 >> 16;
          const int x = (odd ? lo2 : lo1) + klt_dpp<0xB1>(odd ? lo1 : lo2);
          const int y = (odd ? hi2 : hi1) + klt_dpp<0xB1>(odd ? hi1 : hi2);
          int z = (up ? y : x) + klt_dpp<0x4E>(up ? x : y);
          z = klt_half_sum(z);
          const int l1 = klt_class<0>(z), l2 = klt_class<1>(z), h1 = klt_class<2>(z), h2 = klt_class<3>(z);
          const float fb1 = klt_combine(h1, l1) * FLT_SCALE;
          const float fb2 = klt_combine(h2, l2) * FLT_SCALE;
          const float dx = (A12 * fb2 - A22 * fb1) * D;
          const float dy = (A12 * fb1 - A11 * fb2) * D;
          nextx += dx; nexty += dy;
          outx = nextx + half; outy = nexty + half;
          const float d2 = dx * dx + dy * dy;
          bool conv;
          if (d2 < A.eps_lo) conv = true;
          else if (d2 > A.eps_hi) conv = false;
          else conv = (double)dx * (double)dx + (double)dy * (double)dy <= A.eps2;
          j++;
          if (conv) act = false;
          else if (j > 1 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
            outx -= dx * 0.5f; outy -= dy * 0.5f;
            act = false;
          }
          pdx = dx; pdy = dy;
          if (j >= A.max_count) act = false;
        }
      }
    }
    if (ok) n_it = j;
    if (iters && exists && !dead && l32 == 0) iters[pt * A.iters_stride + level] = n_it;
    if (ok && st && level == 0) {
      const float nx = outx - half, ny = outy - half;
      const float fnx = floorf(nx), fny = floorf(ny);
      const int inx = (int)fnx, iny = (int)fny;
      if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
        st = 0;
      } else {
        uint32_t jt, jb;
        lk_weights(nx - fnx, ny - fny, jt, jb);
        const uint32_t vj = lane_off + (uint32_t)(iny + VO_PAD) * (uint32_t)L.pitch + (uint32_t)(inx + VO_PAD);
        uint32_t Tj[16];
#pragma unroll
        for (int s = 0; s < 16; s++) Tj[s] = __builtin_amdgcn_raw_buffer_load_b32(rJ, (int)vj, (int)((uint32_t)s * (uint32_t)L.pitch), 0);
        const uint32_t Tj16 = row_next2(Tj[0], lane);
        int e = 0;
#pragma unroll
        for (int s = 0; s < 16; s++) {
          const uint32_t B = (s < 15) ? Tj[(s + 1) & 15] : Tj16;
          const uint32_t d = pk_abs(pk_sub(sample2(Tj[s], B, jt, jb), tI[s]));
          const uint32_t ones = (16 * r + s < win) ? colones : 0u;
          e = s ? dot2(d, ones, e) : dot2k(d, ones, 0);
        }
        e += klt_dpp<0xB1>(e); e += klt_dpp<0x4E>(e);                    // quad total in every lane of the quad
        e = klt_half_sum(e);
        errv = (float)e * 1.f / (float)(32 * win * win);
      }
    }
  }
  if (exists && !dead && l32 == 0) {
    p1[2 * pt] = outx; p1[2 * pt + 1] = outy;
    status[pt] = (uint8_t)st;
    err[pt] = st ? errv : 0.f;
  }
}
#endif  // VO_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
extern "C" int32_t vo_klt_default_params(vo_klt_params* p) {
  if (!p) return VO_E_INVALID;
  p->win = 31; p->max_level = 3; p->max_count = 30; p->epsilon = 0.03; p->min_eig_threshold = 1e-4f; p->_pad = 0;
  return VO_OK;
}

// smallest float x with x / c >= thr in float arithmetic (c > 0): "num / c < thr" and "num < x" are the same predicate because
// a correctly rounded division is monotone in its numerator.  Bisection over the ordered float encodings.
static float klt_min_eig_numerator(float thr, float c) {
  auto from_ord = [](int64_t k) {            // order-preserving map of [-2^31 + 1, 2^31 - 1] onto the floats (NaNs excluded by the range)
    const uint32_t u = (k >= 0) ? (uint32_t)k : (0x80000000u | (uint32_t)(-k));
    float f; memcpy(&f, &u, 4); return f;
  };
  const int64_t inf = 0x7F800000ll;          // +infinity; -inf = -0x7F800000
  if (!(thr == thr)) return __builtin_nanf("");                                  // NaN threshold: the comparison is always false
  int64_t lo = -inf, hi = inf;               // invariant: f(lo) fails (or lo = -inf boundary), f(hi) holds
  { volatile float q = from_ord(lo) / c; if (q >= thr) return from_ord(lo); }    // every numerator passes
  { volatile float q = from_ord(hi) / c; if (!(q >= thr)) return from_ord(hi); } // only +inf ... none passes below it
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    volatile float q = from_ord(mid) / c;
    if (q >= thr) hi = mid; else lo = mid;
  }
  return from_ord(hi);
}

static int32_t klt_launch(vo_ctx* c, int n, const vo_klt_params* prm, size_t off_in, size_t off_out, const int32_t* counts) {
  VO_CHECK(c, c->n_pushed >= 2, VO_E_STATE, "need two pushed frames");
  VO_CHECK(c, n >= 0 && n <= c->max_pts, VO_E_CAPACITY, "n exceeds max_pts");
  VO_CHECK(c, prm && prm->win >= 3 && prm->win <= VO_MAX_WIN && (prm->win & 1), VO_E_INVALID, "win must be odd, 3..31");
  VO_CHECK(c, prm->max_level >= 0 && prm->max_level < VO_MAX_LEVELS, VO_E_INVALID, "bad max_level");
  if (n == 0) return VO_OK;
  klt_args A;
  const vo_frame& P = c->fr[c->cur ^ 1];
  const vo_frame& C = c->fr[c->cur];
  // effective top level: levels available in the frame store, truncated by this call's window rule
  int top = 0;
  for (int l = 1; l <= c->top && l <= prm->max_level; l++) {
    if (c->lv[l].w <= prm->win || c->lv[l].h <= prm->win) break;
    top = l;
  }
  for (int l = 0; l <= top; l++) {
    A.lv[l].imgI = P.img[l]; A.lv[l].derI = reinterpret_cast<const uint32_t*>(P.der[l]); A.lv[l].imgJ = C.img[l];
    A.lv[l].seq_px = c->lvl_px[l];
    A.lv[l].w = c->lv[l].w; A.lv[l].h = c->lv[l].h; A.lv[l].pitch = c->lv[l].pitch;
  }
  A.top = top; A.win = prm->win;
  A.xcd_remap = (!c->tune.xcd_remap_off && c->batch % 8 == 0) ? 1 : 0;
  int mc = prm->max_count; if (mc < 0) mc = 0; if (mc > 100) mc = 100;
  double eps = prm->epsilon; if (eps < 0) eps = 0; if (eps > 10) eps = 10;
  A.max_count = mc; A.eps2 = eps * eps; A.n = n;
  A.min_eig_num = klt_min_eig_numerator(prm->min_eig_threshold, (float)(2 * prm->win * prm->win));
  A.eps_lo = (float)(A.eps2 * (1.0 - 1e-6)); A.eps_hi = (float)(A.eps2 * (1.0 + 1e-6));
  A.iters_stride = prm->max_level + 1;
  A.slab_seq = c->slab_seq; A.iters_seq = (size_t)c->max_pts * VO_MAX_LEVELS;
  c->iters_stride = A.iters_stride;
  {
    vo_prof_scope prof(c, VO_PROF_KLT);   // brackets exactly this launch (bench.py roofline figure)
    const int waves = c->tune.klt_waves > 0 ? c->tune.klt_waves : 6;
#define VO_KLT_LAUNCH(WV) hipLaunchKernelGGL(k_klt_track<WV>, dim3(n, c->batch), dim3(64), 0, c->stream, A,                \
                       vo_slab<const float>(c, off_in), vo_slab<float>(c, off_out), vo_slab<uint8_t>(c, c->off_status),     \
                       vo_slab<float>(c, c->off_err), c->d_iters, c->d_dbg, counts)
#ifdef VO_EXPERIMENTS
    // vo_tuning.klt_pair (builds with -DVO_EXPERIMENTS): two keypoints per wave (k_klt_track2, slower); read per launch so that a test can compare both
    const int pair = c->tune.klt_pair;
    if (pair) {
      const int npair = (n + 1) / 2;
#define VO_KLT_LAUNCH2(WV) hipLaunchKernelGGL(k_klt_track2<WV>, dim3(npair, c->batch), dim3(64), 0, c->stream, A,            \
                       vo_slab<const float>(c, off_in), vo_slab<float>(c, off_out), vo_slab<uint8_t>(c, c->off_status),     \
                       vo_slab<float>(c, c->off_err), c->d_iters, counts)
      if (pair == 3) VO_KLT_LAUNCH2(3); else if (pair == 5) VO_KLT_LAUNCH2(5); else VO_KLT_LAUNCH2(4);
#undef VO_KLT_LAUNCH2
    } else
#endif
    if (waves <= 4) VO_KLT_LAUNCH(4); else if (waves == 5) VO_KLT_LAUNCH(5); else VO_KLT_LAUNCH(6);
#undef VO_KLT_LAUNCH
  }
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

// strided copies between [batch][n * elem] host arrays and the per-sequence slab rows
static hipError_t slab_h2d(vo_ctx* c, size_t off, const void* h, size_t row_bytes) {
  return hipMemcpy2DAsync(c->d_slab + off, c->slab_seq, h, row_bytes, row_bytes, c->batch, hipMemcpyHostToDevice, c->stream);
}
static hipError_t slab_d2h(vo_ctx* c, void* h, size_t off, size_t row_bytes) {
  return hipMemcpy2DAsync(h, row_bytes, c->d_slab + off, c->slab_seq, row_bytes, c->batch, hipMemcpyDeviceToHost, c->stream);
}

extern "C" int32_t vo_klt_track(vo_ctx* c, const float* p0, int32_t n, const vo_klt_params* prm,
                                float* p1, uint8_t* status, float* err, int32_t* iters) {
  if (!c) return VO_E_INVALID;
  vo_klt_params def;
  if (!prm) { vo_klt_default_params(&def); prm = &def; }
  VO_CHECK(c, n >= 0 && n <= c->max_pts, VO_E_CAPACITY, "n exceeds max_pts");
  if (n == 0) return VO_OK;
  VO_CHECK(c, p0 && p1 && status && err, VO_E_INVALID, "null buffer");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  const size_t off_in = vo_off_p(c), off_out = vo_off_p_next(c);
  VO_HIP(c, slab_h2d(c, off_in, p0, sizeof(float) * 2 * n));
  int32_t r = klt_launch(c, n, prm, off_in, off_out, nullptr);
  if (r != VO_OK) return r;
  VO_HIP(c, slab_d2h(c, p1, off_out, sizeof(float) * 2 * n));
  VO_HIP(c, slab_d2h(c, status, c->off_status, n));
  VO_HIP(c, slab_d2h(c, err, c->off_err, sizeof(float) * n));
  if (iters) {
    const size_t row = sizeof(int32_t) * (size_t)n * (prm->max_level + 1);
    VO_HIP(c, hipMemcpy2DAsync(iters, row, c->d_iters, sizeof(int32_t) * (size_t)c->max_pts * VO_MAX_LEVELS, row, c->batch,
                               hipMemcpyDeviceToHost, c->stream));
  }
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

extern "C" int32_t vo_points_upload(vo_ctx* c, const float* p, int32_t n) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, p && n >= 0 && n <= c->max_pts, VO_E_CAPACITY, "bad point set");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  if (n > 0) VO_HIP(c, slab_h2d(c, vo_off_p(c), p, sizeof(float) * 2 * n));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  c->n_resident = n;
  return VO_OK;
}

extern "C" int32_t vo_points_download(vo_ctx* c, float* p, uint8_t* status, float* err, int32_t* iters, int32_t n) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, n >= 0 && n <= c->n_resident, VO_E_INVALID, "n exceeds the resident point set");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  if (n > 0) {
    if (p) VO_HIP(c, slab_d2h(c, p, vo_off_p(c), sizeof(float) * 2 * n));
    if (status) VO_HIP(c, slab_d2h(c, status, c->off_status, n));
    if (err) VO_HIP(c, slab_d2h(c, err, c->off_err, sizeof(float) * n));
    if (iters && c->iters_stride > 0) {
      const size_t row = sizeof(int32_t) * (size_t)n * c->iters_stride;
      VO_HIP(c, hipMemcpy2DAsync(iters, row, c->d_iters, sizeof(int32_t) * (size_t)c->max_pts * VO_MAX_LEVELS, row, c->batch,
                                 hipMemcpyDeviceToHost, c->stream));
    }
  }
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

extern "C" int32_t vo_klt_track_resident(vo_ctx* c, int32_t n, const vo_klt_params* prm) {
  if (!c) return VO_E_INVALID;
  return vo_klt_track_resident_counts(c, n, prm, c->d_pt_counts);      // (non-null only while a vo_tracks_* table is seeded)
}

int32_t vo_klt_track_resident_counts(vo_ctx* c, int32_t n, const vo_klt_params* prm, const int32_t* d_counts) {
  vo_klt_params def;
  if (!prm) { vo_klt_default_params(&def); prm = &def; }
  VO_CHECK(c, n >= 0 && n <= c->n_resident, VO_E_INVALID, "n exceeds the resident point set");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = klt_launch(c, n, prm, vo_off_p(c), vo_off_p_next(c), d_counts);
  if (r != VO_OK) return r;
  c->p_parity ^= 1;   // tracked positions become the resident set
  return VO_OK;
}
