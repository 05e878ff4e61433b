// Context lifetime + frame store: padded image pyramid and Scharr derivatives in HBM.
//
// Replaces the pyramid / derivative construction cv2.calcOpticalFlowPyrLK performs on every call
// (/root/reference/src/extractor/extractor.py:44-45,65-66 -> 8 pyramid builds per frame in the
// reference; here ONE per frame, kept resident and rotated prev <-> cur).
//
// HBM layout per frame, per level l (w_l x h_l interior):
//   img[l] : uint8, (h_l + 64) rows x pitch_l, interior origin at (32, 32), border = the
//            BORDER_REFLECT_101 extension of the interior (what buildOpticalFlowPyramid pads with),
//            so the tracker reads windows with plain unaligned dword loads and no index math.
//   der[l] : int16 x 2 interleaved (4 Ix, 4 Iy) -- Scharr times 4, see k_scharr_pyrdown --, same pixel pitch, border = 0 (BORDER_CONSTANT).
// pitch_l is a multiple of 64 pixels so rows start on 64 B (img) / 256 B (der) boundaries.
// All arithmetic here is integer and bit-exact against oracle/vo_oracle.c.
#include "vo_internal.h"

#include <stdlib.h>
#include <string.h>

#include <new>

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

// Level 0: raw (w x h, tight) -> padded image with reflect-101 border.  grid (x blocks, padded rows, batch).
// `raw` is the frame of sequence 0 (frame_idx == nullptr) or the base of the resident sequences, in which case the
// frame index is read from device memory (lets a captured graph be replayed for any frame).
// taps of the loader's bilateral pre-filter (passed by value)
struct bil_args {
  int maxk;
  signed char dx[49], dy[49];
  float sw[49];
};

// Level 0 of the frame store WITH the loader's pre-filter, cv2.bilateralFilter(img, d, sigmaColor, sigmaSpace) at
// /root/reference/src/loader/loader.py:16-20,86 (OpenCV 4.4 bilateralFilter_8u, scalar tap order; oracle:
// vo_oracle_bilateral): every padded pixel evaluates the filter at its reflected source position, so the filtered
// image never makes a round trip through HBM.  Float arithmetic in the oracle's order (the unit is built
// contract-off), IEEE division, round half to even.
__global__ void __launch_bounds__(256) k_pad_level0_bilateral(const uint8_t* __restrict__ raw, size_t raw_seq_stride,
                                                              const int32_t* __restrict__ frame_idx, int w, int h,
                                                              uint8_t* __restrict__ dst, size_t dst_seq_stride, int pitch,
                                                              int ph, bil_args A, const float* __restrict__ color_w) {
  __shared__ float s_cw[256];
  s_cw[threadIdx.x] = color_w[threadIdx.x];
  __syncthreads();
  const int X = blockIdx.x * blockDim.x + threadIdx.x;   // padded column
  const int Y = blockIdx.y;
  if (X >= w + 2 * VO_PAD || Y >= ph) return;
  raw += (size_t)blockIdx.z * raw_seq_stride;
  dst += (size_t)blockIdx.z * dst_seq_stride;
  if (frame_idx) raw += (size_t)(*frame_idx) * w * h;
  const int x = d_reflect101(X - VO_PAD, w), y = d_reflect101(Y - VO_PAD, h);
  const int val0 = raw[(size_t)y * w + x];
  float sum = 0.f, wsum = 0.f;
  for (int k = 0; k < A.maxk; k++) {
    const int val = raw[(size_t)d_reflect101(y + A.dy[k], h) * w + d_reflect101(x + A.dx[k], w)];
    const int ad = val > val0 ? val - val0 : val0 - val;
    const float wt = A.sw[k] * s_cw[ad];
    sum += (float)val * wt;
    wsum += wt;
  }
  dst[(size_t)Y * pitch + X] = (uint8_t)(int)__builtin_rintf(sum / wsum);
}

// Flat thread index -> (row, group) of a padded image whose groups [g_lo, g_hi) of every row take the interior path and the others the
// border path (reflect-101 index arithmetic per byte): the interior pairs of ALL rows come first, the border pairs behind them, so a wave
// runs ONE of the two paths.  (With the plain row-major index nearly every wave held a few border groups and executed both: k_pad_level0
// issued 229 vector instructions per wave for a 16-byte copy, the pyrDown role of k_scharr_pyrdown ~300 instead of ~60.)
// Returns true for an interior pair; Y >= rows: no work.
__device__ __forceinline__ bool vo_split_index(unsigned gid, int rows, int gpr, int g_lo, int g_hi, int& Y, int& G) {
  const int gi = g_hi - g_lo, gb = gpr - gi;
  const unsigned n_int = (unsigned)rows * (unsigned)gi;
  if (gid < n_int) {
    Y = (int)(gid / (unsigned)gi);
    G = g_lo + (int)(gid - (unsigned)Y * (unsigned)gi);
    return true;
  }
  gid -= n_int;
  Y = gb > 0 ? (int)(gid / (unsigned)gb) : rows;
  const int j = gb > 0 ? (int)(gid - (unsigned)Y * (unsigned)gb) : 0;
  G = j < g_lo ? j : g_hi + (j - g_lo);
  return false;
}

__global__ void __launch_bounds__(256) k_pad_level0(const uint8_t* __restrict__ raw, size_t raw_seq_stride,
                                                    const int32_t* __restrict__ frame_idx, int w, int h,
                                                    uint8_t* __restrict__ dst, size_t dst_seq_stride, int pitch, int ph, int remap) {
  // 16 consecutive padded columns per thread, one 16-byte store (pitch is a multiple of 64, VO_PAD of 16); the threads of the grid
  // run over (row, 16-byte group) pairs in one flat index so that every workgroup is full.  (4 columns per thread: 4.6 M threads
  // that each moved 8 bytes -- 21 us for 32 MB.)
  int blk, bseq;
  vo_xcd_assign(blockIdx.z * gridDim.x + blockIdx.x, gridDim.x, remap, blk, bseq);
  const int gpr = (w + 2 * VO_PAD + 15) / 16;                        // 16-byte groups per padded row
  const unsigned gid = (unsigned)blk * blockDim.x + threadIdx.x;
  // interior groups: X >= VO_PAD and X + 15 - VO_PAD < w
  const int g_lo = VO_PAD / 16, g_hi = max(g_lo, min(gpr, (w + VO_PAD - 16 >= 0 ? (w + VO_PAD - 16) / 16 + 1 : 0)));
  int Y, G;
  const bool interior = vo_split_index(gid, ph, gpr, g_lo, g_hi, Y, G);
  const int X = G * 16;
  if (Y >= ph) return;
  raw += (size_t)bseq * raw_seq_stride;
  dst += (size_t)bseq * dst_seq_stride;
  if (frame_idx) raw += (size_t)(*frame_idx) * w * h;
  const int y = d_reflect101(Y - VO_PAD, h);
  const uint8_t* row = raw + (size_t)y * w;
  uint32_t v[4];
  if (interior) {
    __builtin_memcpy(v, row + (X - VO_PAD), 16);         // interior: four unaligned dword loads
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v[q] = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) v[q] |= (uint32_t)row[d_reflect101(X + 4 * q + k - VO_PAD, w)] << (8 * k);
    }
  }
  uint4 pk; pk.x = v[0]; pk.y = v[1]; pk.z = v[2]; pk.w = v[3];
  *reinterpret_cast<uint4*>(dst + (size_t)Y * pitch + X) = pk;
}

// One launch per level l, two block roles (grid.y = batch):
//   blocks [0, nb_scharr)            : Scharr derivative of level l (interior only)
//   blocks [nb_scharr, gridDim.x)    : pyrDown level l -> l+1 including its reflect-101 border
// Both only READ level l, so they are independent inside the launch.
__device__ __forceinline__ int d_byte(uint32_t v, int k) { return (int)((v >> (8 * k)) & 0xFFu); }

__global__ void __launch_bounds__(256) k_scharr_pyrdown(const uint8_t* __restrict__ src, size_t src_seq_px, int w, int h, int pitch,
                                                        int16_t* __restrict__ der, int nb_scharr,
                                                        uint8_t* __restrict__ dst, size_t dst_seq_px, int dw, int dh, int dpitch, int remap) {
  int bx, bseq;
  vo_xcd_assign(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, remap, bx, bseq);
  src += (size_t)bseq * src_seq_px;
  if (bx < nb_scharr) {
    // ---- Scharr: Ix = [3 10 3]^T (x) [-1 0 1], Iy = [-1 0 1]^T (x) [3 10 3], un-normalised.
    //      A thread owns a strip of 4 pixels x SR rows: SR + 2 source rows of 3 aligned dwords each (12 row loads per 16 pixels
    //      instead of 36), one 16-byte store of 4 (4 Ix | 4 Iy << 16) per row.  The threads run over (row strip, 4-pixel group)
    //      pairs in one flat index, so every workgroup is full (a 1241-pixel row used to take two 1024-pixel workgroups, the second
    //      one fifth full).  The kernel is bound by the latency of many tiny workgroups, not by its stores (without the derivative
    //      stores the level-0 launch took 31 instead of 34 us): more work per thread is what helps ----
    constexpr int SR = 4;
    const int gpr = (w + 3) / 4;                                          // 4-pixel groups per row
    const unsigned gid = (unsigned)bx * 256u + threadIdx.x;
    const int ys = (int)(gid / (unsigned)gpr);                            // row strip
    const int x0 = (int)(gid - (unsigned)ys * (unsigned)gpr) * 4;
    const int y0 = ys * SR;
    if (y0 >= h) return;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    // columns x0 - 1 .. x0 + 4 of a source row as three pairs of 16-bit values (one byte gather each)
    auto load_row = [&](int yy, u16x2 (&a)[3]) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(src + (size_t)(yy + VO_PAD) * pitch + (x0 + VO_PAD));
      const uint32_t L = q[-1], M = q[0], R = q[1];
      a[0] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(M, L, 0x0c040c03u));   // (x0 - 1, x0)
      a[1] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, M, 0x0c020c01u));  // (x0 + 1, x0 + 2)
      a[2] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(R, M, 0x0c040c03u));   // (x0 + 3, x0 + 4)
    };
    u16x2 rows[SR + 2][3];
#pragma unroll
    for (int r = 0; r < SR + 2; r++) load_row(min(y0 - 1 + r, h), rows[r]);      // rows y0 - 1 .. y0 + SR (clamped into the padded image)
#pragma unroll
    for (int r = 0; r < SR; r++) {
      const int y = y0 + r;
      if (y >= h) break;
      // stored times 4 (|4 Scharr| <= 16 320 fits int16): k_klt_track's interpolation sum then carries its result in the upper
      // 16 bits (vo_klt.hip deriv1); vo_pyramid_read hands out the plain values
      s16x2 cs[3], dv[3];                                                 // 4 x [3 10 3]^T column sums, row differences
#pragma unroll
      for (int c = 0; c < 3; c++) {
        cs[c] = __builtin_bit_cast(s16x2, (u16x2)((rows[r][c] + rows[r + 2][c]) * (unsigned short)12 + rows[r + 1][c] * (unsigned short)40));
        dv[c] = __builtin_bit_cast(s16x2, (u16x2)(rows[r + 2][c] - rows[r][c]));
      }
      const uint32_t ix01 = __builtin_bit_cast(uint32_t, (s16x2)(cs[1] - cs[0]));      // Ix of pixels 0, 1: column k + 2 minus column k
      const uint32_t ix23 = __builtin_bit_cast(uint32_t, (s16x2)(cs[2] - cs[1]));
      const s16x2 w_lo = {12, 40}, w_hi = {40, 12}, w_0 = {12, 0}, w_1 = {0, 12};
      const int iy0 = __builtin_amdgcn_sdot2(dv[1], w_0, __builtin_amdgcn_sdot2(dv[0], w_lo, 0, false), false);   // 12 d0 + 40 d1 + 12 d2
      const int iy1 = __builtin_amdgcn_sdot2(dv[1], w_hi, __builtin_amdgcn_sdot2(dv[0], w_1, 0, false), false);   // 12 d1 + 40 d2 + 12 d3
      const int iy2 = __builtin_amdgcn_sdot2(dv[2], w_0, __builtin_amdgcn_sdot2(dv[1], w_lo, 0, false), false);
      const int iy3 = __builtin_amdgcn_sdot2(dv[2], w_hi, __builtin_amdgcn_sdot2(dv[1], w_1, 0, false), false);
      uint32_t out[4];
      out[0] = __builtin_amdgcn_perm((uint32_t)iy0, ix01, 0x05040100u);     // (Ix | Iy << 16)
      out[1] = __builtin_amdgcn_perm((uint32_t)iy1, ix01, 0x05040302u);
      out[2] = __builtin_amdgcn_perm((uint32_t)iy2, ix23, 0x05040100u);
      out[3] = __builtin_amdgcn_perm((uint32_t)iy3, ix23, 0x05040302u);
      if (x0 + 3 >= w) {                                                    // columns >= w belong to the zero (BORDER_CONSTANT) frame of the derivative image
#pragma unroll
        for (int k = 1; k < 4; k++) if (x0 + k >= w) out[k] = 0u;
      }
      uint4 pk; pk.x = out[0]; pk.y = out[1]; pk.z = out[2]; pk.w = out[3];
      const size_t o = (size_t)(y + VO_PAD) * pitch + (x0 + VO_PAD);       // multiple of 4
      *reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(der) + (size_t)bseq * src_seq_px + o) = pk;
    }
  } else {
    // ---- pyrDown: 5x5 [1 4 6 4 1]^2, (sum + 128) >> 8, output padded domain, 4 outputs per thread ----
    const int b = bx - nb_scharr;
    const int pw = dw + 2 * VO_PAD;
    const int gpr = (pw + 3) / 4;                                         // dwords per padded destination row
    const unsigned gid = (unsigned)b * 256u + threadIdx.x;
    // interior dwords: X0 >= VO_PAD and X0 + 3 - VO_PAD < dw
    const int g_lo = VO_PAD / 4, g_hi = max(g_lo, min(gpr, (dw + VO_PAD - 4 >= 0 ? (dw + VO_PAD - 4) / 4 + 1 : 0)));
    int Y, G;
    const bool interior = vo_split_index(gid, dh + 2 * VO_PAD, gpr, g_lo, g_hi, Y, G);
    const int X0 = G * 4;
    if (Y >= dh + 2 * VO_PAD) return;
    const int y = d_reflect101(Y - VO_PAD, dh);
    // source is padded by 32 with reflect-101, so 2x-2 .. 2x+2 never needs index reflection
    const uint8_t* prow = src + (size_t)(2 * y + VO_PAD) * pitch + VO_PAD;
    uint32_t res = 0;
    if (interior) {
      // interior: outputs x0 .. x0 + 3 read source columns 2 x0 - 2 .. 2 x0 + 8: 4 aligned dwords per source row
      const int x0 = X0 - VO_PAD;
      // the 25 taps of an output are five dwords-pairs: v_dot4_u32_u8 with the vertical weight folded into the byte weights
      // (<= 36) does extraction, multiplication and accumulation at once -- 10 instructions per output, no other arithmetic
      uint32_t acc[4] = {128u, 128u, 128u, 128u};
#pragma unroll
      for (int j = -2; j <= 2; j++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(prow + (ptrdiff_t)j * pitch + 2 * x0);
        const uint32_t d0 = q[-1], d1 = q[0], d2 = q[1], d3 = q[2];   // columns 2 x0 - 4 .. 2 x0 + 11
        const uint32_t wj = (j == 0) ? 6u : ((j == -1 || j == 1) ? 4u : 1u);
        acc[0] = __builtin_amdgcn_udot4(d1, wj * 0x00010406u, __builtin_amdgcn_udot4(d0, wj * 0x04010000u, acc[0], false), false);
        acc[1] = __builtin_amdgcn_udot4(d2, wj * 0x00000001u, __builtin_amdgcn_udot4(d1, wj * 0x04060401u, acc[1], false), false);
        acc[2] = __builtin_amdgcn_udot4(d2, wj * 0x00010406u, __builtin_amdgcn_udot4(d1, wj * 0x04010000u, acc[2], false), false);
        acc[3] = __builtin_amdgcn_udot4(d3, wj * 0x00000001u, __builtin_amdgcn_udot4(d2, wj * 0x04060401u, acc[3], false), false);
      }
      // byte 1 of every sum (255 * 256 + 128 < 2^16)
      res = __builtin_amdgcn_perm(__builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0501u), __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0501u), 0x05040100u);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int x = d_reflect101(X0 + k - VO_PAD, dw);
        const uint8_t* p = prow + 2 * x;
        int sum = 0;
#pragma unroll
        for (int j = -2; j <= 2; j++) {
          const uint8_t* r = p + (ptrdiff_t)j * pitch;
          const int row = r[-2] + 4 * r[-1] + 6 * r[0] + 4 * r[1] + r[2];
          const int wj = (j == 0) ? 6 : ((j == -1 || j == 1) ? 4 : 1);
          sum += wj * row;
        }
        res |= (uint32_t)((sum + 128) >> 8) << (8 * k);
      }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)bseq * dst_seq_px + (size_t)Y * dpitch + X0) = res;
  }
}

// ------------------------------------------------------------------------------------------------
// in-stream timing
// ------------------------------------------------------------------------------------------------
static hipEvent_t prof_event(vo_ctx* c) {
  vo_prof& p = c->prof;
  if (p.used == p.pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    p.pool.push_back(e);
  }
  return p.pool[p.used++];
}

vo_prof_scope::vo_prof_scope(vo_ctx* c_, int region_) : c(c_), region(region_) {
  if (!((c->prof.mask >> region) & 1)) return;
  e0 = prof_event(c); e1 = prof_event(c);
  if (e0) (void)hipEventRecord(e0, c->stream);
}

vo_prof_scope::~vo_prof_scope() {
  if (!e0 || !e1) return;
  (void)hipEventRecord(e1, c->stream);
  c->prof.pairs[region].push_back(std::make_pair(e0, e1));
}

extern "C" int32_t vo_profile_enable(vo_ctx* c, int32_t region_mask) {
  if (!c) return VO_E_INVALID;
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  c->prof.mask = region_mask;
  c->prof.used = 0;
  for (int r = 0; r < VO_PROF_COUNT; r++) c->prof.pairs[r].clear();
  return VO_OK;
}

extern "C" int32_t vo_debug_cycles(vo_ctx* c, int32_t which, int64_t* out8) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, which >= 0 && which < 4 && out8, VO_E_INVALID, "bad selector");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  unsigned long long h[8];
  VO_HIP(c, hipMemcpy(h, c->d_dbg + 8 * which, sizeof(h), hipMemcpyDeviceToHost));
  out8[0] = 0;
  for (int i = 1; i < 8; i++) out8[i] = (h[i] && h[i - 1] && !(which == 0 && i >= 5)) ? (int64_t)(h[i] - h[i - 1]) : 0;
  for (int i = 1; i < 8; i++) out8[0] += out8[i];
  if (which == 0) { out8[7] = (int64_t)h[6]; out8[5] = (h[5] && h[2]) ? (int64_t)(h[5] - h[2]) : 0; }   // rounds; cycles of round 0
  return VO_OK;
}

extern "C" int32_t vo_profile_read(vo_ctx* c, int32_t region, double* total_ms, int32_t* count) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, region >= 0 && region < VO_PROF_COUNT && total_ms && count, VO_E_INVALID, "bad region");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  double t = 0;
  for (auto& pr : c->prof.pairs[region]) {
    float ms = 0;
    VO_HIP(c, hipEventElapsedTime(&ms, pr.first, pr.second));
    t += ms;
  }
  *total_ms = t; *count = (int32_t)c->prof.pairs[region].size();
  return VO_OK;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" int32_t vo_abi_version(void) { return VO_ABI_VERSION; }

// the ONE environment switch of the library that is not a debug trace: how a host thread waits (a property of the process's thread / core budget,
// not of a context's kernels)
bool vo_blocking_sync() {
  static const bool on = getenv("VO_BLOCKING_SYNC") && atoi(getenv("VO_BLOCKING_SYNC")) != 0;
  return on;
}

extern "C" int32_t vo_device_count(int32_t* n) {
  if (!n) return VO_E_INVALID;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) { *n = 0; return VO_E_HIP; }
  *n = c;
  return VO_OK;
}

static thread_local std::string g_create_err;

extern "C" const char* vo_last_error(const vo_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

extern "C" int32_t vo_ctx_destroy(vo_ctx* c) {
  if (!c) return VO_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);   // side branch of the frame step (an error path may have left it unjoined)
  if (c->stream_h2d) (void)hipStreamSynchronize(c->stream_h2d);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  (void)vo_comm_destroy(c);
  vo_pipe_destroy(c);
  vo_trk_destroy(c);
  vo_pnp_destroy(c);
  vo_ess_destroy(c);
  vo_match_destroy(c);
  vo_sift_destroy(c);
  vo_st_destroy(c);
  vo_ba_destroy(c);
  for (int f = 0; f < 2; f++)
    for (int l = 0; l < VO_MAX_LEVELS; l++) {
      if (c->fr[f].img[l]) (void)hipFree(c->fr[f].img[l]);
      if (c->fr[f].der[l]) (void)hipFree(c->fr[f].der[l]);
    }
  for (auto& g : c->step_graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
  void* bufs[] = {c->d_bil_cw, c->d_dbg, c->d_raw, c->d_seq, c->d_iters, c->d_uv0, c->d_uv1, c->d_slab, c->d_frame_idx, c->d_dlt_cam};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (c->h_slab) (void)hipHostFree(c->h_slab);
  if (c->h_frame_idx) (void)hipHostFree(c->h_frame_idx);
  if (c->h_raw) (void)hipHostFree(c->h_raw);
  if (c->ev_raw) (void)hipEventDestroy(c->ev_raw);
  for (hipEvent_t e : c->prof.pool) (void)hipEventDestroy(e);
  for (int k = 0; k < 2; k++) if (c->ev_step[k]) (void)hipEventDestroy(c->ev_step[k]);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  for (int k = 0; k < 2; k++) {
    if (c->d_host_raw[k]) (void)hipFree(c->d_host_raw[k]);
    if (c->ev_h2d[k]) (void)hipEventDestroy(c->ev_h2d[k]);
    if (c->ev_raw_free[k]) (void)hipEventDestroy(c->ev_raw_free[k]);
  }
  if (c->stream_h2d) (void)hipStreamDestroy(c->stream_h2d);
  if (c->h_ptr_tab) (void)hipHostFree(c->h_ptr_tab);
  for (int k = 0; k < 2; k++) if (c->ev_ba_wide[k]) (void)hipEventDestroy(c->ev_ba_wide[k]);
  for (int k = 0; k < 2; k++) { if (c->ev_ba[k]) (void)hipEventDestroy(c->ev_ba[k]); if (c->ev_pub[k]) (void)hipEventDestroy(c->ev_pub[k]); if (c->ev_copy1[k]) (void)hipEventDestroy(c->ev_copy1[k]); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return VO_OK;
}

// a non-blocking stream; reserve_cus > 0: its queue is confined to the first 256 - reserve_cus bits of the compute-unit mask (the bits are dealt
// round-robin to the 8 XCDs, so a multiple of 8 takes the same number of CUs from each)
hipError_t vo_stream_create(hipStream_t* st, int reserve_cus) {
  int dev = 0, n_cu = 0;
  if (reserve_cus > 0 && (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)) n_cu = 0;
  if (reserve_cus <= 0 || n_cu <= 2 * reserve_cus || n_cu > 1024) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  uint32_t mask[32];
  for (int k = 0; k < 32; k++) mask[k] = 0;
  for (int b = 0; b < n_cu - reserve_cus; b++) mask[b >> 5] |= 1u << (b & 31);
  return hipExtStreamCreateWithCUMask(st, (uint32_t)((n_cu + 31) / 32), mask);
}

extern "C" int32_t vo_ctx_create(int32_t device, int32_t width, int32_t height, int32_t max_pts,
                                 int32_t max_level, int32_t win, vo_ctx** out) {
  return vo_ctx_create_batched(device, width, height, max_pts, max_level, win, 1, out);
}

extern "C" int32_t vo_ctx_create_batched(int32_t device, int32_t width, int32_t height, int32_t max_pts,
                                         int32_t max_level, int32_t win, int32_t batch, vo_ctx** out) {
  if (!out) return VO_E_INVALID;
  *out = nullptr;
  if (width < 8 || height < 8 || max_pts < 1 || max_level < 0 || max_level >= VO_MAX_LEVELS ||
      win < 3 || win > VO_MAX_WIN || (win & 1) == 0 || batch < 1 || batch > 1024) {
    g_create_err = "vo_ctx_create: invalid argument";
    return VO_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_err = "vo_ctx_create: no HIP device (this library has no CPU fallback)";
    return VO_E_HIP;
  }
  if (device < 0 || device >= ndev) { g_create_err = "vo_ctx_create: bad device index"; return VO_E_INVALID; }
  vo_ctx* c = new (std::nothrow) vo_ctx();
  if (!c) return VO_E_NOMEM;
  for (int f = 0; f < 2; f++)
    for (int l = 0; l < VO_MAX_LEVELS; l++) { c->fr[f].img[l] = nullptr; c->fr[f].der[l] = nullptr; }
  c->device = device; c->width = width; c->height = height; c->max_pts = max_pts; c->batch = batch;
  c->max_level = max_level; c->win = win;
  auto fail = [&](int32_t code) { g_create_err = c->err; vo_ctx_destroy(c); return code; };
#define CR(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { c->err = std::string(#expr) + " -> " + hipGetErrorString(_e); return fail(VO_E_HIP); } } while (0)
  CR(hipSetDevice(device));
  CR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CR(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));

  CR(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  CR(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  {
    // VO_BLOCKING_SYNC=1: vo_frame_fetch sleeps in the driver instead of spinning on the step's event (for more host threads than cores)
    const unsigned fl = hipEventDisableTiming | (vo_blocking_sync() ? hipEventBlockingSync : 0u);
    CR(hipEventCreateWithFlags(&c->ev_step[0], fl));
    CR(hipEventCreateWithFlags(&c->ev_step[1], fl));
  }
  // pyramid geometry (buildOpticalFlowPyramid truncation rule)
  int w = width, h = height, top = 0;
  for (int l = 0; l <= max_level; l++) {
    if (l > 0) {
      const int nw = (w + 1) / 2, nh = (h + 1) / 2;
      if (nw <= win || nh <= win) break;
      w = nw; h = nh;
    }
    c->lv[l].w = w; c->lv[l].h = h;
    c->lv[l].pitch = ((w + 2 * VO_PAD + 63) / 64) * 64;
    c->lv[l].ph = h + 2 * VO_PAD;
    c->lvl_px[l] = (size_t)c->lv[l].pitch * c->lv[l].ph;
    top = l;
  }
  c->top = top;
  for (int f = 0; f < 2; f++)
    for (int l = 0; l <= top; l++) {
      const size_t px = c->lvl_px[l] * (size_t)batch;
      CR(hipMalloc((void**)&c->fr[f].img[l], px));
      CR(hipMalloc((void**)&c->fr[f].der[l], px * 4));
      CR(hipMemsetAsync(c->fr[f].img[l], 0, px, c->stream));
      CR(hipMemsetAsync(c->fr[f].der[l], 0, px * 4, c->stream));   // border stays 0 forever
    }
  CR(hipMalloc((void**)&c->d_raw, (size_t)width * height * batch));
  CR(hipMalloc((void**)&c->d_dbg, sizeof(unsigned long long) * 32));
  CR(hipMemsetAsync(c->d_dbg, 0, sizeof(unsigned long long) * 32, c->stream));
  {
    // result slab layout (all offsets 256-byte aligned)
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t M = (size_t)max_pts;
    c->off_pa = take(8 * M); c->off_pb = take(8 * M); c->off_status = take(M); c->off_err = take(4 * M);
    c->off_X4 = take(16 * M); c->off_depth = take(8 * M); c->off_reproj = take(8 * M);
    c->off_st_scalars = take(64); c->off_st_out = take(8 * 4096);
    c->slab_seq = off;
    c->slab_bytes = off * (size_t)batch;
    CR(hipMalloc((void**)&c->d_slab, c->slab_bytes));
    CR(hipMemsetAsync(c->d_slab, 0, c->slab_bytes, c->stream));
    CR(hipHostMalloc((void**)&c->h_slab, 2 * c->slab_bytes, hipHostMallocDefault));
  }
  CR(hipMalloc((void**)&c->d_iters, sizeof(int32_t) * (size_t)max_pts * VO_MAX_LEVELS * batch));
  CR(hipMalloc((void**)&c->d_uv0, sizeof(float) * 2 * (size_t)max_pts * batch));
  CR(hipMalloc((void**)&c->d_uv1, sizeof(float) * 2 * (size_t)max_pts * batch));
  CR(hipMalloc((void**)&c->d_frame_idx, sizeof(int32_t)));
  CR(hipHostMalloc((void**)&c->h_frame_idx, sizeof(int32_t) * 64, hipHostMallocDefault));
  CR(hipStreamSynchronize(c->stream));
#undef CR
  *out = c;
  return VO_OK;
}

extern "C" int32_t vo_sync(vo_ctx* c) {
  if (!c) return VO_E_INVALID;
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (c->stream2) VO_HIP(c, hipStreamSynchronize(c->stream2));     // branches of a frame step that has not been fetched
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
  if (c->stream_h2d) VO_HIP(c, hipStreamSynchronize(c->stream_h2d));
  return VO_OK;
}

// ------------------------------------------------------------------------------------------------
// frames
// ------------------------------------------------------------------------------------------------
int32_t vo_build_pyramid(vo_ctx* c, const uint8_t* d_raw_img, size_t raw_seq_stride, const int32_t* d_frame_idx) {
  vo_prof_scope prof(c, VO_PROF_FRAME);
  c->cur ^= 1;
  vo_frame& F = c->fr[c->cur];
  const int B = c->batch;
  const int remap = (!c->tune.xcd_remap_off && B % 8 == 0) ? 1 : 0;       // every sequence's frame chain on one XCD (vo_xcd_assign)
  {
    const vo_level& L = c->lv[0];
    dim3 grid(vo_div_up(((L.w + 2 * VO_PAD + 15) / 16) * L.ph, 256), 1, B);   // 16 columns per thread, flat (row, group) index
    dim3 grid1(vo_div_up(L.w + 2 * VO_PAD, 256), L.ph, B);    // bilateral variant: 1 column per thread
    if (c->bil_maxk > 0) {
      bil_args A;
      A.maxk = c->bil_maxk;
      for (int k = 0; k < 49; k++) { A.dx[k] = c->bil_dx[k]; A.dy[k] = c->bil_dy[k]; A.sw[k] = c->bil_sw[k]; }
      hipLaunchKernelGGL(k_pad_level0_bilateral, grid1, dim3(256), 0, c->stream, d_raw_img, raw_seq_stride, d_frame_idx, L.w,
                         L.h, F.img[0], c->lvl_px[0], L.pitch, L.ph, A, c->d_bil_cw);
    } else {
      hipLaunchKernelGGL(k_pad_level0, grid, dim3(256), 0, c->stream, d_raw_img, raw_seq_stride, d_frame_idx, L.w, L.h,
                         F.img[0], c->lvl_px[0], L.pitch, L.ph, remap);
    }
  }
  for (int l = 0; l <= c->top; l++) {
    const vo_level& L = c->lv[l];
    const int nb_scharr = vo_div_up(vo_div_up(L.w, 4) * vo_div_up(L.h, 4), 256);      // threads = (4-pixel groups per row) x (4-row strips)
    int nb_down = 0;
    uint8_t* dst = nullptr; int dw = 0, dh = 0, dpitch = 0; size_t dpx = 0;
    if (l < c->top) {
      const vo_level& D = c->lv[l + 1];
      dst = F.img[l + 1]; dw = D.w; dh = D.h; dpitch = D.pitch; dpx = c->lvl_px[l + 1];
      nb_down = vo_div_up(vo_div_up(D.w + 2 * VO_PAD, 4) * D.ph, 256);
    }
    hipLaunchKernelGGL(k_scharr_pyrdown, dim3(nb_scharr + nb_down, B), dim3(256), 0, c->stream,
                       F.img[l], c->lvl_px[l], L.w, L.h, L.pitch, F.der[l], nb_scharr, dst, dpx, dw, dh, dpitch, remap);
  }
  VO_HIP(c, hipGetLastError());
  c->n_pushed++;
  return VO_OK;
}

// img: `batch` images, sequence b at img + b * seq_stride bytes (seq_stride = stride * height when 0)
extern "C" int32_t vo_frame_push(vo_ctx* c, const uint8_t* img, int32_t stride) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, img != nullptr && stride >= c->width, VO_E_INVALID, "bad image / stride");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  // the batch's images are contiguous: [batch][height] rows of `stride` bytes.  They go through a pinned staging buffer: a 2-D copy out of
  // pageable memory cost 2.5 ms per 1241 x 376 frame on this runtime (row by row), the memcpy + one linear copy 0.1 ms -- and the call no
  // longer waits for the pyramid (the caller's buffer is free as soon as the memcpy is done)
  const size_t rows = (size_t)c->height * c->batch, bytes = rows * c->width;
  if (!c->h_raw) {
    VO_HIP(c, hipHostMalloc((void**)&c->h_raw, bytes, hipHostMallocDefault));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_raw, hipEventDisableTiming));
  } else {
    VO_HIP(c, hipEventSynchronize(c->ev_raw));    // the previous frame has left the staging buffer
  }
  if (stride == c->width) memcpy(c->h_raw, img, bytes);
  else for (size_t y = 0; y < rows; y++) memcpy(c->h_raw + y * c->width, img + y * (size_t)stride, c->width);
  VO_HIP(c, hipMemcpyAsync(c->d_raw, c->h_raw, bytes, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipEventRecord(c->ev_raw, c->stream));
  return vo_build_pyramid(c, c->d_raw, (size_t)c->width * c->height, nullptr);
}

// frames: [batch][n_frames][height][width] uint8
extern "C" int32_t vo_seq_upload(vo_ctx* c, const uint8_t* frames, int32_t n_frames) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, frames != nullptr && n_frames > 0, VO_E_INVALID, "bad sequence");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (c->stream2) VO_HIP(c, hipStreamSynchronize(c->stream2));
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
  if (c->d_seq) { VO_HIP(c, hipFree(c->d_seq)); c->d_seq = nullptr; c->seq_n = 0; }
  const size_t bytes = (size_t)c->width * c->height * n_frames * c->batch;
  VO_HIP(c, hipMalloc((void**)&c->d_seq, bytes));
  VO_HIP(c, hipMemcpy(c->d_seq, frames, bytes, hipMemcpyHostToDevice));
  c->seq_n = n_frames;
  return VO_OK;
}

extern "C" int32_t vo_frame_push_resident(vo_ctx* c, int32_t idx) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->d_seq != nullptr && idx >= 0 && idx < c->seq_n, VO_E_STATE, "no resident sequence / bad index");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  const size_t fr = (size_t)c->width * c->height;
  return vo_build_pyramid(c, c->d_seq + (size_t)idx * fr, fr * c->seq_n, nullptr);
}

extern "C" int32_t vo_pyramid_level_size(vo_ctx* c, int32_t level, int32_t* w, int32_t* h) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, level >= 0 && level <= c->top && w && h, VO_E_INVALID, "bad level");
  *w = c->lv[level].w; *h = c->lv[level].h;
  return VO_OK;
}

extern "C" int32_t vo_pyramid_read(vo_ctx* c, int32_t which, int32_t level, uint8_t* img_out, int16_t* deriv_out) {
  return vo_pyramid_read_seq(c, 0, which, level, img_out, deriv_out);
}

extern "C" int32_t vo_pyramid_read_seq(vo_ctx* c, int32_t seq, int32_t which, int32_t level, uint8_t* img_out, int16_t* deriv_out) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, level >= 0 && level <= c->top && (which == 0 || which == 1) && seq >= 0 && seq < c->batch, VO_E_INVALID, "bad level / which / seq");
  VO_CHECK(c, c->n_pushed >= (which == 0 ? 2 : 1), VO_E_STATE, "frame not pushed yet");
  VO_HIP(c, hipSetDevice(c->device));
  const vo_frame& F = c->fr[which == 1 ? c->cur : (c->cur ^ 1)];
  const vo_level& L = c->lv[level];
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (img_out)
    VO_HIP(c, hipMemcpy2D(img_out, L.w, F.img[level] + (size_t)seq * c->lvl_px[level] + (size_t)VO_PAD * L.pitch + VO_PAD, L.pitch, L.w, L.h,
                          hipMemcpyDeviceToHost));
  if (deriv_out)
    VO_HIP(c, hipMemcpy2D(deriv_out, (size_t)L.w * 4, F.der[level] + ((size_t)seq * c->lvl_px[level] + (size_t)VO_PAD * L.pitch + VO_PAD) * 2,
                          (size_t)L.pitch * 4, (size_t)L.w * 4, L.h, hipMemcpyDeviceToHost));
  if (deriv_out)
    for (size_t i = 0, n = (size_t)L.w * L.h * 2; i < n; i++) deriv_out[i] = (int16_t)(deriv_out[i] / 4);   // device keeps 4 x Scharr (exact)
  return VO_OK;
}

// Loader pre-filter of every frame entering the frame store: cv2.bilateralFilter(img, d, sigmaColor, sigmaSpace),
// /root/reference/src/loader/loader.py:16-20 (d 5, sigmas 1.5), :86.  d = 0 switches it off (default).  The tap list and
// the weight tables are built exactly like OpenCV 4.4 bilateralFilter_8u does (double exp rounded to float).
extern "C" int32_t vo_set_prefilter(vo_ctx* c, int32_t d, double sigma_color, double sigma_space) {
  if (!c) return VO_E_INVALID;
  if (d == 0) { c->bil_maxk = 0; return VO_OK; }
  if (sigma_color <= 0) sigma_color = 1;
  if (sigma_space <= 0) sigma_space = 1;
  int radius = d < 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
  if (radius < 1) radius = 1;
  VO_CHECK(c, radius <= 3, VO_E_CAPACITY, "bilateral pre-filter: diameter up to 7");
  VO_HIP(c, hipSetDevice(c->device));
  const double gc = -0.5 / (sigma_color * sigma_color), gs = -0.5 / (sigma_space * sigma_space);
  float cw[256];
  for (int i = 0; i < 256; i++) cw[i] = (float)exp((double)i * i * gc);
  int maxk = 0;
  for (int i = -radius; i <= radius; i++)
    for (int j = -radius; j <= radius; j++) {
      const double r = sqrt((double)i * i + (double)j * j);
      if (r > radius) continue;
      c->bil_sw[maxk] = (float)exp(r * r * gs);
      c->bil_dx[maxk] = (signed char)j; c->bil_dy[maxk] = (signed char)i;
      maxk++;
    }
  if (!c->d_bil_cw) VO_HIP(c, hipMalloc((void**)&c->d_bil_cw, sizeof(cw)));
  VO_HIP(c, hipMemcpyAsync(c->d_bil_cw, cw, sizeof(cw), hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  c->bil_maxk = maxk;
  return VO_OK;
}
