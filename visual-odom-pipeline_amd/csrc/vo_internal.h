// Internal definitions shared by the HIP translation units of libvo_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "vo_mi355x.h"

#define VO_PAD 32          // border (pixels) around every pyramid level, >= win + 1
#define VO_MAX_LEVELS 8
#define VO_MAX_WIN 31

struct vo_level {
  int w, h;        // interior size
  int pitch;       // padded row pitch in PIXELS (multiple of 64)
  int ph;          // padded rows = h + 2*VO_PAD
};

struct vo_frame {
  uint8_t* img[VO_MAX_LEVELS];   // padded, origin of the interior at (VO_PAD, VO_PAD)
  int16_t* der[VO_MAX_LEVELS];   // padded, interleaved (Ix, Iy), same pixel pitch; border = 0
};

#include <vector>
struct vo_prof {
  int mask = 0;                                                  // bit r = region r is timed
  std::vector<hipEvent_t> pool;                                  // all events ever created (reused)
  size_t used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs[VO_PROF_COUNT];
};

struct vo_st_ws;   // Shi-Tomasi workspace (vo_shi_tomasi.hip)
struct vo_ba_ws;   // bundle-adjustment workspace (vo_ba.hip)

struct vo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int width = 0, height = 0, max_pts = 0, max_level = 0, win = 0;
  int top = 0;                       // highest pyramid level index built
  vo_level lv[VO_MAX_LEVELS];
  vo_frame fr[2];
  int cur = 0;                       // index of the current frame in fr[]
  int n_pushed = 0;
  uint8_t* d_raw = nullptr;          // staging of one raw frame
  uint8_t* d_seq = nullptr;          // preloaded sequence (vo_seq_upload)
  int seq_n = 0;
  // tracked point set
  float* d_p0 = nullptr;             // n x 2
  float* d_p1 = nullptr;
  float* d_err = nullptr;
  uint8_t* d_status = nullptr;
  int32_t* d_iters = nullptr;        // n x (max_level + 1)
  int n_resident = 0;
  int iters_stride = 0;              // of the last KLT launch
  // DLT scratch
  float* d_uv0 = nullptr; float* d_uv1 = nullptr; float* d_X4 = nullptr;
  double* d_depth = nullptr; double* d_reproj = nullptr;
  int dlt_n = 0, dlt_stats = 0;
  float dlt_P0[12], dlt_P1[12];
  double dlt_K[9], dlt_H0[16], dlt_H1[16];
  vo_st_ws* st = nullptr;
  vo_ba_ws* ba = nullptr;
  vo_prof prof;
  unsigned long long* d_dbg = nullptr;   // 4 x 8 phase stamps (vo_debug_cycles)
  // result slab: every per-frame output of the front end lives in ONE device allocation (mirrored in pinned host
  // memory) so that a frame's results come back with a single D2H copy instead of ten.
  uint8_t* d_slab = nullptr;
  uint8_t* h_slab = nullptr;
  size_t slab_bytes = 0;
  size_t off_pa = 0, off_pb = 0, off_status = 0, off_err = 0, off_X4 = 0, off_depth = 0, off_reproj = 0,
         off_st_scalars = 0, off_st_out = 0;
  // per-frame step (vo_frame_step_resident) captured as hipGraphs, one per frame parity
  hipGraphExec_t step_graph[2] = {nullptr, nullptr};
  int step_sig[2][8];                    // launch signature the graph was captured for
  int32_t* d_frame_idx = nullptr;        // frame index consumed by the captured k_pad_level0
  int32_t* h_frame_idx = nullptr;        // pinned ring of frame indices (H2D source must outlive the copy)
  int frame_ring = 0;
  int use_graph = 0;                     // hipGraph replay is opt-in (vo_set_graph_mode): on ROCm 7.2 it is slower than plain launches
  std::string err;
};

inline int32_t vo_fail(vo_ctx* c, int32_t code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define VO_HIP(c, expr)                                                                   \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _b[512];                                                                       \
      snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
               hipGetErrorString(_e));                                                    \
      return vo_fail((c), VO_E_HIP, _b);                                                  \
    }                                                                                     \
  } while (0)

#define VO_CHECK(c, cond, code, msg)                                                      \
  do {                                                                                    \
    if (!(cond)) return vo_fail((c), (code), std::string(__func__) + ": " + (msg));       \
  } while (0)

// RAII bracket: records an event pair on the ctx stream around a region when profiling is on
struct vo_prof_scope {
  vo_ctx* c; int region; hipEvent_t e0 = nullptr, e1 = nullptr;
  vo_prof_scope(vo_ctx* c_, int region_);
  ~vo_prof_scope();
};

// phase stamp helper for the diagnostic cycle counters (thread 0 of a workgroup)
#define VO_STAMP(buf, idx) do { if ((buf) && threadIdx.x == 0) (buf)[idx] = __builtin_amdgcn_s_memtime(); } while (0)

static inline int vo_div_up(int a, int b) { return (a + b - 1) / b; }

// cross-unit internals used by the fused frame step (vo_step.hip)
int32_t vo_build_pyramid(vo_ctx* c, const uint8_t* d_raw_img, const int32_t* d_frame_idx);
int32_t vo_ba_enqueue_pub_copy(vo_ctx* c);
void vo_ba_unpack_pub(vo_ctx* c, double* poses_out, double* points_out, vo_ba_stats* stats);
bool vo_ba_ready(const vo_ctx* c);
bool vo_st_ready(const vo_ctx* c);
int vo_st_last_max_corners(const vo_ctx* c);
int32_t vo_st_prepare(vo_ctx* c);

// sub-workspace lifetime hooks
void vo_st_destroy(vo_ctx* c);
void vo_ba_destroy(vo_ctx* c);
