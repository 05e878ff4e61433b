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
  unsigned long long* d_dbg = nullptr;   // 3 x 8 phase stamps (vo_debug_cycles)
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

// sub-workspace lifetime hooks
void vo_st_destroy(vo_ctx* c);
void vo_ba_destroy(vo_ctx* c);
