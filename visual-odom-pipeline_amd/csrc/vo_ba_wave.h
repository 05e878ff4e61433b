// Bundle adjustment, windows of <= 10 slots: the WAVE-PRIVATE form of k_ba_build / k_ba_update (round 5).  Included by vo_ba.hip.
//
// Replaces (same reference lines as vo_ba.hip: bundle_adjuster.py:18-65, :189-194) the lane-per-observation kernels for the shapes the
// path is quoted on (BASELINE configs[2]: window 10; the reference's own window: 4).  What the old form paid for at a batch of 256
// problems (profiles/r04_kernel_stats_default.csv: 36 % of the kernel time of a step):
//   * a landmark owned 16 lanes, 10 of them busy at W = 10, and every lane repeated the landmark's 3 x 3 factorisation and took part in
//     nine 16-lane all-reduces;
//   * a 256-lane workgroup shared ONE panel (16 landmarks), so every landmark chunk cost four workgroup barriers, and the Gram tiles of a
//     chunk (20 KB) were stored -- and read back and stored again by the next chunk of the walk: 300 MB of HBM writes per launch;
//   * every MFMA fetched its own two operands from LDS (2 or 3 tiles per wave).
// Here a WAVE is the unit: it walks landmark chunks of 64 / LPP landmarks (LPP = 8 lanes per landmark, a lane serves SPL = ceil(W / 8)
// window slots), owns a private panel of 3 * 64 / LPP rows in LDS (no workgroup barrier inside the walk: the LDS pipe serves one wave's
// instructions in order), keeps ALL upper Gram tiles (10 at W = 10) in its accumulator registers over the whole walk -- one operand fetch
// per 16-column block and k-step feeds every tile that uses it --, keeps the camera sums half reduced in registers (two swap stages per
// chunk, the last stage once per walk), and only at the end the four waves of a workgroup add their tiles and camera sums through LDS in
// a fixed order and store ONE partial set.  A problem of the batch is served by G workgroups (2 at a batch of 256: 8 waves, 31 chunks
// each), so k_ba_solve sums 2 partial sets itself and k_ba_reduce is not launched.
//
// Linearisation: p = (K R) X + K t with K R, K t staged per camera; d(u, v)/dX = ((K R)_{0,1} - (u, v) (K R)_2) / p_2 directly; the
// rotation block as (X x Jl_k)^T Jr (= -Jl_k [X]x Jr); 1 / p_2 and the Huber weight from v_rcp_f64 / v_rsq_f64 + two Newton steps.
// Same algorithm as oracle/ba_oracle.py (checked step by step in tests/test_gpu_ba.py); the summation order is this file's.
#pragma once
#pragma clang fp contract(fast)

#define BA2_CAM 21           // per window slot in LDS: K R (9), K t (3), Jr (9)
#define BA2_TARGET_WAVES 2048   // waves a batched launch aims for: 2 per SIMD (the register budget of the build kernel)

// reciprocal to double precision without the division's scaling / fix-up sequence (the operand is a depth times the focal scale: far
// from the denormals and from overflow)
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  return y;
}

// ---- lane maps ----
// LPP = 8: landmark = lane / 8 (8 per wave), slot group q = lane % 8.
// LPP = 5 (windows of 9 and 10 slots: a lane serves slots q and q + 5, every lane busy twice -- with 8 lanes per landmark the second slot
//          of six lanes in eight is empty): a 16-lane row holds three landmarks (lanes 0-4, 5-9, 10-14), lane 15 of every row is idle;
//          12 landmarks per wave.  The sums of a landmark form in its LAST lane (q = 4, the "leader"); what the other lanes need comes
//          back through ds_bpermute.
template <int LPP> struct ba2_map;
template <> struct ba2_map<8> {
  static constexpr int LPC = 8, LEAD = 0;
  int pl, q; bool idle;
  __device__ __forceinline__ ba2_map(int lane) : pl(lane >> 3), q(lane & 7), idle(false) {}
};
template <> struct ba2_map<4> {      // windows of <= 4 slots (the reference's own window): 16 landmarks per wave
  static constexpr int LPC = 16, LEAD = 0;
  int pl, q; bool idle;
  __device__ __forceinline__ ba2_map(int lane) : pl(lane >> 2), q(lane & 3), idle(false) {}
};
template <> struct ba2_map<5> {
  static constexpr int LPC = 12, LEAD = 4;
  int pl, q; bool idle;
  __device__ __forceinline__ ba2_map(int lane) {
    const int c = lane & 15, g = (c * 13) >> 6;      // c / 5 for c < 16
    pl = 3 * (lane >> 4) + (g < 3 ? g : 2); q = c - 5 * g; idle = c == 15;
    if (idle) q = 0;
  }
};

// sum over the lanes of a landmark: everywhere (LPP = 8) / in the leader lane (LPP = 5: row_shr 1, 2, 4 -- the leader's five-term window
// ends inside its own group; the other lanes hold partial or foreign sums nobody reads)
template <int LPP>
__device__ __forceinline__ double ba2_group_sum(double v) {
  if (LPP == 8) return group8_allreduce(v);
  if (LPP == 4) { v += dpp_f64<0xB1>(v); return v + dpp_f64<0x4E>(v); }     // quad_perm [1, 0, 3, 2], [2, 3, 0, 1]
  double s = v + dpp_f64<0x111>(v);     // row_shr:1   v[l] + v[l-1]
  s += dpp_f64<0x112>(s);               // row_shr:2   ... + v[l-2] + v[l-3]
  return s + dpp_f64<0x114>(v);         // row_shr:4   ... + v[l-4]
}
// the leader's value in every lane of its group (LPP = 5); src4 = 4 * (lane index of the group's leader)
template <int LPP>
__device__ __forceinline__ double ba2_from_leader(double v, int src4) {
  if (LPP != 5) return v;
  const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// the landmark positions of the lane index that the per-chunk swap stages (lane bits 4 and 5) leave: bit 3 (LPP = 8); the three groups of a
// row (LPP = 5: row_shl 5 and 10, valid in lanes 0..4 of a row)
template <int LPP>
__device__ __forceinline__ double ba2_finish_landmark_sum(double v) {
  if (LPP == 8) return v + dpp_f64<0x128>(v);                 // row_ror:8
  if (LPP == 4) { v += dpp_f64<0x128>(v); return v + dpp_f64<0x124>(v); }   // row_ror:8, row_ror:4
  return v + dpp_f64<0x105>(v) + dpp_f64<0x10A>(v);           // row_shl:5, row_shl:10
}

// K R, K t, Jr of camera i into LDS from the 21 doubles [R | t | Jr] of d_camera
__device__ __forceinline__ void ba2_stage_camera(const double* __restrict__ K, const double* c, double* dst) {
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int k = 0; k < 3; k++) dst[3 * r + k] = K[3 * r] * c[k] + K[3 * r + 1] * c[3 + k] + K[3 * r + 2] * c[6 + k];
    dst[9 + r] = K[3 * r] * c[9] + K[3 * r + 1] * c[10] + K[3 * r + 2] * c[11];
  }
#pragma unroll
  for (int k = 0; k < 9; k++) dst[12 + k] = c[12 + k];
}

// ---- which problem of the batch, and which part of it, a workgroup serves ----
// The grid is one-dimensional: G0 workgroups per problem of the batch.  In the first iteration workgroup idx serves part idx % G0 of problem
// idx / G0.  Later the problems that finished in an earlier iteration need nobody, and what matters for the launch is how long the slowest
// still-running problem takes (at a batch of 256 a tail group with ten running problems took as long as a full one: 2 workgroups walking
// 31 chunks each): the workgroups are dealt out again, G = min(Gcap, grid / running problems) to every running problem in ascending order.
// Every kernel of the iteration derives the same assignment from the `done` flags of the previous iteration's states (written before this
// launch group started, not touched by it); G of an iteration is also left in P.gdyn[it & 1][problem] for the readers of its partial sets
// (k_ba_solve, k_ba_reduce) and of its step statistics (the next decision).  The order of a problem's partial sums depends on G, and so --
// like on the batch size before -- on how many problems of the batch are still running.
struct ba2_work { int prob, part, G; };
template <bool CARRY>
__device__ __forceinline__ ba2_work ba2_select_work(const ba_ptrs& Pall, int it, int G0, int Gcap) {
  __shared__ unsigned long long s_mask[16];   // running problems, 64 per word (batch <= 1024)
  __shared__ int s_out[3];
  const int B = Pall.batch, idx = blockIdx.x, tid = threadIdx.x;
  ba2_work w;
  if (it == 0) { w.G = G0; w.prob = idx / G0; w.part = idx - w.prob * G0; return w; }
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + tid;
    bool run = false;
    if (b < B) {
      ba_state* sp = Pall.state + 2 * (size_t)b;
      run = !sp[(it - 1) & 1].done;
      if (CARRY && idx == 0 && !run) sp[it & 1] = sp[(it - 1) & 1];     // a finished problem only carries its state forward
    }
    const unsigned long long m = __ballot(run);
    if ((tid & 63) == 0 && b0 + tid < B) s_mask[(b0 + tid) >> 6] = m;
  }
  __syncthreads();
  if (tid == 0) {
    const int nw = (B + 63) >> 6;
    int n = 0;
    for (int k = 0; k < nw; k++) n += __popcll(s_mask[k]);
    int G = n > 0 ? (int)gridDim.x / n : G0;
    if (G > Gcap) G = Gcap;
    if (G < G0) G = G0;
    int k = idx / G, prob = -1;
    const int part = idx - k * G;
    if (k < n) {
      for (int q = 0; q < nw; q++) {
        unsigned long long m = s_mask[q];
        const int c = __popcll(m);
        if (k >= c) { k -= c; continue; }
        for (; k > 0; k--) m &= m - 1;          // drop the k lowest set bits
        prob = 64 * q + (int)__ffsll((long long)m) - 1;
        break;
      }
    }
    s_out[0] = prob; s_out[1] = part; s_out[2] = G;
  }
  __syncthreads();
  w.prob = s_out[0]; w.part = s_out[1]; w.G = s_out[2];
  return w;
}

// a value every lane of the workgroup holds alike, as the compiler cannot know (it came out of LDS): moved to scalar registers, so that the pointers
// and constants derived from it (the problem's base addresses, K, lambda) stop occupying vector registers -- 30 of the build kernel's 256
__device__ __forceinline__ int ba2_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double ba2_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

struct ba2_lin {
  double e0, e1, w, rho;
  double Jl[2][3];
  double A[2][3];      // d(u, v) / d(camera-frame point) = the translation block of Jp
};

// residual, weight, landmark block and translation block of one observation; an unobserved slot (uo = NaN) comes out as all zeros
// (zero Jacobians, w = 1, rho = 0) without a branch: 1 / p_2 and the measurement are replaced, nothing else
__device__ __forceinline__ void ba2_linearize(const double* __restrict__ K, const double* __restrict__ cam, const double X[3],
                                              double uo, double vo, double delta, ba2_lin& o) {
  const bool have = uo == uo;
  const double p0 = fma(cam[0], X[0], fma(cam[1], X[1], fma(cam[2], X[2], cam[9])));
  const double p1 = fma(cam[3], X[0], fma(cam[4], X[1], fma(cam[5], X[2], cam[10])));
  const double p2 = fma(cam[6], X[0], fma(cam[7], X[1], fma(cam[8], X[2], cam[11])));
  const double ip2 = have ? rcp_nr(p2) : 0.0;
  const double u = p0 * ip2, v = p1 * ip2;
  o.e0 = u - (have ? uo : 0.0); o.e1 = v - (have ? vo : 0.0);
  const double s = o.e0 * o.e0 + o.e1 * o.e1;
  const double d2 = delta * delta;
  const bool inl = s <= d2;
  const double irs = rsqrt_nr(inl ? 1.0 : s);          // 1 / |e| (outliers only)
  o.w = inl ? 1.0 : delta * irs;
  o.rho = inl ? s : 2.0 * delta * (s * irs) - d2;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    o.Jl[0][c] = (cam[c] - u * cam[6 + c]) * ip2;
    o.Jl[1][c] = (cam[3 + c] - v * cam[6 + c]) * ip2;
    o.A[0][c] = (K[c] - u * K[6 + c]) * ip2;
    o.A[1][c] = (K[3 + c] - v * K[6 + c]) * ip2;
  }
}

// ------------------------------------------------------------------------------------------------
// k_ba_build_w
// ------------------------------------------------------------------------------------------------
// RT = 16-column blocks of the panel (RP = 16 RT >= 6 W + 1), SPL = window slots per lane (W <= LPP SPL), LPP = lanes per landmark.
// grid (G0 * batch), 256 lanes; partial set = the part of the problem a workgroup serves (ba2_select_work)
template <int RT, int SPL, int LPP>
__global__ void __launch_bounds__(256, 2) k_ba_build_w(ba_ptrs Pall, ba_params_dev prm, int it, double probe_lambda, int G0, int Gcap) {
  constexpr int LPC = ba2_map<LPP>::LPC;     // landmarks of a chunk
  constexpr int LEAD = ba2_map<LPP>::LEAD;   // the lane of a landmark's group that holds its sums
  constexpr int ROWS = 3 * LPC;              // panel rows of a chunk
  constexpr int RP = 16 * RT, PITCH = RP + (LPP != 5 ? 16 : 0);   // + 16: the four rows an MFMA operand fetch touches lie on disjoint banks (12 landmarks
                                                                  // per wave: no room for it -- two workgroups of 4 x 36 x 64 doubles fill a CU's LDS)
  constexpr int NT = RT * (RT + 1) / 2;
  constexpr int REGION = ROWS * PITCH;       // doubles of LDS a wave owns
  ba2_work wk = ba2_select_work<true>(Pall, it, G0, Gcap);
  wk.prob = ba2_uniform(wk.prob); wk.part = ba2_uniform(wk.part); wk.G = ba2_uniform(wk.G);
  if (wk.prob < 0) return;                   // (uniform) more workgroups than the running problems can use
  const ba_ptrs P = ba_select(Pall, wk.prob);
  const int part = wk.part, G = wk.G;
  extern __shared__ double dyn[];            // [4 waves][ROWS][PITCH] panels | [W][21] cameras
  double* const s_cam = dyn + 4 * REGION;
  __shared__ double s_gmax[4];
  __shared__ ba_state s_st;
  __shared__ double s_esum[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // what the decision reads -- the previous state, k_ba_solve's record, the live-slot count -- is requested BEFORE the step statistics are
  // summed: one trip to memory instead of three dependent ones at the head of every workgroup
  const int n_live_ld = ba2_uniform(Pall.n_live ? *P.n_live : P.N);
  ba_state prev;
  ba_info inf;
  if (it > 0) {
    if (tid == 0) { prev = P.state[(it - 1) & 1]; inf = *P.info; }
    ba_reduce_evalpart<256>(P.sharded ? P.xstat : P.evalpart, P.sharded ? 1 : Pall.gdyn[((it - 1) & 1) * Pall.batch + wk.prob], s_esum);
  }
  if (tid == 0) {
    ba_state st;
    if (it == 0) st = ba_init_state(prm);
    else ba_decide(prev, inf, s_esum, prm, st);
    if (probe_lambda >= 0) st.lambda = probe_lambda;
    s_st = st;
    if (part == 0) { P.state[it & 1] = st; Pall.gdyn[(it & 1) * Pall.batch + wk.prob] = G; }
  }
  __syncthreads();
  ba_state st = s_st;
  st.done = ba2_uniform(st.done); st.cur = ba2_uniform(st.cur); st.lambda = ba2_uniform(st.lambda);
  if (st.done) return;
  unsigned long long* dbgb = (blockIdx.x == 0 && P.dbg) ? P.dbg + 16 : nullptr;
  VO_STAMP(dbgb, 0);
  const int W = P.W, N = P.N;
  const double* poses = (it == 0) ? P.x0 : ba_x(P, st.cur);
  const double* pts = poses + 6 * W;
  // (the first chunk's landmark and observations are requested before the cameras are staged: two trips to memory overlap)
  const ba2_map<LPP> mp(lane);
  const int pl = mp.pl, q = mp.q;
  const int lead4 = 4 * (lane - q + LEAD);        // (LPP = 5) ds_bpermute source: the leader of this lane's group
  const int nchunk = (N + LPC - 1) / LPC;
  const int n_live = n_live_ld;
  const int nchunk_live = min(nchunk, (n_live + LPC - 1) / LPC);
  const bool seed_x = it == 0;
  double* const pan = dyn + wave * REGION;
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
  double camacc[7 * SPL];
#pragma unroll
  for (int t = 0; t < 7 * SPL; t++) camacc[t] = 0.0;
  double gm = 0.0;
  const double lam = st.lambda, delta = prm.delta;
  // (K in SCALAR registers: as vector registers its nine values were 18 of the 256, eight of them spilled and reloaded twice per chunk)
  const double Kk[9] = {ba2_uniform(P.K[0]), ba2_uniform(P.K[1]), ba2_uniform(P.K[2]), ba2_uniform(P.K[3]), ba2_uniform(P.K[4]), ba2_uniform(P.K[5]),
                        ba2_uniform(P.K[6]), ba2_uniform(P.K[7]), ba2_uniform(P.K[8])};
  // this lane's landmark and observations of a chunk; the NEXT chunk's are requested before the Gram phase of this one (a trip to HBM / L2
  // at the head of every chunk otherwise, with one other wave on the SIMD to cover it)
  double Xn[3], uon[SPL], von[SPL];
  auto fetch = [&](const int ch) {
    const int jn = ch * LPC + pl;
    Xn[0] = Xn[1] = Xn[2] = 0.0;
    if (ch < nchunk && jn < N && !mp.idle) { Xn[0] = pts[3 * jn]; Xn[1] = pts[3 * jn + 1]; Xn[2] = pts[3 * jn + 2]; }
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const int s = q + LPP * i;
      uon[i] = __builtin_nan(""); von[i] = 0.0;
      if (ch < nchunk_live && s < W && jn < N && !mp.idle) { const double2 ob = *reinterpret_cast<const double2*>(P.obs + ((size_t)s * N + jn) * 2); uon[i] = ob.x; von[i] = ob.y; }
    }
  };
  fetch(part * 4 + wave);
  // cameras: [R | t | Jr] from the poses (iteration 0) or as k_ba_solve left them, then K R, K t per camera
  if (tid < W) {
    double c[BA_CAM];
    if (it == 0) d_camera(poses + 6 * tid, c);
    else {
      const double* cg = P.cams + ((size_t)st.cur * W + tid) * BA_CAM;
#pragma unroll
      for (int k = 0; k < BA_CAM; k++) c[k] = cg[k];
    }
    ba2_stage_camera(P.K, c, s_cam + BA2_CAM * tid);
    if (it == 0 && part == 0) {
#pragma unroll
      for (int k = 0; k < BA_CAM; k++) P.cams[(size_t)tid * BA_CAM + k] = c[k];   // cams[0] <-> x[0]
    }
  }
  if (it == 0 && part == 0 && tid < 6 * W) P.xa[tid] = poses[tid];
  // the padding columns behind 6 W + 1 start as zeros and are never written afterwards (every chunk rewrites the columns before them in
  // every row: a lane without a landmark writes zeros)
  {
    const int npad = RP - (6 * W + 1);
    for (int i = tid; i < 4 * ROWS * npad; i += 256) { const int r = i / npad; dyn[r * PITCH + 6 * W + 1 + (i - r * npad)] = 0.0; }
    if (PITCH > RP) for (int i = tid; i < 4 * ROWS * (PITCH - RP); i += 256) { const int r = i / (PITCH - RP); dyn[r * PITCH + RP + (i - r * (PITCH - RP))] = 0.0; }
  }
  __syncthreads();
  int nwalk = 0;
#pragma unroll 1
  for (int chunk = part * 4 + wave; chunk < nchunk; chunk += 4 * G) {
    const int j = chunk * LPC + pl;
    const bool inr = j < N && !mp.idle;
    if (nwalk == 2) VO_STAMP(dbgb, 1);   // (diagnostic stamps 1..4: the third chunk of the walk, steady state)
    const double X[3] = {Xn[0], Xn[1], Xn[2]};
    double uo[SPL], vo[SPL];
#pragma unroll
    for (int i = 0; i < SPL; i++) { uo[i] = uon[i]; vo[i] = von[i]; }
    if (seed_x && q == LEAD && inr) { double* dst = P.xa + 6 * W + 3 * j; dst[0] = X[0]; dst[1] = X[1]; dst[2] = X[2]; }
    if (chunk >= nchunk_live) { fetch(chunk + 4 * G); continue; }      // (wave-uniform) an unused part of the table: only x[0] is seeded
    // ---- pass 1: residual, weight and landmark block of this lane's observations; landmark sums over its slots.  What pass 2 needs again is
    //      kept per slot as 7 values (u, v, 1 / p_2, w, e, rho): the Jacobian blocks of ALL slots of a lane (36 values each) do not fit the
    //      256 registers beside the 80 accumulators, and d(u, v)/dX is 12 operations to form again ----
    double ku[SPL], kv[SPL], kip[SPL], kw[SPL], ke0[SPL], ke1[SPL], krho[SPL];
    double h00 = 0, h10 = 0, h11 = 0, h20 = 0, h21 = 0, h22 = 0, g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const int s = min(q + LPP * i, W - 1);      // (a lane without an i-th slot computes zeros on a valid camera)
      const double* cam = s_cam + BA2_CAM * s;
      const bool have = uo[i] == uo[i];
      const double p0 = fma(cam[0], X[0], fma(cam[1], X[1], fma(cam[2], X[2], cam[9])));
      const double p1 = fma(cam[3], X[0], fma(cam[4], X[1], fma(cam[5], X[2], cam[10])));
      const double p2 = fma(cam[6], X[0], fma(cam[7], X[1], fma(cam[8], X[2], cam[11])));
      const double ip2 = have ? rcp_nr(p2) : 0.0;             // an unobserved slot: zero Jacobians, zero residual (w = 1, rho = 0)
      const double u = p0 * ip2, v = p1 * ip2;
      const double e0 = u - (have ? uo[i] : 0.0), e1 = v - (have ? vo[i] : 0.0);
      const double sq = e0 * e0 + e1 * e1;
      const double d2 = delta * delta;
      const bool inl = sq <= d2;
      const double irs = rsqrt_nr(inl ? 1.0 : sq);            // 1 / |e| (outliers only)
      const double w = inl ? 1.0 : delta * irs;
      ku[i] = u; kv[i] = v; kip[i] = ip2; kw[i] = w;
      ke0[i] = e0; ke1[i] = e1;
      krho[i] = inl ? sq : 2.0 * delta * (sq * irs) - d2;
      double l0[3], l1[3];
#pragma unroll
      for (int c = 0; c < 3; c++) { l0[c] = (cam[c] - u * cam[6 + c]) * ip2; l1[c] = (cam[3 + c] - v * cam[6 + c]) * ip2; }
      const double wl0[3] = {w * l0[0], w * l0[1], w * l0[2]};
      const double wl1[3] = {w * l1[0], w * l1[1], w * l1[2]};
      h00 += wl0[0] * l0[0] + wl1[0] * l1[0];
      h10 += wl0[1] * l0[0] + wl1[1] * l1[0];
      h11 += wl0[1] * l0[1] + wl1[1] * l1[1];
      h20 += wl0[2] * l0[0] + wl1[2] * l1[0];
      h21 += wl0[2] * l0[1] + wl1[2] * l1[1];
      h22 += wl0[2] * l0[2] + wl1[2] * l1[2];
      g0 += wl0[0] * e0 + wl1[0] * e1;
      g1 += wl0[1] * e0 + wl1[1] * e1;
      g2 += wl0[2] * e0 + wl1[2] * e1;
    }
    h00 = ba2_group_sum<LPP>(h00); h10 = ba2_group_sum<LPP>(h10); h11 = ba2_group_sum<LPP>(h11);
    h20 = ba2_group_sum<LPP>(h20); h21 = ba2_group_sum<LPP>(h21); h22 = ba2_group_sum<LPP>(h22);
    g0 = ba2_group_sum<LPP>(g0); g1 = ba2_group_sum<LPP>(g1); g2 = ba2_group_sum<LPP>(g2);
    // ---- damped 3x3 block: Cholesky C C^T, Cinv = C^-1 (lower), y = Cinv g, z = Cinv^T y = M g ----
    const double a00 = h00 + lam * fmax(h00, 1e-12), a11 = h11 + lam * fmax(h11, 1e-12), a22 = h22 + lam * fmax(h22, 1e-12);
    double i00 = rsqrt_nr(a00);
    const double c10 = h10 * i00, c20 = h20 * i00;
    double i11 = rsqrt_nr(a11 - c10 * c10);
    const double c21 = (h21 - c20 * c10) * i11;
    double i22 = rsqrt_nr(a22 - c20 * c20 - c21 * c21);
    double i10 = -c10 * i00 * i11;
    double i21 = -c21 * i11 * i22;
    double i20 = -(c20 * i00 + c21 * i10) * i22;
    const double y0 = i00 * g0, y1 = i10 * g0 + i11 * g1, y2 = i20 * g0 + i21 * g1 + i22 * g2;
    if (nwalk == 2) VO_STAMP(dbgb, 2);   // pass 1 + factor
    if (q == LEAD && inr) {
      double2* ax = reinterpret_cast<double2*>(P.aux + (size_t)j * BA_AUX);
      ax[0] = make_double2(h00, h10); ax[1] = make_double2(h11, h20); ax[2] = make_double2(h21, h22);
      ax[3] = make_double2(g0, g1); ax[4] = make_double2(g2, i00); ax[5] = make_double2(i10, i11);
      ax[6] = make_double2(i20, i21); ax[7] = make_double2(i22, i00 * y0 + i10 * y1 + i20 * y2);
      ax[8] = make_double2(i11 * y1 + i21 * y2, i22 * y2);
    }
    if (inr && (LPP != 5 || q == LEAD)) gm = fmax(gm, fmax(fabs(g0), fmax(fabs(g1), fabs(g2))));
    if (LPP == 5) {      // the factor of the landmark to every lane of its group
      i00 = ba2_from_leader<LPP>(i00, lead4); i10 = ba2_from_leader<LPP>(i10, lead4); i11 = ba2_from_leader<LPP>(i11, lead4);
      i20 = ba2_from_leader<LPP>(i20, lead4); i21 = ba2_from_leader<LPP>(i21, lead4); i22 = ba2_from_leader<LPP>(i22, lead4);
    }
    // ---- pass 2, slot by slot: the camera block Jp = [(X x Jl_k)^T Jr | A], the slot's camera sums, the landmark's panel rows
    //      Y[a][c] = sum_k Jp[k][a] (w Z[k][c]),  Z = Jl Cinv^T;  column 6 W = y ----
    double* const rw0 = pan + (3 * pl) * PITCH;
    double* const rw1 = rw0 + PITCH;
    double* const rw2 = rw1 + PITCH;
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const int sr = q + LPP * i, s = min(sr, W - 1);
      const double* cam = s_cam + BA2_CAM * s;
      const double u = ku[i], v = kv[i], ip2 = kip[i], w = kw[i];
      double Jl[2][3], Jp[2][6];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Jl[0][c] = (cam[c] - u * cam[6 + c]) * ip2; Jl[1][c] = (cam[3 + c] - v * cam[6 + c]) * ip2;
        Jp[0][3 + c] = (Kk[c] - u * Kk[6 + c]) * ip2; Jp[1][3 + c] = (Kk[3 + c] - v * Kk[6 + c]) * ip2;
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const double c0 = X[1] * Jl[k][2] - X[2] * Jl[k][1];
        const double c1 = X[2] * Jl[k][0] - X[0] * Jl[k][2];
        const double c2 = X[0] * Jl[k][1] - X[1] * Jl[k][0];
#pragma unroll
        for (int c = 0; c < 3; c++) Jp[k][c] = c0 * cam[12 + c] + c1 * cam[15 + c] + c2 * cam[18 + c];
      }
      // camera sums of the slot: 28 values (21 of the upper H_pp, 6 of g_p, the cost) four at a time through two reduce-scatter stages over
      // the landmark bits 5 and 4 of the lane index: the lane with bits (b5, b4) then holds value 4 n + 2 b4 + b5 summed over four of the
      // chunk's landmarks, and keeps adding to it chunk after chunk; the remaining landmark bits are summed once, after the walk
      constexpr int QA[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
      constexpr int QC[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
      const double we0 = w * ke0[i], we1 = w * ke1[i], hrho = 0.5 * krho[i];
      auto term = [&](int t) -> double {
        if (t < 21) return w * (Jp[0][QA[t]] * Jp[0][QC[t]] + Jp[1][QA[t]] * Jp[1][QC[t]]);
        if (t < 27) return Jp[0][t - 21] * we0 + Jp[1][t - 21] * we1;
        return hrho;
      };
#pragma unroll
      for (int n = 0; n < 7; n++) {
        camacc[7 * i + n] += rs16_sum(rs32_sum(term(4 * n), term(4 * n + 1)), rs32_sum(term(4 * n + 2), term(4 * n + 3)));
        // the sum is taken HERE: left to itself the compiler carries the 14 reduce-scatter results through the Gram phase to the loop's latch and
        // spills half of the accumulators to make room for them (7 x 8 bytes of scratch stored and reloaded per chunk)
        asm volatile("" : "+v"(camacc[7 * i + n]));
      }
      if (sr < W && !mp.idle) {
        double Z[2][3];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const double wa = w * Jl[k][0], wb = w * Jl[k][1], wc = w * Jl[k][2];
          Z[k][0] = i00 * wa;
          Z[k][1] = i10 * wa + i11 * wb;
          Z[k][2] = i20 * wa + i21 * wb + i22 * wc;
        }
#pragma unroll
        for (int a = 0; a < 6; a += 2) {
          *reinterpret_cast<double2*>(rw0 + 6 * sr + a) = make_double2(Jp[0][a] * Z[0][0] + Jp[1][a] * Z[1][0], Jp[0][a + 1] * Z[0][0] + Jp[1][a + 1] * Z[1][0]);
          *reinterpret_cast<double2*>(rw1 + 6 * sr + a) = make_double2(Jp[0][a] * Z[0][1] + Jp[1][a] * Z[1][1], Jp[0][a + 1] * Z[0][1] + Jp[1][a + 1] * Z[1][1]);
          *reinterpret_cast<double2*>(rw2 + 6 * sr + a) = make_double2(Jp[0][a] * Z[0][2] + Jp[1][a] * Z[1][2], Jp[0][a + 1] * Z[0][2] + Jp[1][a + 1] * Z[1][2]);
        }
      }
    }
    if (q == LEAD && !mp.idle) { rw0[6 * W] = inr ? y0 : 0.0; rw1[6 * W] = inr ? y1 : 0.0; rw2[6 * W] = inr ? y2 : 0.0; }
    // ---- Gram matrix of the chunk's panel into the accumulators: one operand fetch per column block and k-step ----
    //   A[i][k] = panel[k0 + k][16 ta + i]  (lane: i = l & 15, k = l >> 4),  B[k][j] = panel[k0 + k][16 tb + j]
    //   D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    __builtin_amdgcn_wave_barrier();
    if (nwalk == 2) VO_STAMP(dbgb, 3);   // first chunk: pass 2, panel written
    fetch(chunk + 4 * G);
    {
      const double* base = pan + (lane >> 4) * PITCH + (lane & 15);
#pragma unroll
      for (int k0 = 0; k0 < ROWS; k0 += 4) {
        double op[RT];
#pragma unroll
        for (int c = 0; c < RT; c++) op[c] = base[k0 * PITCH + 16 * c];
        int t = 0;
#pragma unroll
        for (int ta = 0; ta < RT; ta++)
#pragma unroll
          for (int tb = ta; tb < RT; tb++, t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[ta], op[tb], acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (nwalk == 2) VO_STAMP(dbgb, 4);
    nwalk++;   // first chunk: Gram
  }
  VO_STAMP(dbgb, 5);   // walk
  // ---- after the walk: the landmark bits the chunk stages left, then the four waves' sums through LDS in wave order ----
#pragma unroll
  for (int t = 0; t < 7 * SPL; t++) camacc[t] = ba2_finish_landmark_sum<LPP>(camacc[t]);
  for (int ofs = 32; ofs > 0; ofs >>= 1) gm = fmax(gm, __shfl_xor(gm, ofs));
  if (lane == 0) s_gmax[wave] = gm;
  constexpr int WMAX = (RP - 1) / 6 < LPP * SPL ? (RP - 1) / 6 : LPP * SPL;   // largest window this instance serves
  constexpr int CAMV = WMAX * BA_POSE_VALS;                  // room for the camera sums behind the tiles of a round
  constexpr int TPR = (REGION - CAMV) / 256 < NT ? (REGION - CAMV) / 256 : NT;   // tiles per round
  static_assert(TPR >= 1, "a wave's LDS region holds at least one tile beside the camera sums");
  double* const camsum = pan + TPR * 256;
  {
    // writer lanes: the remaining landmark bits clear (every lane of a (b5, b4, q) class holds the same total)
    const int b5 = lane >> 5, b4 = (lane >> 4) & 1;
    const bool wr = (LPP == 8) ? ((lane & 8) == 0) : (LPP == 4) ? ((lane & 12) == 0) : ((lane & 15) < 5);
    if (wr) {
#pragma unroll
      for (int i = 0; i < SPL; i++) {
        const int s = q + LPP * i;
        if (s < W) {
#pragma unroll
          for (int n = 0; n < 7; n++) camsum[s * BA_POSE_VALS + 4 * n + 2 * b4 + b5] = camacc[7 * i + n];
        }
      }
    }
  }
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TPR) {
    if (t0 > 0) __syncthreads();                            // the previous round has been read
#pragma unroll
    for (int t = t0; t < NT && t < t0 + TPR; t++) {
      double2* o2 = reinterpret_cast<double2*>(pan + (t - t0) * 256 + lane * 4);
      o2[0] = make_double2(acc[t][0], acc[t][1]);
      o2[1] = make_double2(acc[t][2], acc[t][3]);
    }
    __syncthreads();
    const int nel = (NT - t0 < TPR ? NT - t0 : TPR) * 256;
    double* out = P.tiles + ((size_t)part * NT + t0) * 256;
    for (int e = tid; e < nel; e += 256)
      out[e] = (dyn[e] + dyn[REGION + e]) + (dyn[2 * REGION + e] + dyn[3 * REGION + e]);
    if (t0 == 0) {
      for (int t = tid; t < W * BA_POSE_VALS; t += 256) {
        const int o = TPR * 256 + t;
        P.posepart[(size_t)part * W * BA_POSE_VALS + t] = (dyn[o] + dyn[REGION + o]) + (dyn[2 * REGION + o] + dyn[3 * REGION + o]);
      }
      if (tid == 0) P.gmax[part] = fmax(fmax(s_gmax[0], s_gmax[1]), fmax(s_gmax[2], s_gmax[3]));
    }
  }
  VO_STAMP(dbgb, 6);
}

// ------------------------------------------------------------------------------------------------
// k_ba_update_w : the same wave walk for the landmark back-substitution, the trial x and the trial cost
// ------------------------------------------------------------------------------------------------
// grid (G0 * batch), 256 lanes, the same assignment as k_ba_build_w of the iteration; step statistics: one entry of evalpart per part
template <int SPL, int LPP>
__global__ void __launch_bounds__(256, 2) k_ba_update_w(ba_ptrs Pall, ba_params_dev prm, int it, double* __restrict__ probe_dl, int G0, int Gcap) {
  constexpr int LPC = ba2_map<LPP>::LPC, LEAD = ba2_map<LPP>::LEAD;
  ba2_work wk = ba2_select_work<false>(Pall, it, G0, Gcap);
  wk.prob = ba2_uniform(wk.prob); wk.part = ba2_uniform(wk.part); wk.G = ba2_uniform(wk.G);
  if (wk.prob < 0) return;
  const ba_ptrs P = ba_select(Pall, wk.prob);
  const int part = wk.part, G = wk.G;
  if (wk.prob != 0) probe_dl = nullptr;
  __shared__ double s_cam[BA2_CAM * 10];            // current poses: K R, K t, Jr (windows of <= 10 slots)
  __shared__ double s_camt[12 * 10];                // trial poses: K R, K t
  __shared__ double s_dp[6 * 10];
  __shared__ double s_red[4 * BA_EVAL_VALS];
  const ba_state st = P.state[it & 1];       // (a scalar load: P is uniform now)
  if (st.done) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, W = P.W, N = P.N;
  const double* poses = ba_x(P, st.cur);
  const double* pts = poses + 6 * W;
  double* tposes = ba_x(P, st.cur ^ 1);
  double* tpts = tposes + 6 * W;
  if (tid < W) {
    const double* cc = P.cams + ((size_t)st.cur * W + tid) * BA_CAM;          // current poses (k_ba_build it == 0 / k_ba_solve)
    const double* ct = P.cams + ((size_t)(st.cur ^ 1) * W + tid) * BA_CAM;    // trial poses (k_ba_solve of this iteration)
    double c[BA_CAM], t[BA2_CAM];
#pragma unroll
    for (int k = 0; k < BA_CAM; k++) c[k] = cc[k];
    ba2_stage_camera(P.K, c, s_cam + BA2_CAM * tid);
#pragma unroll
    for (int k = 0; k < 12; k++) c[k] = ct[k];
    ba2_stage_camera(P.K, c, t);
#pragma unroll
    for (int k = 0; k < 12; k++) s_camt[12 * tid + k] = t[k];
  }
  for (int a = tid; a < 6 * W; a += 256) {
    const double d = P.dp[a];
    s_dp[a] = d;
    if (part == 0) tposes[a] = poses[a] + d;
  }
  __syncthreads();
  const ba2_map<LPP> mp(lane);
  const int pl = mp.pl, q = mp.q;
  const int lead4 = 4 * (lane - q + LEAD);
  const int nchunk = (N + LPC - 1) / LPC;
  const int n_live = ba2_uniform(P.n_live ? *P.n_live : N);
  const int nchunk_live = min(nchunk, (n_live + LPC - 1) / LPC);
  const double lam = st.lambda, delta = prm.delta, d2 = delta * delta;
  const double Kk[9] = {ba2_uniform(P.K[0]), ba2_uniform(P.K[1]), ba2_uniform(P.K[2]), ba2_uniform(P.K[3]), ba2_uniform(P.K[4]), ba2_uniform(P.K[5]),
                        ba2_uniform(P.K[6]), ba2_uniform(P.K[7]), ba2_uniform(P.K[8])};
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll 1
  for (int chunk = part * 4 + wave; chunk < nchunk_live; chunk += 4 * G) {
    const int j = chunk * LPC + pl;
    const bool inr = j < N && !mp.idle;
    double X[3] = {0.0, 0.0, 0.0};
    if (inr) { X[0] = pts[3 * j]; X[1] = pts[3 * j + 1]; X[2] = pts[3 * j + 2]; }
    double uo[SPL], vo[SPL];
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const int s = q + LPP * i;
      uo[i] = __builtin_nan(""); vo[i] = 0.0;
      if (s < W && inr) { const double2 ob = *reinterpret_cast<const double2*>(P.obs + ((size_t)s * N + j) * 2); uo[i] = ob.x; vo[i] = ob.y; }
    }
    double ax[15];
    if (inr) {
      const double* axp = P.aux + (size_t)j * BA_AUX;
#pragma unroll
      for (int k = 0; k < 15; k++) ax[k] = axp[k];
    } else {
#pragma unroll
      for (int k = 0; k < 15; k++) ax[k] = 0.0;
    }
    // B^T d_pose = sum over the slots of w Jl^T (Jp d_pose), Jp d without forming Jp:  -Jl (X x (Jr d_rot)) + A d_trans
    double v0 = 0, v1 = 0, v2 = 0;
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const int s = min(q + LPP * i, W - 1);
      const double* cam = s_cam + BA2_CAM * s;
      const double* d = s_dp + 6 * s;
      ba2_lin o;
      ba2_linearize(Kk, cam, X, uo[i], vo[i], delta, o);
      const double u0 = cam[12] * d[0] + cam[13] * d[1] + cam[14] * d[2];
      const double u1 = cam[15] * d[0] + cam[16] * d[1] + cam[17] * d[2];
      const double u2 = cam[18] * d[0] + cam[19] * d[1] + cam[20] * d[2];
      const double t0 = X[1] * u2 - X[2] * u1, t1 = X[2] * u0 - X[0] * u2, t2 = X[0] * u1 - X[1] * u0;
      const double q0 = (o.A[0][0] * d[3] + o.A[0][1] * d[4] + o.A[0][2] * d[5]) - (o.Jl[0][0] * t0 + o.Jl[0][1] * t1 + o.Jl[0][2] * t2);
      const double q1 = (o.A[1][0] * d[3] + o.A[1][1] * d[4] + o.A[1][2] * d[5]) - (o.Jl[1][0] * t0 + o.Jl[1][1] * t1 + o.Jl[1][2] * t2);
      v0 += o.w * (o.Jl[0][0] * q0 + o.Jl[1][0] * q1);
      v1 += o.w * (o.Jl[0][1] * q0 + o.Jl[1][1] * q1);
      v2 += o.w * (o.Jl[0][2] * q0 + o.Jl[1][2] * q1);
    }
    v0 = ba2_from_leader<LPP>(ba2_group_sum<LPP>(v0), lead4); v1 = ba2_from_leader<LPP>(ba2_group_sum<LPP>(v1), lead4);
    v2 = ba2_from_leader<LPP>(ba2_group_sum<LPP>(v2), lead4);
    const double w0 = ax[6] + v0, w1 = ax[7] + v1, w2 = ax[8] + v2;
    const double i00 = ax[9], i10 = ax[10], i11 = ax[11], i20 = ax[12], i21 = ax[13], i22 = ax[14];
    const double t0 = i00 * w0, t1 = i10 * w0 + i11 * w1, t2 = i20 * w0 + i21 * w1 + i22 * w2;   // Cinv u
    const double dl0 = -(i00 * t0 + i10 * t1 + i20 * t2), dl1 = -(i11 * t1 + i21 * t2), dl2 = -(i22 * t2);
    const double Xt[3] = {X[0] + dl0, X[1] + dl1, X[2] + dl2};
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const int s = min(q + LPP * i, W - 1);
      const double* cam = s_camt + 12 * s;
      const bool have = uo[i] == uo[i];
      const double p0 = fma(cam[0], Xt[0], fma(cam[1], Xt[1], fma(cam[2], Xt[2], cam[9])));
      const double p1 = fma(cam[3], Xt[0], fma(cam[4], Xt[1], fma(cam[5], Xt[2], cam[10])));
      const double p2 = fma(cam[6], Xt[0], fma(cam[7], Xt[1], fma(cam[8], Xt[2], cam[11])));
      const double ip2 = have ? rcp_nr(p2) : 0.0;
      const double r0 = p0 * ip2 - (have ? uo[i] : 0.0), r1 = p1 * ip2 - (have ? vo[i] : 0.0);
      const double sq = r0 * r0 + r1 * r1;
      const bool inl = sq <= d2;
      const double irs = rsqrt_nr(inl ? 1.0 : sq);
      e0 += 0.5 * (inl ? sq : 2.0 * delta * (sq * irs) - d2);
    }
    if (q == LEAD && inr) {
      tpts[3 * j] = Xt[0]; tpts[3 * j + 1] = Xt[1]; tpts[3 * j + 2] = Xt[2];
      if (probe_dl) { probe_dl[3 * j] = dl0; probe_dl[3 * j + 1] = dl1; probe_dl[3 * j + 2] = dl2; }
      e1 += lam * (fmax(ax[0], 1e-12) * dl0 * dl0 + fmax(ax[2], 1e-12) * dl1 * dl1 + fmax(ax[5], 1e-12) * dl2 * dl2)
            - (ax[6] * dl0 + ax[7] * dl1 + ax[8] * dl2);
      e2 += dl0 * dl0 + dl1 * dl1 + dl2 * dl2;
      e3 += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
    }
  }
  // four wave-wide sums as a reduce-scatter: after the two swap stages the 16-lane row r of the wave holds, per lane, the column sums of
  // value {e0, e2, e1, e3}[r]; one row all-reduce finishes all four at once
  {
    double u = rs16_sum(rs32_sum(e0, e1), rs32_sum(e2, e3));
    u += dpp_f64<0x128>(u); u += dpp_f64<0x124>(u); u += dpp_f64<0x122>(u); u += dpp_f64<0x121>(u);   // row_ror 8, 4, 2, 1
    if ((lane & 15) == 0) {
      const int r = lane >> 4;
      s_red[wave * 4 + ((r & 1) ? (r == 1 ? 2 : 3) : (r == 0 ? 0 : 1))] = u;
    }
  }
  __syncthreads();
  if (tid < BA_EVAL_VALS) P.evalpart[part * BA_EVAL_VALS + tid] = (s_red[tid] + s_red[4 + tid]) + (s_red[8 + tid] + s_red[12 + tid]);
}
