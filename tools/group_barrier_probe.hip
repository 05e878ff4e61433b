// Diagnostic for the bundle-adjustment launch structure: what does ONE synchronisation point between the workgroups of a
// problem cost when it is (a) a kernel boundary, as today (k_ba_build -> k_ba_reduce -> k_ba_solve -> k_ba_update), and when it
// is (b, c) a barrier among the problem's G workgroups INSIDE one persistent launch?
//   geometry of the batched solve: 32 problems x G = 32 workgroups of 256 threads = 1024 workgroups, 4 per CU (all co-resident)
//   every phase: each workgroup writes a 2 KB record (its partial sums) that the other workgroups of ITS problem read next phase
//   (a) phases as separate launches of the same body, back to back on one stream
//   (b) persistent, plain stores + agent-scope release fence -> arrive on the problem's counter -> poll -> acquire fence
//   (c) persistent, write-through (sc1) stores and sc1 loads of the records, no fence (MI355X_MICROARCH.md, "Valid forms")
// Prints microseconds per phase (launch wall time / phases) and whether every record that was read was the fresh one.
// Build: hipcc --offload-arch=gfx950 -O3 tools/group_barrier_probe.hip -o /tmp/gbp ; run: /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define G 32
#define B 32
#define REC 256            // doubles per record (2 KB)

__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one phase of "work": write my record (value encodes the phase), nothing else
template <bool SC1>
__device__ __forceinline__ void write_record(double* rec, int wg, int prob, int phase) {
  double* r = rec + ((size_t)prob * G + wg) * REC;
  const double v = (double)(phase * 1000 + wg);
  if (SC1) st_sc1(r + threadIdx.x, v); else r[threadIdx.x] = v;
}
// read the records of all workgroups of my problem written in `phase` and count the stale ones
template <bool SC1>
__device__ __forceinline__ int read_records(const double* rec, int prob, int phase) {
  int bad = 0;
  for (int w = 0; w < G; w++) {
    const double* r = rec + ((size_t)prob * G + w) * REC;
    const double v = SC1 ? ld_sc1(r + threadIdx.x) : r[threadIdx.x];
    bad += (v != (double)(phase * 1000 + w));
  }
  return bad;
}

__global__ void __launch_bounds__(256) k_phase(double* rec, int* bad_out, int phase) {
  const int prob = blockIdx.y, wg = blockIdx.x;
  // records are double-buffered by phase parity so that a fast workgroup cannot overwrite what a slow one still reads
  int bad = (phase > 0) ? read_records<false>(rec + (size_t)((phase - 1) & 1) * B * G * REC, prob, phase - 1) : 0;
  write_record<false>(rec + (size_t)(phase & 1) * B * G * REC, wg, prob, phase);
  if (bad) atomicAdd(bad_out, bad);
}

template <int MODE>   // 1: fences, 2: sc1
__global__ void __launch_bounds__(256) k_persistent(double* rec, unsigned* counters, int* bad_out, int phases, unsigned* timeout) {
  const int prob = blockIdx.y, wg = blockIdx.x;
  unsigned* cnt = counters + prob * 32;            // one 128-B line per problem
  int bad = 0;
  for (int phase = 0; phase < phases; phase++) {
    double* buf = rec + (size_t)(phase & 1) * B * G * REC;
    if (MODE == 1) write_record<false>(buf, wg, prob, phase); else write_record<true>(buf, wg, prob, phase);
    // ---- barrier among the G workgroups of this problem ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(phase + 1) * G;
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 20000000u) { *timeout = 1; break; }
      }
      if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    if (MODE == 1) bad += read_records<false>(buf, prob, phase); else bad += read_records<true>(buf, prob, phase);
  }
  if (bad) atomicAdd(bad_out, bad);
}

int main() {
  double* rec; unsigned* counters; int* bad; unsigned* tmo;
  hipMalloc(&rec, sizeof(double) * 2 * B * G * REC);
  hipMalloc(&counters, sizeof(unsigned) * B * 32);
  hipMalloc(&bad, 4); hipMalloc(&tmo, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int phases = 200;
  float ms; int hbad; unsigned htmo;
  // (a) kernel boundaries
  for (int rep = 0; rep < 2; rep++) {
    hipMemset(bad, 0, 4);
    hipEventRecord(e0, 0);
    for (int p = 0; p < phases; p++) hipLaunchKernelGGL(k_phase, dim3(G, B), dim3(256), 0, 0, rec, bad, p);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    if (rep) printf("(a) kernel boundary per phase        : %6.2f us  (stale reads %d)\n", ms * 1e3 / phases, hbad);
  }
  for (int mode = 1; mode <= 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(bad, 0, 4); hipMemset(tmo, 0, 4); hipMemset(counters, 0, sizeof(unsigned) * B * 32);
      hipEventRecord(e0, 0);
      if (mode == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(G, B), dim3(256), 0, 0, rec, counters, bad, phases, tmo);
      else hipLaunchKernelGGL(k_persistent<2>, dim3(G, B), dim3(256), 0, 0, rec, counters, bad, phases, tmo);
      hipEventRecord(e1, 0); hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost);
      if (rep) printf("(%c) in-launch barrier, %-14s: %6.2f us  (stale reads %d, timeout %u)\n", mode == 1 ? 'b' : 'c',
                      mode == 1 ? "fences" : "sc1 no fence", ms * 1e3 / phases, hbad, htmo);
    }
  }
  return 0;
}
