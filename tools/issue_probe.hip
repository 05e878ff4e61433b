// Diagnostic: vector-instruction ISSUE RATE of the instruction kinds k_klt_track is made of, per SIMD, as a function of the
// number of waves that share the SIMD.  Answers "what is the VALU roof for THIS instruction mix" (bench.py roofline.valu).
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe ; run: /tmp/issue_probe
// Every test body is 32 independent instructions (distinct destination registers) inside a loop; all 256 CUs run it with
// W waves per SIMD; cycles come from s_memtime around the loop of each wave (median over waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY32(INS) asm volatile(REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") REP8(INS "\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) :);

template <int OP>
__global__ void probe(unsigned long long* out, int iters, unsigned seed) {
  unsigned a = threadIdx.x * 3 + seed, b = threadIdx.x * 5 + 1, c = threadIdx.x + 7, d = threadIdx.x * 11 + 3, e = threadIdx.x ^ 0x5555u, f = 0x01020304u + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) BODY32("v_dot2c_i32_i16 %0, %4, %5\n v_dot2c_i32_i16 %1, %4, %5\n v_dot2c_i32_i16 %2, %4, %5\n v_dot2c_i32_i16 %3, %4, %5")
    if (OP == 1) BODY32("v_perm_b32 %0, %4, %5, %0\n v_perm_b32 %1, %4, %5, %1\n v_perm_b32 %2, %4, %5, %2\n v_perm_b32 %3, %4, %5, %3")
    if (OP == 2) BODY32("v_add_u32_dpp %0, %4, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %4, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %3 row_ror:4 row_mask:0xf bank_mask:0xf")
    if (OP == 3) BODY32("v_pk_sub_i16 %0, %4, %0\n v_pk_sub_i16 %1, %4, %1\n v_pk_sub_i16 %2, %4, %2\n v_pk_sub_i16 %3, %4, %3")
    if (OP == 4) BODY32("v_ashrrev_i32 %0, 9, %0\n v_ashrrev_i32 %1, 9, %1\n v_ashrrev_i32 %2, 9, %2\n v_ashrrev_i32 %3, 9, %3")
    if (OP == 5) BODY32("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3")
    if (OP == 6) BODY32("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3")
    if (OP == 7) BODY32("v_mad_i32_i24 %0, %4, %5, %0\n v_mad_i32_i24 %1, %4, %5, %1\n v_mad_i32_i24 %2, %4, %5, %2\n v_mad_i32_i24 %3, %4, %5, %3")
    if (OP == 8) BODY32("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3")
    if (OP == 9) BODY32("v_pk_mad_i16 %0, %4, %5, %0\n v_pk_mad_i16 %1, %4, %5, %1\n v_pk_mad_i16 %2, %4, %5, %2\n v_pk_mad_i16 %3, %4, %5, %3")
    if (OP == 10) BODY32("v_mov_b32_dpp %0, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")
    if (OP == 11) BODY32("v_fma_f64 %0, %4, %5, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3")   // placeholder, unused
    if (OP == 12) BODY32("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n v_cndmask_b32 %3, %4, %3, vcc")
    if (OP == 13) BODY32("v_and_b32 %0, %4, %0\n v_lshlrev_b32 %1, 3, %1\n v_and_b32 %2, %4, %2\n v_lshlrev_b32 %3, 3, %3")
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a + b + c + d == 0x12345678u) out[0] = a;     // keep the results alive
}

template <int OP>
static void run(const char* name, unsigned long long* d) {
  const int iters = 200, per_iter = 128;
  for (int w : {1, 2, 4, 8}) {
    const int threads = 64 * 4 * w;            // one workgroup per CU, w waves on each of its 4 SIMDs
    if (threads > 1024) {                      // 8 waves per SIMD: two workgroups of 1024 per CU
      hipLaunchKernelGGL(probe<OP>, dim3(512), dim3(1024), 0, 0, d, iters, 1u);
    } else {
      hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, d, iters, 1u);
    }
    hipDeviceSynchronize();
    const int nw = (threads > 1024) ? 512 * 16 : 256 * 4 * w;
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[nw / 2];
    printf("%-18s waves/SIMD %d : %.2f cycles per instruction per wave, %.2f cycles per instruction per SIMD\n", name, w,
           cyc / (iters * per_iter), cyc / (iters * per_iter) / w);
  }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 16384);
  run<6>("v_fma_f32", d);
  run<5>("v_add_u32", d);
  run<0>("v_dot2c_i32_i16", d);
  run<8>("v_dot4_u32_u8", d);
  run<1>("v_perm_b32", d);
  run<2>("v_add_u32 dpp", d);
  run<10>("v_mov_b32 dpp quad", d);
  run<3>("v_pk_sub_i16", d);
  run<9>("v_pk_mad_i16", d);
  run<4>("v_ashrrev_i32", d);
  run<7>("v_mad_i32_i24", d);
  run<12>("v_cndmask_b32", d);
  run<13>("v_and/v_lshl", d);
  return 0;
}
