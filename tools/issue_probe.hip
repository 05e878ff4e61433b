// Diagnostic: vector-instruction ISSUE RATE of the instruction kinds the hot kernels are made of, per SIMD, as a function of
// the number of waves that share the SIMD.  Answers "what is the VALU roof for THIS instruction mix" (bench.py roofline.valu).
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe ; run: /tmp/issue_probe
// A test body is 64 instructions in 8 independent dependency chains inside a long loop; one workgroup of 4 W waves per CU puts W
// waves on every SIMD of all 256 CUs.  The whole launch is timed with hipEvents (milliseconds of work, so launch overheads
// vanish) and converted to shader cycles with the clock the kernel itself observes (s_memtime ticks per s_memrealtime tick).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define R8(x) x x x x x x x x
// eight chains: operands %0..%7 read-write, %8 / %9 read-only
#define BODY(I0, I1, I2, I3, I4, I5, I6, I7) \
  asm volatile(R8(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n") \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e), "v"(f) :);
#define SAME(INS) BODY(INS(0), INS(1), INS(2), INS(3), INS(4), INS(5), INS(6), INS(7))

#define I_FMA32(k) "v_fma_f32 %" #k ", %8, %9, %" #k
#define I_ADD(k) "v_add_u32 %" #k ", %8, %" #k
#define I_AND(k) "v_and_b32 %" #k ", %8, %" #k
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 3, %" #k
#define I_ASHR(k) "v_ashrrev_i32 %" #k ", 9, %" #k
#define I_DOT2C(k) "v_dot2c_i32_i16 %" #k ", %8, %9"
#define I_DOT4(k) "v_dot4_u32_u8 %" #k ", %8, %9, %" #k
#define I_PERM(k) "v_perm_b32 %" #k ", %8, %9, %" #k
#define I_ADDDPP(k) "v_add_u32_dpp %" #k ", %8, %" #k " row_ror:4 row_mask:0xf bank_mask:0xf"
#define I_MOVDPP(k) "v_mov_b32_dpp %" #k ", %8 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf"
#define I_PKSUB(k) "v_pk_sub_i16 %" #k ", %8, %" #k
#define I_MAD24(k) "v_mad_i32_i24 %" #k ", %8, %9, %" #k
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %8, %" #k
#define I_CVT(k) "v_cvt_f32_i32 %" #k ", %" #k
#define I_RNDNE(k) "v_rndne_f32 %" #k ", %" #k
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %8, 16, %" #k
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %8, %9, %" #k
#define I_OR(k) "v_or_b32 %" #k ", %8, %" #k
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 9, %" #k
#define I_BFE(k) "v_bfe_u32 %" #k ", %8, 8, 8"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %9, %" #k
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %8, %9, 8"
#define I_ALIGNBYTE(k) "v_alignbyte_b32 %" #k ", %8, %9, 1"
#define I_SUB(k) "v_sub_u32 %" #k ", %8, %" #k
#define I_MAXI(k) "v_max_i32 %" #k ", %8, %" #k
#define I_MULF(k) "v_mul_f32 %" #k ", %8, %" #k
#define I_ADDF(k) "v_add_f32 %" #k ", %8, %" #k
#define I_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %8"
#define I_PACKF16(k) "v_pack_b32_f16 %" #k ", %8, %" #k
#define I_MULU24(k) "v_mul_u32_u24 %" #k ", %8, %" #k
#define I_ADD3(k) "v_add3_u32 %" #k ", %8, %9, %" #k
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %8, 7, %" #k
#define I_ADDSDWA(k) "v_add_u32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
#define I_XOR(k) "v_xor_b32 %" #k ", %8, %" #k
#define I_MOV(k) "v_mov_b32 %" #k ", %8"
#define I_FLOOR(k) "v_floor_f32 %" #k ", %" #k
#define I_MAD_U16(k) "v_mad_u16 %" #k ", %8, %9, %" #k
#define I_FMA64(k) "v_fma_f64 %" #k ", %8, %9, %" #k
#define I_MUL64(k) "v_mul_f64 %" #k ", %8, %" #k
#define I_ADD64(k) "v_add_f64 %" #k ", %8, %" #k
#define I_RSQ64(k) "v_rsq_f64 %" #k ", %" #k
#define I_RCP64(k) "v_rcp_f64 %" #k ", %" #k

template <int OP>
__global__ void probe(unsigned long long* out, int iters, unsigned seed) {
  const unsigned t = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (OP < 100) {
    unsigned a0 = t + seed, a1 = t * 3 + 1, a2 = t * 5 + 2, a3 = t * 7 + 3, a4 = t * 11, a5 = t * 13, a6 = t ^ 0x3333u, a7 = t + 99;
    unsigned e = t ^ 0x5555u, f = 0x01020304u + t;
    for (int i = 0; i < iters; i++) {
      if (OP == 0) SAME(I_FMA32)
      if (OP == 1) SAME(I_ADD)
      if (OP == 2) SAME(I_AND)
      if (OP == 3) SAME(I_LSHL)
      if (OP == 4) SAME(I_ASHR)
      if (OP == 5) SAME(I_DOT2C)
      if (OP == 6) SAME(I_DOT4)
      if (OP == 7) SAME(I_PERM)
      if (OP == 8) SAME(I_ADDDPP)
      if (OP == 9) SAME(I_MOVDPP)
      if (OP == 10) SAME(I_PKSUB)
      if (OP == 11) SAME(I_MAD24)
      if (OP == 12) SAME(I_MULLO)
      if (OP == 13) SAME(I_CVT)
      if (OP == 14) SAME(I_RNDNE)
      if (OP == 30) SAME(I_LSHLOR)
      if (OP == 31) SAME(I_ANDOR)
      if (OP == 32) SAME(I_OR)
      if (OP == 33) SAME(I_LSHR)
      if (OP == 34) SAME(I_BFE)
      if (OP == 35) SAME(I_BFI)
      if (OP == 36) SAME(I_ALIGNBIT)
      if (OP == 37) SAME(I_ALIGNBYTE)
      if (OP == 38) SAME(I_SUB)
      if (OP == 39) SAME(I_MAXI)
      if (OP == 40) SAME(I_MULF)
      if (OP == 41) SAME(I_ADDF)
      if (OP == 42) SAME(I_CVTUB)
      if (OP == 43) SAME(I_PACKF16)
      if (OP == 44) SAME(I_MULU24)
      if (OP == 45) SAME(I_ADD3)
      if (OP == 46) SAME(I_LSHLADD)
      if (OP == 47) SAME(I_ADDSDWA)
      if (OP == 48) SAME(I_XOR)
      if (OP == 49) SAME(I_MOV)
      if (OP == 50) SAME(I_FLOOR)
      if (OP == 51) SAME(I_MAD_U16)
      // the inner loop of k_klt_track per pixel pair: 3 byte gathers / packs, 6 dot products, 2 shifts, 1 packed subtract
      if (OP == 20) BODY(I_PERM(0), I_DOT2C(1), I_DOT2C(2), I_ASHR(1), I_PERM(3), I_DOT2C(4), I_DOT2C(5), I_ASHR(4))
      if (OP == 21) BODY(I_PERM(0), I_DOT2C(1), I_DOT2C(2), I_PKSUB(3), I_PERM(4), I_DOT2C(5), I_DOT2C(6), I_ASHR(7))
      // half fast-class, half slow-class
      if (OP == 22) BODY(I_ADD(0), I_DOT2C(1), I_ADD(2), I_DOT2C(3), I_ADD(4), I_DOT2C(5), I_ADD(6), I_DOT2C(7))
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345678u) out[1 << 20] = a0;
  } else {
    double a0 = t + seed, a1 = t * 3 + 1, a2 = t * 5 + 2, a3 = t * 7 + 3, a4 = t * 11 + 1, a5 = t * 13 + 1, a6 = t + 0.5, a7 = t + 99;
    double e = 1.0 + 1e-9 * t, f = 1e-12;
    for (int i = 0; i < iters; i++) {
      if (OP == 100) SAME(I_FMA64)
      if (OP == 101) SAME(I_MUL64)
      if (OP == 102) SAME(I_ADD64)
      if (OP == 103) SAME(I_RSQ64)
      if (OP == 104) SAME(I_RCP64)
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.12345) out[1 << 20] = 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((t & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + t / 64;
    out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
  }
}

template <int OP>
static void run(const char* name, unsigned long long* d) {
  const int iters = 4000, per_iter = 64;
  printf("%-22s", name);
  for (int w : {1, 2, 4, 8}) {
    const int wg = (w == 8) ? 2 : 1, threads = 64 * 4 * w / wg;     // w waves on each SIMD of every CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(256 * wg), dim3(threads), 0, 0, d, 50, 1u);      // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<OP>, dim3(256 * wg), dim3(threads), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int nw = 256 * 4 * w;
    std::vector<unsigned long long> h(2 * nw);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; i++) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double mhz = clk[nw / 2];
    const double inst = (double)iters * per_iter;
    // per SIMD: W waves x inst instructions in (ms) of wall time at mhz
    const double cps_wall = (ms * 1e-3 * mhz * 1e6) / (inst * w), cps_wave = cyc[nw / 2] / inst / w;
    printf(" | W=%d %.2f (%.2f) @%4.0f", w, cps_wall, cps_wave, mhz);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  printf("\n");
}

int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * ((1 << 20) + 64));
  printf("cycles per wave-instruction per SIMD: from the launch's wall time (from the median wave's own s_memtime span) @ in-kernel MHz\n");
  run<0>("v_fma_f32", d);
  run<1>("v_add_u32", d);
  run<2>("v_and_b32", d);
  run<3>("v_lshlrev_b32", d);
  run<4>("v_ashrrev_i32", d);
  run<5>("v_dot2c_i32_i16", d);
  run<6>("v_dot4_u32_u8", d);
  run<7>("v_perm_b32", d);
  run<8>("v_add_u32 dpp row_ror", d);
  run<9>("v_mov_b32 dpp quad", d);
  run<10>("v_pk_sub_i16", d);
  run<11>("v_mad_i32_i24", d);
  run<12>("v_mul_lo_u32", d);
  run<13>("v_cvt_f32_i32", d);
  run<14>("v_rndne_f32", d);
  run<30>("v_lshl_or_b32", d); run<31>("v_and_or_b32", d); run<32>("v_or_b32", d); run<33>("v_lshrrev_b32", d); run<34>("v_bfe_u32", d);
  run<35>("v_bfi_b32", d); run<36>("v_alignbit_b32", d); run<37>("v_alignbyte_b32", d); run<38>("v_sub_u32", d); run<39>("v_max_i32", d);
  run<40>("v_mul_f32", d); run<41>("v_add_f32", d); run<42>("v_cvt_f32_ubyte1", d); run<43>("v_pack_b32_f16", d); run<44>("v_mul_u32_u24", d);
  run<45>("v_add3_u32", d); run<46>("v_lshl_add_u32", d); run<47>("v_add_u32 sdwa byte", d); run<48>("v_xor_b32", d); run<49>("v_mov_b32", d);
  run<50>("v_floor_f32", d); run<51>("v_mad_u16", d);
  run<20>("klt mix A (sample2)", d);
  run<21>("klt mix B (+pk_sub)", d);
  run<22>("add/dot2c alternating", d);
  run<100>("v_fma_f64", d);
  run<101>("v_mul_f64", d);
  run<102>("v_add_f64", d);
  run<103>("v_rsq_f64", d);
  run<104>("v_rcp_f64", d);
  return 0;
}
