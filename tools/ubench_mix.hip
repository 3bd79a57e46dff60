// Microbenchmark: issue cost of the vector instructions the render kernels are made of, per SIMD, on gfx950.
// The fp32 vector peak (157 TFLOP/s) is quoted on packed FMAs; tools/ubench_valu.hip showed a plain v_mul_f32 going
// out every ~2 cycles per SIMD.  The kernels here are not made of multiplies alone: selects, compares, integer
// address arithmetic, lane reads, the pieces of IEEE division and square root.  This measures each kind the same way:
// 16 independent instructions of one kind per loop body, 4 / 8 waves per SIMD, time per instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_mix tools/ubench_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed, int sel)
{
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
  const float m = 1.0000001f;
  int lanesel = sel & 63;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 1) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 2) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 3) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : );
      REP16(OP)
#undef OP
    } else if (MODE == 4) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      REP16(OP)
#undef OP
    } else if (MODE == 5) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 6) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 7) {
      int s0;
#define OP(i) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s0) : "v"(a[i]), "s"(lanesel));
      REP16(OP)
#undef OP
    } else if (MODE == 8) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 9) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP16(OP)
#undef OP
    } else if (MODE == 10) {
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      REP16(OP)
#undef OP
    } else if (MODE == 11) {
#define OP(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(m) : "vcc");
      REP16(OP)
#undef OP
    } else if (MODE == 12) {
#define OP(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m) : "vcc");
      REP16(OP)
#undef OP
    } else if (MODE == 13) {
#define OP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 14) {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      REP16(OP)
#undef OP
    } else if (MODE == 15) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 16) {
#define OP(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
      REP16(OP)
#undef OP
    } else if (MODE == 17) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 18) { // a select whose condition is a scalar pair other than vcc
      unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)sel;
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(msk));
      REP16(OP)
#undef OP
    } else if (MODE == 19) {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 20) {
#define OP(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
      REP16(OP)
#undef OP
    } else if (MODE == 21) {
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      REP16(OP)
#undef OP
    } else if (MODE == 22) { // a compare that writes a scalar pair (VOP3)
      unsigned long long msk;
#define OP(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(msk) : "v"(a[i]), "v"(m));
      REP16(OP)
#undef OP
    } else if (MODE == 23) {
#define OP(i) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(a[i]) : "s"(sel));
      REP16(OP)
#undef OP
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float *d;
template <int MODE> void run(const char *name)
{
  for (int wg_per_cu : {4, 8}) {
    const int iters = 4000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 200, 1.0f, 5); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1.0f, 5); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double waves_per_simd = double(blocks) * 4 / 1024;
    const double ns_per_inst = best * 1e6 / iters / 16 / waves_per_simd;
    printf("%-28s %d waves/SIMD  %.3f ns per instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, (int)waves_per_simd, ns_per_inst, ns_per_inst * 2.4);
  }
}

int main()
{
  hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w = 0; w < 30; ++w) k<0><<<2048, 256>>>(d, 4000, 1.0f, 5); // let the clocks settle
  hipDeviceSynchronize();
  run<0>("v_mul_f32"); run<1>("v_add_f32"); run<20>("v_sub_f32 (sgpr operand)"); run<2>("v_fma_f32"); run<8>("v_max_f32"); run<19>("v_max3_f32");
  run<3>("v_cndmask_b32 (vcc)"); run<18>("v_cndmask_b32 (sgpr pair)"); run<4>("v_cmp_lt_f32 -> vcc"); run<22>("v_cmp_lt_f32 -> sgpr pair");
  run<5>("v_add_u32"); run<17>("v_and_b32"); run<15>("v_mul_lo_u32"); run<6>("v_mov_b32"); run<16>("v_cvt_f32_i32");
  run<7>("v_readlane_b32"); run<23>("v_writelane_b32"); run<14>("v_mov_b32_dpp row_mirror");
  run<9>("v_rcp_f32"); run<21>("v_rsq_f32"); run<10>("v_sqrt_f32"); run<11>("v_div_scale_f32"); run<12>("v_div_fmas_f32"); run<13>("v_div_fixup_f32");
  run<0>("v_mul_f32 (again: drift)"); run<8>("v_max_f32 (again)"); run<2>("v_fma_f32 (again)");
  return 0;
}
