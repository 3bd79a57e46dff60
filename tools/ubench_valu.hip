// Microbenchmark: issue rate of plain vs packed fp32 VALU ops and broadcast LDS reads on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 m = {1.0000001f, 0.9999999f};
  __shared__ float4 lds[1024];
  if (MODE >= 3) { for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(seed, 1, 2, 3); __syncthreads(); }
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));
    } else if (MODE == 1) { // 4 independent v_pk_mul_f32 (same flops as mode 0)
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m));
    } else if (MODE == 2) { // 8 v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));
    } else if (MODE == 3) { // 4 broadcast ds_read_b128 (uniform address)
      const float4 *p = lds + ((i * 4) & 1020);
      float4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
      acc.x += x0.x; acc.y += x1.y; acc.z += x2.z; acc.w += x3.w;
    } else if (MODE == 4) { // 4 broadcast ds_read_b128 + 32 v_mul
      const float4 *p = lds + ((i * 4) & 1020);
      float4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
      for (int r = 0; r < 4; ++r)
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));
      acc.x += x0.x; acc.y += x1.y; acc.z += x2.z; acc.w += x3.w;
    } else if (MODE == 5) { // 8 v_mul with an SGPR operand
      asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                   "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
    } else if (MODE == 6) { // 4 v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m));
    } else if (MODE == 7) { // 2 v_rcp_f32 + 6 v_mul
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + acc.x + acc.y + acc.z + acc.w;
}

template <int MODE> void run(const char *name, int wg_per_cu, double flop_per_iter_lane, float *d)
{
  int iters = 20000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 1000, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double lanes = double(blocks) * 256;
  // wave-instructions per SIMD: blocks*4 waves over 1024 SIMDs
  double waves_per_simd = double(blocks) * 4 / 1024;
  double ns_per_iter_per_wave = ms * 1e6 / iters / waves_per_simd;
  printf("%-44s wg/cu=%d  %.3f ms  %.2f ns per loop body per SIMD  %.1f TFLOP/s\n", name, wg_per_cu, ms, ns_per_iter_per_wave,
         flop_per_iter_lane * lanes * iters / (ms * 1e-3) / 1e12);
}

int main()
{
  float *d; hipMalloc(&d, 256 * 16 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    if (w == 1) {
    run<0>("8x v_mul_f32", 1, 8, d); run<1>("4x v_pk_mul_f32", 1, 8, d); run<2>("8x v_fma_f32", 1, 16, d);
    run<3>("4x ds_read_b128 broadcast", 1, 0, d); run<4>("4x ds_read_b128 + 32 v_mul", 1, 32, d); run<5>("8x v_mul sgpr", 1, 8, d);
    run<6>("4x v_pk_add_f32", 1, 8, d); run<7>("2 rcp + 6 mul", 1, 8, d); }
    if (w == 2) {
    run<0>("8x v_mul_f32", 2, 8, d); run<1>("4x v_pk_mul_f32", 2, 8, d); run<2>("8x v_fma_f32", 2, 16, d);
    run<3>("4x ds_read_b128 broadcast", 2, 0, d); run<4>("4x ds_read_b128 + 32 v_mul", 2, 32, d); run<5>("8x v_mul sgpr", 2, 8, d);
    run<6>("4x v_pk_add_f32", 2, 8, d); run<7>("2 rcp + 6 mul", 2, 8, d); }
    if (w == 4) {
    run<0>("8x v_mul_f32", 4, 8, d); run<1>("4x v_pk_mul_f32", 4, 8, d); run<2>("8x v_fma_f32", 4, 16, d);
    run<3>("4x ds_read_b128 broadcast", 4, 0, d); run<4>("4x ds_read_b128 + 32 v_mul", 4, 32, d); run<5>("8x v_mul sgpr", 4, 8, d);
    run<6>("4x v_pk_add_f32", 4, 8, d); run<7>("2 rcp + 6 mul", 4, 8, d); }
    if (w == 8) {
    run<0>("8x v_mul_f32", 8, 8, d); run<1>("4x v_pk_mul_f32", 8, 8, d); run<3>("4x ds_read_b128 broadcast", 8, 0, d);
    run<4>("4x ds_read_b128 + 32 v_mul", 8, 32, d); }
  }
  return 0;
}
