// Microbenchmark: what ONE wave alone on its SIMD pays per instruction on gfx950 -- the situation of the bounce
// kernel's tail (mrt_bounce.hip: a few hundred waves, one per SIMD, ~1500 mostly dependent instructions per loop
// iteration).  Shader-clock cycles (s_memtime) per operation for dependent chains of length 1, 2 and 4, the IEEE
// division and square root expansions the exact tests need, a broadcast LDS read, a ballot + scalar branch.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/bin/ubench_latency tools/ubench_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
  float a = seed + threadIdx.x * 1e-3f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
  const float m = 1.0000001f;
  __shared__ float4 lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = make_float4(float((i * 7 + 1) & 255), 1, 2, 3);
  __syncthreads();
  int idx = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { // 8 dependent v_mul_f32
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    } else if (MODE == 1) { // 2 chains x 4
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));
    } else if (MODE == 2) { // 4 chains x 2
#pragma unroll
      for (int r = 0; r < 2; ++r)
        asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    } else if (MODE == 3) { // 8 dependent IEEE divisions
#pragma unroll
      for (int r = 0; r < 8; ++r) a = b / a;
    } else if (MODE == 4) { // 8 divisions in 2 independent chains
#pragma unroll
      for (int r = 0; r < 4; ++r) { a = c / a; b = d / b; }
    } else if (MODE == 5) { // 8 dependent IEEE square roots
#pragma unroll
      for (int r = 0; r < 8; ++r) a = sqrtf(a + 2.0f);
    } else if (MODE == 6) { // 8 dependent broadcast ds_read_b128 (the address comes from the value read)
#pragma unroll
      for (int r = 0; r < 8; ++r) { const float4 v = lds[idx & 255]; idx = (int)v.x; a += v.y; }
    } else if (MODE == 7) { // 8 x (compare, ballot, scalar branch on it, v_mul)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (__builtin_amdgcn_ballot_w64(a > 1e30f)) a = a * 0.5f;
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
      }
    } else if (MODE == 8) { // 8 dependent v_rcp_f32
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    } else if (MODE == 9) { // 8 x dependent dot3 without contraction (5 ops: 3 mul, 2 add; critical path 3)
#pragma unroll
      for (int r = 0; r < 8; ++r) a = a * b + (c * a + d * a);
    } else if (MODE == 10) { // 8 x readlane broadcast feeding a v_mul
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 5));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "s"(s));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + float(idx);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, float *d, unsigned long long *c, int waves)
{
  const int iters = 20000;
  k<MODE><<<waves, 64>>>(d, c, 1000, 1.0f);
  hipDeviceSynchronize();
  k<MODE><<<waves, 64>>>(d, c, iters, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[4096];
  hipMemcpy(h, c, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < waves; ++i) s += double(h[i]);
  // s_memtime ticks at 100 MHz on gfx9xx: report ns per op as well, which does not depend on the tick
  printf("%-58s waves=%5d  %8.2f ticks per op (s_memtime)\n", name, waves, s / waves / iters / 8.0);
}

int main()
{
  float *d; unsigned long long *c;
  hipMalloc(&d, 4096 * 64 * 4); hipMalloc(&c, 4096 * 8);
  for (int waves : {1, 1024, 4096}) { // one wave on the GPU; one per SIMD; four per SIMD
    run<0>("v_mul_f32, one dependent chain", d, c, waves);
    run<1>("v_mul_f32, two chains", d, c, waves);
    run<2>("v_mul_f32, four chains", d, c, waves);
    run<9>("dependent dot3-like (5 ops, path 3)", d, c, waves);
    run<8>("v_rcp_f32 dependent", d, c, waves);
    run<3>("IEEE division, dependent", d, c, waves);
    run<4>("IEEE division, two chains", d, c, waves);
    run<5>("IEEE sqrt, dependent", d, c, waves);
    run<6>("broadcast ds_read_b128, dependent address", d, c, waves);
    run<7>("compare + ballot + scalar branch + v_mul", d, c, waves);
    run<10>("readlane -> sgpr operand of v_mul", d, c, waves);
  }
  return 0;
}
