// ubench_launch.hip -- what a frame-sized grid costs before it computes anything (diagnostic, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_launch tools/ubench_launch.hip && /tmp/ubench_launch
// Cases: empty kernels of 16200 / 8100 / 4050 workgroups of 1 / 2 / 4 waves; the same with a 4 KB argument block;
// kernels that only write a 1920x1080 frame of 20 B per pixel (12 + 4 + 4, like the render kernels) with different
// pixels per wave, plain and non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Big { float f[960]; };

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void empty_big_kernel(Big b, int *p) { if (p && threadIdx.x == 12345) *p = (int)b.f[5]; }

// every wave reads 12 x 64 floats (lanes below `live` only) from the argument block or from a device buffer and ends
__global__ void read_args_kernel(Big b, int live, float *out)
{
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (lane < live)
#pragma unroll
    for (int i = 0; i < 12; ++i) acc += b.f[i * 64 + lane];
  if (acc == 12345.f) *out = acc;
}
__global__ void read_buf_kernel(const float *__restrict__ b, int live, float *out)
{
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (lane < live)
#pragma unroll
    for (int i = 0; i < 12; ++i) acc += b[i * 64 + lane];
  if (acc == 12345.f) *out = acc;
}

typedef float nfloat2 __attribute__((ext_vector_type(2)));
typedef float nfloat4 __attribute__((ext_vector_type(4)));

// a wave writes a tile of (8 * P * T) x 8 pixels: lane (tx, ty) owns P adjacent pixels in each of T tiles side by side
// (LX lanes side by side, 64 / LX rows: LX = 8 is the render kernels' shape, LX = 16 writes 128-byte runs of id and t)
template <int P, bool NT, int LX = 8>
__global__ __launch_bounds__(256) void store_kernel(float *rgb, int *id, float *t, int W, int H, int T)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tx = lane & (LX - 1), ty = lane / LX;
  const int row = blockIdx.y * (64 / LX) + ty;
  if (row >= H) return;
  for (int j = 0; j < T; ++j) {
    const int col0 = ((blockIdx.x * (blockDim.x >> 6) + wave) * T + j) * LX * P + tx * P;
    if (col0 + P > W) return;
    const size_t pix = (size_t)row * W + col0;
    float v = (float)(row + col0) * 1e-6f;
    if constexpr (P == 2) {
      nfloat2 a = {v, v + 1.f}, b = {v + 2.f, v}, c = {v + 1.f, v + 2.f};
      nfloat2 *d = reinterpret_cast<nfloat2 *>(rgb + 3 * pix);
      if (NT) { __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 1); __builtin_nontemporal_store(c, d + 2);
                __builtin_nontemporal_store(a, reinterpret_cast<nfloat2 *>(id + pix)); __builtin_nontemporal_store(b, reinterpret_cast<nfloat2 *>(t + pix)); }
      else { d[0] = a; d[1] = b; d[2] = c; *reinterpret_cast<nfloat2 *>(id + pix) = a; *reinterpret_cast<nfloat2 *>(t + pix) = b; }
    } else if constexpr (P == 4) {
      nfloat4 a = {v, v + 1.f, v + 2.f, v};
      nfloat4 *d = reinterpret_cast<nfloat4 *>(rgb + 3 * pix);
      if (NT) { __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(a, d + 1); __builtin_nontemporal_store(a, d + 2);
                __builtin_nontemporal_store(a, reinterpret_cast<nfloat4 *>(id + pix)); __builtin_nontemporal_store(a, reinterpret_cast<nfloat4 *>(t + pix)); }
      else { d[0] = a; d[1] = a; d[2] = a; *reinterpret_cast<nfloat4 *>(id + pix) = a; *reinterpret_cast<nfloat4 *>(t + pix) = a; }
    } else {
      if (NT) { __builtin_nontemporal_store(v, rgb + 3 * pix); __builtin_nontemporal_store(v, rgb + 3 * pix + 1); __builtin_nontemporal_store(v, rgb + 3 * pix + 2);
                __builtin_nontemporal_store(v, reinterpret_cast<float *>(id + pix)); __builtin_nontemporal_store(v, t + pix); }
      else { rgb[3 * pix] = v; rgb[3 * pix + 1] = v; rgb[3 * pix + 2] = v; id[pix] = (int)v; t[pix] = v; }
    }
  }
}


// the render kernels' tile (8 x 8 lanes, 2 pixels per lane) written with each cache-policy combination of
// global_store_dwordx2 (vector stores; MODE: 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 nt, 6 sc1 nt, 7 sc0 sc1 nt).
// Measured: plain 9.5 us, sc0 9.5, nt 19.7, sc1 18.6, sc0 sc1 18.6, sc0 nt 20.6, sc1 nt 35.0, sc0 sc1 nt 35.0 -- no
// combination beats the two the render kernel has tried (plain, nt).
template <int MODE>
__device__ __forceinline__ void st2(void *p, nfloat2 v)
{
  if constexpr (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(64) void store_mode_kernel(float *rgb, int *id, float *t, int W, int H)
{
  const int lane = threadIdx.x & 63;
  const int tx = lane & 7, ty = lane >> 3;
  const int row = blockIdx.y * 8 + ty, col0 = blockIdx.x * 16 + tx * 2;
  if (row >= H || col0 + 2 > W) return;
  const size_t pix = (size_t)row * W + col0;
  const float v = (float)(row + col0) * 1e-6f;
  nfloat2 a = {v, v + 1.f}, b = {v + 2.f, v}, c = {v + 1.f, v + 2.f};
  st2<MODE>(rgb + 3 * pix, a); st2<MODE>(rgb + 3 * pix + 2, b); st2<MODE>(rgb + 3 * pix + 4, c);
  st2<MODE>(id + pix, a); st2<MODE>(t + pix, b);
}

// ---- round 3: what bounds the frame's stores? -----------------------------------------------------------------
// (1) the write roof: the same 41.5 MB written as one contiguous run, 16 bytes per lane and instruction;
typedef float nfloat3 __attribute__((ext_vector_type(3)));
template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(nfloat4 *dst, size_t n4, int per_thread)
{
  size_t i = (size_t)blockIdx.x * blockDim.x * per_thread + threadIdx.x;
  const nfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (int k = 0; k < per_thread; ++k, i += blockDim.x)
    if (i < n4) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}
// (2) the render kernel's tile (8 x 8 lanes, 2 pixels per lane, one wave per workgroup) with the lane's 24 contiguous
// RGB bytes written as SHAPE 0: three dwordx2 (the kernel's), 1: dwordx4 + dwordx2, 2: two dwordx3,
// 3: through LDS: the tile's 8 rows of 192 bytes go out as 96 dwordx4 (two instructions, each lane a 16-byte piece of a
// row: whole 64-byte segments per instruction and row), id and t as they are
template <int SHAPE, bool NT>
__global__ __launch_bounds__(64) void store_shape_kernel(float *rgb, int *id, float *t, int W, int H)
{
  const int lane = threadIdx.x & 63;
  const int tx = lane & 7, ty = lane >> 3;
  const int row0 = blockIdx.y * 8, col00 = blockIdx.x * 16;
  const int row = row0 + ty, col0 = col00 + tx * 2;
  if (row0 + 8 > H || col00 + 16 > W) return; // whole tiles only (1920 x 1080 divides)
  const size_t pix = (size_t)row * W + col0;
  const float v = (float)(row + col0) * 1e-6f;
  const float o[6] = {v, v + 1.f, v + 2.f, v + 3.f, v + 4.f, v + 5.f};
  float *d = rgb + 3 * pix;
  auto st = [&](auto val, auto *ptr) { if (NT) __builtin_nontemporal_store(val, ptr); else *ptr = val; };
  if constexpr (SHAPE == 0) {
    st(nfloat2{o[0], o[1]}, reinterpret_cast<nfloat2 *>(d)); st(nfloat2{o[2], o[3]}, reinterpret_cast<nfloat2 *>(d) + 1);
    st(nfloat2{o[4], o[5]}, reinterpret_cast<nfloat2 *>(d) + 2);
  } else if constexpr (SHAPE == 1) { // 24-byte stride: every other lane's dwordx4 sits on an 8-byte boundary only
    typedef float u4 __attribute__((ext_vector_type(4), aligned(8)));
    st(u4{o[0], o[1], o[2], o[3]}, reinterpret_cast<u4 *>(d)); st(nfloat2{o[4], o[5]}, reinterpret_cast<nfloat2 *>(d + 4));
  } else if constexpr (SHAPE == 2) {
    typedef float u3 __attribute__((ext_vector_type(3), aligned(4)));
    st(u3{o[0], o[1], o[2]}, reinterpret_cast<u3 *>(d)); st(u3{o[3], o[4], o[5]}, reinterpret_cast<u3 *>(d + 3));
  } else {
    __shared__ float tile[8 * 48];
#pragma unroll
    for (int k = 0; k < 6; ++k) tile[ty * 48 + tx * 6 + k] = o[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int piece = h * 64 + lane; // 96 pieces of 16 bytes: 12 per row
      if (piece < 96) {
        const int r = piece / 12, c = piece - r * 12;
        const nfloat4 val = *reinterpret_cast<const nfloat4 *>(&tile[r * 48 + c * 4]);
        st(val, reinterpret_cast<nfloat4 *>(rgb + 3 * ((size_t)(row0 + r) * W + col00) + c * 4));
      }
    }
  }
  st(nfloat2{v, v}, reinterpret_cast<nfloat2 *>(id + pix)); st(nfloat2{v, v}, reinterpret_cast<nfloat2 *>(t + pix));
}

template <class F> static float time_us(F launch, int n = 400)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 20; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms * 1000.f / n < best ? ms * 1000.f / n : best;
  }
  return best;
}

int main()
{
  const int W = 1920, H = 1080;
  float *rgb, *t; int *id;
  CK(hipMalloc(&rgb, (size_t)W * H * 12)); CK(hipMalloc(&id, (size_t)W * H * 4)); CK(hipMalloc(&t, (size_t)W * H * 4));
  Big big; for (int i = 0; i < 960; ++i) big.f[i] = (float)i;
  printf("empty kernels, total waves 16200 (120 x 135 tiles):\n");
  for (int wpw : {1, 2, 4}) {
    dim3 g(120 / wpw, 135);
    printf("  %d wave(s) per workgroup, %5d workgroups: %.2f us;  with 3.8 KB of arguments: %.2f us\n", wpw, g.x * g.y,
           time_us([&] { hipLaunchKernelGGL(empty_kernel, g, dim3(64 * wpw), 0, 0, (int *)nullptr); }),
           time_us([&] { hipLaunchKernelGGL(empty_big_kernel, g, dim3(64 * wpw), 0, 0, big, (int *)nullptr); }));
  }
  printf("empty kernels, 1 wave per workgroup, by grid size:\n");
  for (int n : {1, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536})
    printf("  %6d workgroups: %.2f us\n", n, time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(n), dim3(64), 0, 0, (int *)nullptr); }));
  printf("16200 one-wave workgroups that read 12 floats per lane and end:\n");
  {
    float *buf; CK(hipMalloc(&buf, sizeof(Big))); CK(hipMemcpy(buf, &big, sizeof(Big), hipMemcpyHostToDevice));
    dim3 g(120, 135);
    for (int live : {10, 64})
      printf("  %2d lanes live: from the argument block %.2f us, from a device buffer %.2f us\n", live,
             time_us([&] { hipLaunchKernelGGL(read_args_kernel, g, dim3(64), 0, 0, big, live, (float *)rgb); }),
             time_us([&] { hipLaunchKernelGGL(read_buf_kernel, g, dim3(64), 0, 0, (const float *)buf, live, (float *)rgb); }));
  }
  printf("store-only kernels (41.5 MB per launch):\n");
#define CASE(P, NT, WPW, T) { dim3 g((W / (8 * P * T) + WPW - 1) / WPW, 135); \
    float us = time_us([&] { hipLaunchKernelGGL((store_kernel<P, NT>), g, dim3(64 * WPW), 0, 0, rgb, id, t, W, H, T); }); \
    printf("  P=%d tiles/wave=%d waves/wg=%d nt=%d (%5d waves): %.2f us = %.0f GB/s\n", P, T, WPW, (int)NT, g.x * g.y * WPW, us, W * H * 20.0 / us * 1e-3); }
  CASE(1, false, 1, 1) CASE(1, true, 1, 1) CASE(1, true, 4, 1)
  CASE(2, false, 1, 1) CASE(2, true, 1, 1) CASE(2, true, 2, 1) CASE(2, true, 4, 1)
  CASE(2, true, 1, 2) CASE(2, true, 1, 4)
  CASE(4, false, 1, 1) CASE(4, true, 1, 1) CASE(4, true, 4, 1) CASE(4, true, 1, 2) CASE(4, true, 1, 4)
  printf("the write roof: 41.5 MB as one contiguous run, dwordx4 per lane:\n");
  {
    const size_t n4 = (size_t)W * H * 20 / 16;
    float *lin; CK(hipMalloc(&lin, n4 * 16));
    for (int per : {1, 4, 16}) {
      const unsigned blocks = (unsigned)((n4 + 256ull * per - 1) / (256ull * per));
      float a = time_us([&] { hipLaunchKernelGGL((fill_kernel<false>), dim3(blocks), dim3(256), 0, 0, (nfloat4 *)lin, n4, per); });
      float b = time_us([&] { hipLaunchKernelGGL((fill_kernel<true>), dim3(blocks), dim3(256), 0, 0, (nfloat4 *)lin, n4, per); });
      printf("  %2d x 16 B per thread, %6u workgroups of 256: plain %.2f us = %.0f GB/s, nt %.2f us = %.0f GB/s\n", per, blocks, a,
             W * H * 20.0 / a * 1e-3, b, W * H * 20.0 / b * 1e-3);
    }
    CK(hipFree(lin));
  }
  printf("16 x 8-pixel tiles, one wave per workgroup, by the shape of the lane's RGB stores:\n");
  {
    const char *names[4] = {"3 x dwordx2 (the kernel's)", "dwordx4 + dwordx2", "2 x dwordx3", "rows through LDS, dwordx4"};
    dim3 g(W / 16, 135);
#define SHAPE_CASE(S) { float a = time_us([&] { hipLaunchKernelGGL((store_shape_kernel<S, false>), g, dim3(64), 0, 0, rgb, id, t, W, H); }); \
                        float b = time_us([&] { hipLaunchKernelGGL((store_shape_kernel<S, true>), g, dim3(64), 0, 0, rgb, id, t, W, H); }); \
                        printf("  %-28s plain %.2f us = %.0f GB/s, nt %.2f us = %.0f GB/s\n", names[S], a, W * H * 20.0 / a * 1e-3, b, W * H * 20.0 / b * 1e-3); }
    SHAPE_CASE(0) SHAPE_CASE(1) SHAPE_CASE(2) SHAPE_CASE(3)
  }
  printf("16 x 8-pixel tiles, dwordx2 stores, by cache policy:\n");
  {
    const char *names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    dim3 g(W / 16, 135);
#define MODE_CASE(M) printf("  %-12s %.2f us\n", names[M], time_us([&] { hipLaunchKernelGGL((store_mode_kernel<M>), g, dim3(64), 0, 0, rgb, id, t, W, H); }));
    MODE_CASE(0) MODE_CASE(1) MODE_CASE(2) MODE_CASE(3) MODE_CASE(4) MODE_CASE(5) MODE_CASE(6) MODE_CASE(7)
  }
  printf("the same with tiles of 16 lanes x 4 rows (32 x 4 pixels at P=2: full 128-byte lines of id and t):\n");
#define CASEW(P, NT, WPW, T) { dim3 g((W / (16 * P * T) + WPW - 1) / WPW, H / 4); \
    float us = time_us([&] { hipLaunchKernelGGL((store_kernel<P, NT, 16>), g, dim3(64 * WPW), 0, 0, rgb, id, t, W, H, T); }); \
    printf("  P=%d tiles/wave=%d waves/wg=%d nt=%d (%5d waves): %.2f us = %.0f GB/s\n", P, T, WPW, (int)NT, g.x * g.y * WPW, us, W * H * 20.0 / us * 1e-3); }
  CASEW(1, false, 1, 1) CASEW(1, true, 1, 1) CASEW(2, false, 1, 1) CASEW(2, true, 1, 1) CASEW(2, true, 1, 2) CASEW(4, true, 1, 1)
  printf("... and 32 lanes x 2 rows:\n");
#define CASEV(P, NT) { dim3 g(W / (32 * P), H / 2); \
    float us = time_us([&] { hipLaunchKernelGGL((store_kernel<P, NT, 32>), g, dim3(64), 0, 0, rgb, id, t, W, H, 1); }); \
    printf("  P=%d nt=%d (%5d waves): %.2f us = %.0f GB/s\n", P, (int)NT, g.x * g.y, us, W * H * 20.0 / us * 1e-3); }
  CASEV(1, false) CASEV(1, true) CASEV(2, false) CASEV(2, true)
  return 0;
}
