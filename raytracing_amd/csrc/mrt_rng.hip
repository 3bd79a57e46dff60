// mrt_rng.hip -- the per-pixel random streams of the diffuse bounce.
//
// The reference keeps one curandState (XORWOW) per pixel, set up once by
// curand_init(1984, pixel_index, 0, &state) (cuda_utils/curand_state.cuh:15-23, :33-45), and every render
// starts from a private COPY of it (src/main.cu:133-135): the stored states never advance.  Here that is one
// buffer of xorshift words in HBM, built by rng_init_kernel the first time a frame needs it and kept in the
// context; the render kernel (mrt_bounce.hip) loads a pixel's five words, draws, and never writes back.
//
// XORWOW as published in curand_kernel.h: five xorshift words v[0..4] plus a Weyl counter d (+362437 per
// draw), output v[4] + d; subsequence n begins n * 2^67 draws into the stream.  The xorshift step is linear
// over GF(2), so that jump is a 160x160 bit matrix applied to v, and d is left alone (362437 n 2^67 = 0 mod
// 2^32) -- d is therefore the same for every pixel and travels as a kernel argument, not in the buffer.
// The matrices A^(2^(67+b)), b = 0..31, are computed on the host by repeated squaring at first use (98
// squarings of a 160x160 bit matrix: a few ms) and uploaded once per context; pixel i applies the matrices of
// the set bits of i.  tests/ checks them against the oracle's own squaring and against rocRAND's precomputed
// table of the same recurrence.  Parity with cuRAND's actual output is unpinned (no CUDA in this image).

#include "mrt_internal.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace mrt {

namespace {

// row (w*32 + j) = image of the state whose only set bit is bit j of word w
struct BitMat { uint32_t row[160][5]; };

inline void step_words(uint32_t v[5]) // one xorshift step, without the Weyl counter
{
  const uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

void apply(const BitMat &m, const uint32_t in[5], uint32_t out[5])
{
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < 5; ++w)
    for (int j = 0; j < 32; ++j)
      if ((in[w] >> j) & 1u)
        for (int k = 0; k < 5; ++k) r[k] ^= m.row[w * 32 + j][k];
  std::memcpy(out, r, sizeof r);
}

void square(BitMat &m)
{
  BitMat sq;
  for (int r = 0; r < 160; ++r) apply(m, m.row[r], sq.row[r]);
  m = sq;
}

std::vector<BitMat> g_jump; // [RNG_JUMP_MATRICES]
std::once_flag g_jump_once;

void build_jump()
{
  g_jump.resize(RNG_JUMP_MATRICES);
  BitMat &a = g_jump[0];
  for (int w = 0; w < 5; ++w)
    for (int j = 0; j < 32; ++j) {
      uint32_t v[5] = {0, 0, 0, 0, 0};
      v[w] = 1u << j;
      step_words(v);
      std::memcpy(a.row[w * 32 + j], v, sizeof v);
    }
  for (int i = 0; i < 67; ++i) square(a); // A^(2^67): one subsequence
  for (int b = 1; b < RNG_JUMP_MATRICES; ++b) { g_jump[b] = g_jump[b - 1]; square(g_jump[b]); }
}

} // namespace

const uint32_t *rng_jump_tables()
{
  std::call_once(g_jump_once, build_jump);
  return &g_jump[0].row[0][0];
}

// curand_init's seeding (curand_kernel.h, _curand_init_scratch): subsequence 0, offset 0
void rng_seed_state(uint64_t seed, uint32_t v[5], uint32_t *d)
{
  const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
  *d = 6615241u + t1 + t0;
  v[0] = 123456789u + t0;
  v[1] = 362436069u ^ t0;
  v[2] = 521288629u + t1;
  v[3] = 88675123u ^ t1;
  v[4] = 5783321u + t0;
}

void rng_host_state(uint64_t seed, uint32_t subsequence, uint32_t out6[6])
{
  rng_seed_state(seed, out6, &out6[5]);
  (void)rng_jump_tables();
  for (int b = 0; b < RNG_JUMP_MATRICES; ++b)
    if ((subsequence >> b) & 1u) apply(g_jump[b], out6, out6);
}

// One lane per state.  The matrix rows come through wave-uniform (scalar) loads; a lane whose index lacks
// bit b walks along and keeps its old words.
__global__ __launch_bounds__(256) void rng_init_kernel(const uint32_t *__restrict__ jump, uint32_t *__restrict__ states,
                                                       uint32_t n, uint32_t stride, uint32_t v0, uint32_t v1,
                                                       uint32_t v2, uint32_t v3, uint32_t v4)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t idx = i < n ? i : 0u;
  uint32_t x0 = v0, x1 = v1, x2 = v2, x3 = v3, x4 = v4;
  for (int b = 0; b < RNG_JUMP_MATRICES; ++b) {
    const bool on = (idx >> b) & 1u;
    if (!__any(on)) continue;
    const uint32_t *m = jump + size_t(b) * 800;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
    auto word = [&](uint32_t xw, const uint32_t *rows) {
#pragma unroll 4
      for (int j = 0; j < 32; ++j) {
        const uint32_t mask = 0u - ((xw >> j) & 1u);
        const uint32_t *row = rows + 5 * j;
        r0 ^= mask & row[0]; r1 ^= mask & row[1]; r2 ^= mask & row[2]; r3 ^= mask & row[3]; r4 ^= mask & row[4];
      }
    };
    word(x0, m); word(x1, m + 160); word(x2, m + 320); word(x3, m + 480); word(x4, m + 640);
    if (on) { x0 = r0; x1 = r1; x2 = r2; x3 = r3; x4 = r4; }
  }
  if (i < n) {
    states[i] = x0;
    states[size_t(stride) + i] = x1;
    states[2 * size_t(stride) + i] = x2;
    states[3 * size_t(stride) + i] = x3;
    states[4 * size_t(stride) + i] = x4;
  }
}

hipError_t launch_rng_init(const uint32_t *jump_dev, uint32_t *states, uint32_t n, uint32_t stride, uint64_t seed,
                           hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  if (!jump_dev || !states || stride < n) return hipErrorInvalidValue;
  uint32_t v[5], d;
  rng_seed_state(seed, v, &d);
  hipLaunchKernelGGL(rng_init_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, jump_dev, states, n, stride, v[0],
                     v[1], v[2], v[3], v[4]);
  return hipGetLastError();
}

} // namespace mrt
