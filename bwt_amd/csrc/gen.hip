// Seeded synthetic genomes and reads generated straight into HBM
// (counter-based splitmix64; bit-identical to bwt_amd/synth.py).
#include "common.hpp"

namespace bwtgpu {

__host__ __device__ __forceinline__ uint64_t rnd64(uint64_t seed, uint64_t k) {
  uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t text_base(const uint8_t *t, uint64_t i) {
  return (t[i >> 2] >> (2 * (3 - (i & 3)))) & 3u;
}

// byte j of the packed text = byte (j & 7) of rnd64(seed, j >> 3); the byte
// that holds base n keeps only its first n % 4 bases
__global__ void k_gen_genome(uint8_t *__restrict__ text, uint64_t n, uint64_t seed) {
  const uint64_t nb = n / 4 + 1;
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t j0 = w * 8;
  if (j0 >= nb) return;
  const uint64_t z = rnd64(seed, w);
  const uint32_t keep[4] = {0x00u, 0xC0u, 0xF0u, 0xFCu};
  for (int k = 0; k < 8; ++k) {
    uint64_t j = j0 + k;
    if (j >= nb) break;
    uint8_t b = (uint8_t)(z >> (8 * k));
    if (j == nb - 1) b &= (uint8_t)keep[n & 3];
    text[j] = b;
  }
}

// reads [first, first + R) of the set that `seed` names, written from reads[0] on
__global__ void k_gen_reads(const uint8_t *__restrict__ text, uint64_t n, uint8_t *__restrict__ reads,
                            uint64_t first, uint64_t R, uint32_t m, uint64_t seed,
                            uint32_t *__restrict__ offsets) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const uint64_t r = first + i;
  uint8_t *out = reads + i * m;
  const uint64_t words = (m + 31) / 32;
  if (rnd64(seed, 2 * r + 1) % 100 == 0) {
    for (uint32_t k = 0; k < m; ++k)
      out[k] = (uint8_t)((rnd64(seed ^ 0xABCDull, r * words + k / 32) >> (2 * (k % 32))) & 3u);
    if (offsets) offsets[i] = 0xFFFFFFFFu;
  } else {
    const uint64_t j = rnd64(seed, 2 * r) % (n - m + 1);
    for (uint32_t k = 0; k < m; ++k) out[k] = (uint8_t)text_base(text, j + k);
    if (offsets) offsets[i] = (uint32_t)j;
  }
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

int bwtgpu_gen_genome(uint8_t *d_text, uint64_t n, uint64_t seed, void *stream) {
  if (!d_text || n == 0) return set_error(kErrArg, "bad genome request");
  const uint64_t words = div_up(n / 4 + 1, 8);
  BWT_LAUNCH("gen_genome", k_gen_genome, dim3((unsigned)div_up(words, 256)), dim3(256), 0,
             (hipStream_t)stream, d_text, n, seed);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_gen_reads_slice(const uint8_t *d_text, uint64_t n, uint8_t *d_reads, uint64_t first, uint64_t R,
                           uint32_t m, uint64_t seed, uint32_t *d_offsets, void *stream) {
  if (!d_text || !d_reads || m == 0 || m > n) return set_error(kErrArg, "bad reads request");
  if (R == 0) return kOk;
  BWT_LAUNCH("gen_reads", k_gen_reads, dim3((unsigned)div_up(R, 256)), dim3(256), 0,
             (hipStream_t)stream, d_text, n, d_reads, first, R, m, seed, d_offsets);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_gen_reads(const uint8_t *d_text, uint64_t n, uint8_t *d_reads, uint64_t R, uint32_t m,
                     uint64_t seed, uint32_t *d_offsets, void *stream) {
  return bwtgpu_gen_reads_slice(d_text, n, d_reads, 0, R, m, seed, d_offsets, stream);
}

}  // extern "C"
