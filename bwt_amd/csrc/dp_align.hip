// nw_fast / sw_fast (smw.c:26-200) for a batch of (read piece, genome piece)
// pairs, one wave per pair: see dp_device.hpp.
#include "common.hpp"
#include "dp_device.hpp"

namespace bwtgpu {

__global__ void __launch_bounds__(64)
k_dp_align(const uint8_t *__restrict__ s1_all, const uint64_t *__restrict__ off1,
           const uint8_t *__restrict__ s2_all, const uint64_t *__restrict__ off2, int mode, uint32_t max1,
           uint32_t max2, DpLds L, int32_t *__restrict__ ret, uint8_t *__restrict__ ops_all,
           uint32_t *__restrict__ nops) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t *s1 = lds + L.total;  // the read piece, behind the wave's matrix area
  const uint64_t r = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint64_t a1 = off1[r], a2 = off2[r];
  const uint64_t l1 = off1[r + 1] - a1, l2 = off2[r + 1] - a2;
  if (l1 > max1 || l2 > max2) {  // not what the launch was sized for
    if (lane == 0) {
      ret[r] = 0;
      nops[r] = 0xFFFFFFFFu;
    }
    return;
  }
  const int len1 = (int)l1, len2 = (int)l2;
  for (int i = (int)lane; i < len1; i += 64) s1[i] = s1_all[a1 + i];
  __syncthreads();
  uint32_t k = 0;
  const uint8_t *s2 = s2_all + a2;
  const int rv = dp_wave([&](int i) -> uint32_t { return s1[i]; }, len1,
                         [&](int j) -> uint32_t { return s2[j]; }, len2, mode, L, lds, ops_all + a1 + a2, &k);
  if (lane == 0) {
    ret[r] = rv;
    nops[r] = k;
  }
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" int bwtgpu_dp_align_batch(const uint8_t *d_s1, const uint64_t *d_off1, const uint8_t *d_s2,
                                     const uint64_t *d_off2, uint64_t R, int mode, uint32_t max_len1,
                                     uint32_t max_len2, int32_t *d_ret, uint8_t *d_ops, uint32_t *d_nops,
                                     void *stream) {
  if (!d_s1 || !d_off1 || !d_s2 || !d_off2 || !d_ret || !d_ops || !d_nops)
    return set_error(kErrArg, "null argument");
  if (mode != 0 && mode != 1) return set_error(kErrArg, "mode must be 0 (nw_fast) or 1 (sw_fast)");
  if (R == 0) return kOk;
  if (R > 0x7FFFFFFFull) return set_error(kErrArg, "too many pairs for one launch");
  if (max_len1 > 65535u || max_len2 > 65535u) return set_error(kErrArg, "pieces of up to 65535 bases");
  const DpLds L = dp_lds_layout(max_len1, max_len2);
  const uint32_t lds_bytes = L.total + (uint32_t)align_up(max_len1 + 1, 4);
  if (lds_bytes > 64u * 1024u)
    return set_error(kErrArg, "pieces of %u x %u bases need %u bytes of LDS (64 KiB at most)", max_len1, max_len2,
                     lds_bytes);
  BWT_LAUNCH("dp_align", k_dp_align, dim3((unsigned)R), dim3(64), lds_bytes, (hipStream_t)stream, d_s1, d_off1,
             d_s2, d_off2, mode, max_len1, max_len2, L, d_ret, d_ops, d_nops);
  BWT_CHECK_LAUNCH();
  return kOk;
}
