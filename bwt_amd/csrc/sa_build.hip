// placeholder until the SA kernels land
#include "common.hpp"
using namespace bwtgpu;
extern "C" {
size_t bwtgpu_sa_workspace_bytes(uint32_t n) { return 256 + (size_t)n; }
int bwtgpu_sa_build(const uint8_t *, uint32_t, uint32_t *, void *, size_t, void *) {
  return set_error(kErrInternal, "sa_build not implemented");
}
int bwtgpu_sa_last_stats(uint64_t s[8]) { for (int i = 0; i < 8; ++i) s[i] = 0; return 0; }
int bwtgpu_bwt_from_sa(const uint8_t *, const uint32_t *, uint32_t, uint8_t *, uint32_t *, void *) {
  return set_error(kErrInternal, "bwt_from_sa not implemented");
}
bwtgpu_index *bwtgpu_index_build(const uint8_t *, uint32_t, void *) {
  set_error(kErrInternal, "index_build not implemented"); return nullptr;
}
bwtgpu_index *bwtgpu_index_build_dev(const uint8_t *, uint32_t, void *) {
  set_error(kErrInternal, "index_build not implemented"); return nullptr;
}
}
