// Device-wide prefix scans over uint32 (three launches: tile reduce, one
// workgroup over the tile totals, tile apply).
#pragma once

#include "common.hpp"

namespace bwtgpu {

enum class ScanOp { kAddExclusive, kMaxInclusive };

// bytes of scratch device_scan_u32 needs for `count` elements
size_t scan_workspace_bytes(uint64_t count);

// out[i] = op-scan of in[0..i) (exclusive add) or in[0..i] (inclusive max).
// in == out is allowed.  If d_total is not null it receives the reduction of
// the whole input.
int device_scan_u32(ScanOp op, const uint32_t *in, uint32_t *out, uint64_t count,
                    uint32_t *d_total, void *d_ws, size_t ws_bytes, hipStream_t s);

}  // namespace bwtgpu
