// oracle/kco_zstd_better.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// PLACEHOLDER: zstd/enc_better.go (betterFastEncoder) is restated in a later milestone.
#pragma once
#include "kco_zstd_fast.h"
namespace kco {
struct BetterFastEncoder : FastEncoder { bool unsupported = true; };
struct BetterFastEncoderDict : FastEncoderDict { bool unsupported = true; };
}  // namespace kco
