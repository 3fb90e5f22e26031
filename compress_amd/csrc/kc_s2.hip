// kc_s2.hip — S2 block encoder kernels (placeholder until the S2 milestone lands).
#include "kc_dev.h"
#include "kc_kernels.h"
#include "../../include/kcgpu.h"
extern "C" kc_status kc_s2_encode_blocks_dev_impl(kc_ctx*, const uint8_t*, const uint64_t*, uint32_t, uint8_t*, uint64_t, uint64_t*) {
    return KC_ERR_UNSUPPORTED;
}
