// Wave-level scan / reduction primitives of the entropy stage (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The emulator only (tools/hipemu runs the lanes of a wave out of lockstep between two cross-lane operations): a point where the
// hardware's lockstep orders an exchange through LDS by itself — every lane reads before any lane writes — and nothing is wanted
// in the device code.
#ifndef KC_EMU_SYNC
#ifdef KC_HIPEMU
#define KC_EMU_SYNC() hipemu::wave_sync()
#else
#define KC_EMU_SYNC() do { } while (0)
#endif
#endif

// Wave-wide scans and reductions on the DPP data path (row shifts inside the 16-lane rows, then the row_bcast:15 / row_bcast:31
// steps across rows): six dependent VALU operations, where __shfl_up / __shfl_xor compile to six dependent ds_bpermute round trips
// through the LDS crossbar.  The scans / reductions below need all 64 lanes active (their callers are in wave-uniform control flow).
// kc_dpp_or0 alone is also used under DIVERGENT control flow by the match finders' dependency checks (kc_zstd_match.hip,
// kc_zstd_match_dfast.hip: inside the per-group `while (!fin)` loops, where whole 8-lane groups are masked off): a row_shr:d read
// by a lane with lig >= d stays inside that lane's own group, which is active as a whole; what lanes with lig < d receive comes
// from another group (possibly inactive: then 0, bound_ctrl) and is never looked at (`lig >= d &&` guards every use).  A DPP read
// from an inactive lane is not an error on the hardware — it yields the old value / 0 — so the guard is what makes it correct.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t kc_dpp_or0(uint32_t v) {  // the DPP-selected lane's v, 0 where there is none
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
    v += kc_dpp_or0<0x111, 0xf>(v);  // row_shr:1
    v += kc_dpp_or0<0x112, 0xf>(v);  // row_shr:2
    v += kc_dpp_or0<0x114, 0xf>(v);  // row_shr:4
    v += kc_dpp_or0<0x118, 0xf>(v);  // row_shr:8
    v += kc_dpp_or0<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += kc_dpp_or0<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, kc_dpp_or0<0x111, 0xf>(v));
    v = mx(v, kc_dpp_or0<0x112, 0xf>(v));
    v = mx(v, kc_dpp_or0<0x114, 0xf>(v));
    v = mx(v, kc_dpp_or0<0x118, 0xf>(v));
    v = mx(v, kc_dpp_or0<0x142, 0xa>(v));
    v = mx(v, kc_dpp_or0<0x143, 0xc>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63);
}

