// kc_s2_decode.hip — S2 block decoder on the device: the verifier half of SURVEY.md §8f N1 for S2
// (s2.Decode -> s2Decode, s2/decode.go:58, s2/decode_other.go:22-290).  Eight lanes per block, eight blocks per wave:
// the tag stream is sequential (every lane of the group parses it redundantly from the same bytes, so no broadcast is
// needed), literal and match copies are spread over the lanes.  A match whose offset is shorter than its length repeats
// the last `offset` bytes, which are complete before the copy starts, so lane i reads dst[d - offset + (i mod offset)].
#include "kc_dev.h"
#include "kc_kernels.h"

#define S2D_G 8

__global__ __launch_bounds__(64) void kc_s2_decode_kernel(KcS2DecParams P) {
    const int lane = (int)threadIdx.x, lig = lane % S2D_G, grp = lane / S2D_G;
    const uint32_t bi = blockIdx.x * (64 / S2D_G) + (uint32_t)grp;
    if (bi >= P.n_blocks) return;
    const uint8_t* __restrict__ src = P.enc + P.enc_off[bi];
    const uint32_t n = (uint32_t)(P.enc_off[bi + 1] - P.enc_off[bi]);
    uint8_t* __restrict__ dst = P.dst + P.dst_off[bi];
    const uint64_t want = P.dst_off[bi + 1] - P.dst_off[bi];
    uint32_t err = 0;
    // uvarint decoded length (s2/decode.go:23-40)
    uint32_t s = 0;
    uint64_t dLen = 0;
    {
        int shift = 0;
        for (;;) {
            if (s >= n || shift > 63) { err = 1; break; }
            const uint8_t b = src[s++];
            dLen |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) break;
            shift += 7;
        }
    }
    if (!err && dLen != want) err = 2;  // the caller states the size each block must decode to
    uint64_t d = 0, offset = 0;
    while (!err && s < n) {
        uint64_t length;
        const uint8_t tag = src[s];
        const int kind = tag & 3;
        if (kind == 0) {  // literal
            uint32_t x = tag >> 2;
            if (x < 60) { s += 1; }
            else if (x == 60) { if (s + 2 > n) { err = 3; break; } x = src[s + 1]; s += 2; }
            else if (x == 61) { if (s + 3 > n) { err = 3; break; } x = ld16(src + s + 1); s += 3; }
            else if (x == 62) { if (s + 4 > n) { err = 3; break; } x = (uint32_t)src[s + 1] | (uint32_t)src[s + 2] << 8 | (uint32_t)src[s + 3] << 16; s += 4; }
            else { if (s + 5 > n) { err = 3; break; } x = ld32(src + s + 1); s += 5; }
            length = (uint64_t)x + 1;
            if (length > dLen - d || length > (uint64_t)(n - s)) { err = 4; break; }
            const uint32_t body = (uint32_t)length & ~7u;
            for (uint32_t k = (uint32_t)lig * 8; k < body; k += S2D_G * 8) st64(dst + d + k, ld64(src + s + k));
            for (uint32_t k = body + (uint32_t)lig; k < (uint32_t)length; k += S2D_G) dst[d + k] = src[s + k];
            d += length;
            s += (uint32_t)length;
            continue;
        }
        if (kind == 1) {  // copy1 / repeat (decode_other.go:77-121)
            if (s + 2 > n) { err = 3; break; }
            const uint64_t toffset = ((uint64_t)(tag & 0xe0) << 3) | src[s + 1];
            length = (tag >> 2) & 7;
            s += 2;
            if (toffset == 0) {
                if (length == 5) { if (s + 1 > n) { err = 3; break; } length = (uint64_t)src[s] + 4; s += 1; }
                else if (length == 6) { if (s + 2 > n) { err = 3; break; } length = (uint64_t)ld16(src + s) + (1 << 8); s += 2; }
                else if (length == 7) { if (s + 3 > n) { err = 3; break; } length = ((uint64_t)src[s] | (uint64_t)src[s + 1] << 8 | (uint64_t)src[s + 2] << 16) + (1 << 16); s += 3; }
            } else {
                offset = toffset;
            }
            length += 4;
        } else if (kind == 2) {
            if (s + 3 > n) { err = 3; break; }
            offset = ld16(src + s + 1);
            length = 1 + (uint64_t)(tag >> 2);
            s += 3;
        } else {
            if (s + 5 > n) { err = 3; break; }
            offset = ld32(src + s + 1);
            length = 1 + (uint64_t)(tag >> 2);
            s += 5;
        }
        if (offset == 0 || d < offset || length > dLen - d) { err = 5; break; }
        // The literal bytes written just above by OTHER lanes of the group must be visible before they are read back.
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (offset >= 8 && offset >= length) {
            const uint32_t body = (uint32_t)length & ~7u;
            for (uint32_t k = (uint32_t)lig * 8; k < body; k += S2D_G * 8) st64(dst + d + k, ld64(dst + d - offset + k));
            for (uint32_t k = body + (uint32_t)lig; k < (uint32_t)length; k += S2D_G) dst[d + k] = dst[d - offset + k];
        } else {
            for (uint32_t k = (uint32_t)lig; k < (uint32_t)length; k += S2D_G) dst[d + k] = dst[d - offset + (k % (uint32_t)offset)];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        d += length;
    }
    if (!err && d != dLen) err = 6;
    if (lig == 0) P.status[bi] = err;
}

void kc_launch_s2_decode(const KcS2DecParams& P, hipStream_t st) {
    if (P.n_blocks == 0) return;
    hipLaunchKernelGGL(kc_s2_decode_kernel, dim3((P.n_blocks + 7) / 8), dim3(64), 0, st, P);
}
