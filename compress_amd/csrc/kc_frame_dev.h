// kc_frame_dev.h — the zstd frame header (zstd/frameenc.go:25-92 as EncodeAll / the stream writer fill it in,
// zstd/encoder.go:756-772, 257-300), shared by the entropy kernel and the no-match pre-scan (kc_zstd_prescan.hip).
#pragma once
#include "kc_dev.h"

// Writes the header of a frame over `ulen` content bytes into hdr (<= 14 bytes), returns its length.  window_size is the
// encoder's (WithWindowSize); single_opt < 0: the reference's default (single segment iff ulen <= window && ulen > 1024);
// streamU: the Write ... Close layout (no content size, no single segment, the encoder's own window).
__device__ __forceinline__ int kc_frame_header(uint8_t* hdr, int ulen, int window_size, int single_opt, int crc, uint32_t did, bool streamU) {
    bool single = ulen <= window_size && ulen > 1024;
    if (single_opt >= 0) single = single_opt != 0;
    if (streamU) single = false;
    // fastBase.WindowSize (enc_base.go:42)
    uint32_t windowSize = (uint32_t)window_size;
    if (ulen < window_size && !streamU) {
        const uint32_t bsz = 1u << bits_len32((uint32_t)ulen);
        windowSize = bsz < 1024u ? 1024u : bsz;
    }
    int h = 0;
    hdr[h++] = 0x28; hdr[h++] = 0xb5; hdr[h++] = 0x2f; hdr[h++] = 0xfd;
    uint8_t fhd = 0;
    if (crc) fhd |= 1 << 2;
    if (single) fhd |= 1 << 5;
    int didLen = 0;
    if (did > 0) { if (did < 256) { fhd |= 1; didLen = 1; } else if (did < (1u << 16)) { fhd |= 2; didLen = 2; } else { fhd |= 3; didLen = 4; } }
    uint8_t fcs = 0;
    if (!streamU) {  // streaming: ContentSize 0 -> no FCS field (frameenc.go:40-58)
        if (ulen >= 256) fcs++;
        if (ulen >= 65536 + 256) fcs++;
    }
    fhd |= (uint8_t)(fcs << 6);
    hdr[h++] = fhd;
    if (!single) hdr[h++] = (uint8_t)((bits_len32(windowSize - 1) - 10) << 3);
    for (int i = 0; i < didLen; i++) hdr[h++] = (uint8_t)(did >> (8 * i));
    if (streamU) { /* no content size */ }
    else if (fcs == 0) { if (single) hdr[h++] = (uint8_t)ulen; }
    else if (fcs == 1) { const uint32_t c = (uint32_t)ulen - 256; hdr[h++] = (uint8_t)c; hdr[h++] = (uint8_t)(c >> 8); }
    else { for (int i = 0; i < 4; i++) hdr[h++] = (uint8_t)((uint32_t)ulen >> (8 * i)); }
    return h;
}

// blockHeader (zstd/blockenc.go:109-136): last(1) | type(2) | size(21)
__device__ __forceinline__ void put_block_header(uint8_t* p, bool last, uint32_t type, uint32_t size) {
    const uint32_t h = (last ? 1u : 0u) | (type << 1) | (size << 3);
    p[0] = (uint8_t)h; p[1] = (uint8_t)(h >> 8); p[2] = (uint8_t)(h >> 16);
}
