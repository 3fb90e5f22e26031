// oracle/kco_common.h — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// This directory is a CPU restatement of the klauspost/compress encode hot path
// (zstd EncodeAll at SpeedFastest/Default/Better, S2 block encode).  It exists only
// as the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
// Nothing under compress_amd/ (the product) may include, link or call it.
//
// PARITY STATUS: s2.Encode / s2.EncodeSnappy in their amd64 assembly form are PINNED: oracle/_ref runs the reference's own
// assembly encoders and kco_s2_asm.h equals them byte for byte (tests/test_ref_s2asm.py).  Everything else is
// "parity unpinned" at whole-encoder level — the reference has no test
// that fixes encoder output bytes and no Go toolchain exists here, so the
// oracle is pinned by (1) the reference's KATs (XXH64, matchLen, S2 emitLiteral/emitCopy,
// MaxEncodedLen), (2) frame-boundary KATs for C1 (e.txt header/trailer), and (3) every
// output decoding back to the input with the independent libzstd 1.4.8 / a restated
// S2 decoder.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <vector>
#include <algorithm>

namespace kco {

typedef std::vector<uint8_t> Bytes;

// internal/le/unsafe_enabled.go: unaligned little-endian loads.
static inline uint32_t load32(const uint8_t* b, int64_t i) { uint32_t v; memcpy(&v, b + i, 4); return v; }
static inline uint64_t load64(const uint8_t* b, int64_t i) { uint64_t v; memcpy(&v, b + i, 8); return v; }
static inline uint16_t load16(const uint8_t* b, int64_t i) { uint16_t v; memcpy(&v, b + i, 2); return v; }

// math/bits
static inline int bitsLen32(uint32_t v) { return v == 0 ? 0 : 32 - __builtin_clz(v); }
static inline int bitsLen64(uint64_t v) { return v == 0 ? 0 : 64 - __builtin_clzll(v); }
static inline int tz64(uint64_t v) { return v == 0 ? 64 : __builtin_ctzll(v); }
// zstd/seqenc.go:44 highBit, huff0/huff0.go:335 highBit32, fse/fse.go:142 highBits:
// uint32(bits.Len32(val) - 1); highBit(0) == 0xFFFFFFFF (App. A-22).
static inline uint32_t highBit(uint32_t v) { return (uint32_t)(bitsLen32(v) - 1); }

// Plain LSB-first bit writer. All three reference bit writers (zstd/bitwriter.go,
// huff0/bitwriter.go, fse/bitwriter.go) are an LSB-first concatenation of (value,nbits)
// fields into a 64-bit container with flushes that never change the bits; we restate
// them with the same container discipline so overflow behaviour would match too.
struct BitWriter {
    uint64_t bitContainer = 0;
    uint8_t nBits = 0;
    Bytes* out = nullptr;
    void reset(Bytes* o) { bitContainer = 0; nBits = 0; out = o; }
    // zstd/bitwriter.go:33 / fse/bitwriter.go:29 addBits16NC
    void addBits16NC(uint16_t value, uint8_t bits) {
        static const uint16_t bitMask16[32] = {0, 1, 3, 7, 0xF, 0x1F, 0x3F, 0x7F, 0xFF, 0x1FF, 0x3FF, 0x7FF,
            0xFFF, 0x1FFF, 0x3FFF, 0x7FFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF,
            0xFFFF, 0xFFFF, 0, 0, 0, 0, 0, 0};
        bitContainer |= (uint64_t)(value & bitMask16[bits & 31]) << (nBits & 63);
        nBits += bits;
    }
    // zstd/bitwriter.go:41 addBits32NC
    void addBits32NC(uint32_t value, uint8_t bits) {
        uint32_t m = (bits & 31) == 0 ? 0u : (0xFFFFFFFFu >> (32 - (bits & 31)));
        bitContainer |= (uint64_t)(value & m) << (nBits & 63);
        nBits += bits;
    }
    // zstd/bitwriter.go:60 addBits32Clean / :67 addBits16Clean
    void addBits32Clean(uint32_t value, uint8_t bits) {
        bitContainer |= (uint64_t)value << (nBits & 63);
        nBits += bits;
    }
    // zstd/bitwriter.go:48 addBits64NC
    void addBits64NC(uint64_t value, uint8_t bits) {
        if (bits <= 31) { addBits32Clean((uint32_t)value, bits); return; }
        addBits32Clean((uint32_t)value, 32);
        flush32();
        addBits32Clean((uint32_t)(value >> 32), bits - 32);
    }
    // fse/bitwriter.go:45 addBits16ZeroNC
    void addBits16ZeroNC(uint16_t value, uint8_t bits) {
        if (bits == 0) return;
        value = (uint16_t)(value << ((16 - bits) & 15));
        value = (uint16_t)(value >> ((16 - bits) & 15));
        bitContainer |= (uint64_t)value << (nBits & 63);
        nBits += bits;
    }
    // zstd/bitwriter.go:74 flush32
    void flush32() {
        if (nBits < 32) return;
        out->push_back((uint8_t)bitContainer);
        out->push_back((uint8_t)(bitContainer >> 8));
        out->push_back((uint8_t)(bitContainer >> 16));
        out->push_back((uint8_t)(bitContainer >> 24));
        nBits -= 32;
        bitContainer >>= 32;
    }
    // fse/bitwriter.go:58 flush: all pending full bytes.
    void flush() {
        uint8_t v = nBits >> 3;
        for (uint8_t i = 0; i < v; i++) out->push_back((uint8_t)(bitContainer >> (8 * i)));
        bitContainer = v >= 8 ? 0 : bitContainer >> (v << 3);
        nBits &= 7;
    }
    // zstd/bitwriter.go:88 flushAlign
    void flushAlign() {
        uint8_t nbBytes = (uint8_t)((nBits + 7) >> 3);
        for (uint8_t i = 0; i < nbBytes; i++) out->push_back((uint8_t)(bitContainer >> (i * 8)));
        nBits = 0;
        bitContainer = 0;
    }
    // zstd/bitwriter.go:99 close: end mark bit then align.
    void close() { addBits32Clean(1, 1); flushAlign(); }
};

}  // namespace kco
