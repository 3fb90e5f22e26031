// oracle/kco_xxhash.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// XXH64 seed 0, restating zstd/internal/xxhash/xxhash.go:27-230 (Digest.Reset/Write/Sum64).
#pragma once
#include "kco_common.h"

namespace kco {

struct XXH64 {
    static constexpr uint64_t prime1 = 11400714785074694791ULL;
    static constexpr uint64_t prime2 = 14029467366897019727ULL;
    static constexpr uint64_t prime3 = 1609587929392839161ULL;
    static constexpr uint64_t prime4 = 9650029242287828579ULL;
    static constexpr uint64_t prime5 = 2870177450012600261ULL;
    uint64_t v1, v2, v3, v4, total;
    uint8_t mem[32];
    int n;
    XXH64() { Reset(); }
    // xxhash.go:45 Reset
    void Reset() {
        v1 = prime1 + prime2;
        v2 = prime2;
        v3 = 0;
        v4 = 0 - prime1;
        total = 0;
        n = 0;
    }
    static inline uint64_t rol(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
    // xxhash.go:215 round
    static inline uint64_t round(uint64_t acc, uint64_t input) {
        acc += input * prime2;
        acc = rol(acc, 31);
        acc *= prime1;
        return acc;
    }
    // xxhash.go:222 mergeRound
    static inline uint64_t mergeRound(uint64_t acc, uint64_t val) {
        val = round(0, val);
        acc ^= val;
        acc = acc * prime1 + prime4;
        return acc;
    }
    // xxhash.go:61 Write
    void Write(const uint8_t* b, size_t len) {
        total += len;
        size_t memleft = 32 - (size_t)n;
        if (len < memleft) {
            memcpy(mem + n, b, len);
            n += (int)len;
            return;
        }
        if (n > 0) {
            memcpy(mem + n, b, memleft);
            v1 = round(v1, load64(mem, 0));
            v2 = round(v2, load64(mem, 8));
            v3 = round(v3, load64(mem, 16));
            v4 = round(v4, load64(mem, 24));
            b += memleft;
            len -= memleft;
            n = 0;
        }
        while (len >= 32) {
            v1 = round(v1, load64(b, 0));
            v2 = round(v2, load64(b, 8));
            v3 = round(v3, load64(b, 16));
            v4 = round(v4, load64(b, 24));
            b += 32;
            len -= 32;
        }
        memcpy(mem, b, len);
        n = (int)len;
    }
    // xxhash.go:112 Sum64
    uint64_t Sum64() const {
        uint64_t h;
        if (total >= 32) {
            h = rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18);
            h = mergeRound(h, v1);
            h = mergeRound(h, v2);
            h = mergeRound(h, v3);
            h = mergeRound(h, v4);
        } else {
            h = v3 + prime5;
        }
        h += total;
        const uint8_t* b = mem;
        int len = n & 31;
        for (; len >= 8; b += 8, len -= 8) {
            uint64_t k1 = round(0, load64(b, 0));
            h ^= k1;
            h = rol(h, 27) * prime1 + prime4;
        }
        if (len >= 4) {
            h ^= (uint64_t)load32(b, 0) * prime1;
            h = rol(h, 23) * prime2 + prime3;
            b += 4;
            len -= 4;
        }
        for (; len > 0; b++, len--) {
            h ^= (uint64_t)b[0] * prime5;
            h = rol(h, 11) * prime1;
        }
        h ^= h >> 33;
        h *= prime2;
        h ^= h >> 29;
        h *= prime3;
        h ^= h >> 32;
        return h;
    }
};

}  // namespace kco
