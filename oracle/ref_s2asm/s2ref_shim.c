// oracle/ref_s2asm/s2ref_shim.c — TEST INFRASTRUCTURE ONLY: C entry points over the reference's own amd64 S2 block encoders
// (oracle/_ref/s2ref_amd64.S, made from /root/reference/s2/encodeblock_amd64.s by plan9_to_gas.py).  What is restated here is only
// the Go glue around them: the size dispatch of s2/encode_amd64.go:23-316 and s2.Encode / EncodeBetter / EncodeSnappy /
// EncodeSnappyBetter (s2/encode.go:29-57, 117-144, 204-246, 248-276) — uvarint length, the block, or one literal when the block
// encoder returns 0.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P9(name) extern void p9_##name(uint64_t* frame)
P9(encodeBlockAsm); P9(encodeBlockAsm4MB); P9(encodeBlockAsm12B); P9(encodeBlockAsm10B); P9(encodeBlockAsm8B);
P9(encodeBetterBlockAsm); P9(encodeBetterBlockAsm4MB); P9(encodeBetterBlockAsm12B); P9(encodeBetterBlockAsm10B); P9(encodeBetterBlockAsm8B);
P9(encodeSnappyBlockAsm); P9(encodeSnappyBlockAsm64K); P9(encodeSnappyBlockAsm12B); P9(encodeSnappyBlockAsm10B); P9(encodeSnappyBlockAsm8B);
P9(encodeSnappyBetterBlockAsm); P9(encodeSnappyBetterBlockAsm64K); P9(encodeSnappyBetterBlockAsm12B); P9(encodeSnappyBetterBlockAsm10B);
P9(encodeSnappyBetterBlockAsm8B);
P9(emitLiteral); P9(emitRepeat); P9(emitCopy); P9(emitCopyNoRepeat); P9(matchLen); P9(calcBlockSize); P9(calcBlockSizeSmall);

// func encodeXxx(dst []byte, src []byte, tmp *[N]byte) int: frame = dst base/len/cap, src base/len/cap, tmp, ret
static int64_t call_block(void (*fn)(uint64_t*), uint8_t* dst, uint64_t dst_len, const uint8_t* src, uint64_t n, size_t tmp_bytes) {
    uint64_t f[8];
    void* tmp = aligned_alloc(64, (tmp_bytes + 63) & ~(size_t)63);
    if (!tmp) return -1;
    // (not cleared: the encoders clear their table themselves — the parity tests ran with this buffer poisoned)
    f[0] = (uint64_t)(uintptr_t)dst; f[1] = dst_len; f[2] = dst_len;
    f[3] = (uint64_t)(uintptr_t)src; f[4] = n; f[5] = n;
    f[6] = (uint64_t)(uintptr_t)tmp; f[7] = 0;
    fn(f);
    free(tmp);
    return (int64_t)f[7];
}

enum { minNonLiteralBlockSize = 32 };

// level: 0 encodeBlock, 1 encodeBlockBetter, 2 encodeBlockSnappy, 3 encodeBlockBetterSnappy (s2/encode_amd64.go)
int64_t s2ref_encode_block(int level, uint8_t* dst, uint64_t dst_len, const uint8_t* src, uint64_t n) {
    const uint64_t limit12B = 16 << 10, limit10B = 4 << 10, limit8B = 512;
    switch (level) {
    case 0:
        if (n >= (4u << 20)) return call_block(p9_encodeBlockAsm, dst, dst_len, src, n, 65536);
        if (n >= limit12B) return call_block(p9_encodeBlockAsm4MB, dst, dst_len, src, n, 65536);
        if (n >= limit10B) return call_block(p9_encodeBlockAsm12B, dst, dst_len, src, n, 16384);
        if (n >= limit8B) return call_block(p9_encodeBlockAsm10B, dst, dst_len, src, n, 4096);
        if (n < minNonLiteralBlockSize) return 0;
        return call_block(p9_encodeBlockAsm8B, dst, dst_len, src, n, 1024);
    case 1:
        if (n > (4u << 20)) return call_block(p9_encodeBetterBlockAsm, dst, dst_len, src, n, 589824);
        if (n >= limit12B) return call_block(p9_encodeBetterBlockAsm4MB, dst, dst_len, src, n, 589824);
        if (n >= limit10B) return call_block(p9_encodeBetterBlockAsm12B, dst, dst_len, src, n, 81920);
        if (n >= limit8B) return call_block(p9_encodeBetterBlockAsm10B, dst, dst_len, src, n, 20480);
        if (n < minNonLiteralBlockSize) return 0;
        return call_block(p9_encodeBetterBlockAsm8B, dst, dst_len, src, n, 5120);
    case 2:
        if (n > 65536) return call_block(p9_encodeSnappyBlockAsm, dst, dst_len, src, n, 65536);
        if (n >= limit12B) return call_block(p9_encodeSnappyBlockAsm64K, dst, dst_len, src, n, 65536);
        if (n >= limit10B) return call_block(p9_encodeSnappyBlockAsm12B, dst, dst_len, src, n, 16384);
        if (n >= limit8B) return call_block(p9_encodeSnappyBlockAsm10B, dst, dst_len, src, n, 4096);
        if (n < minNonLiteralBlockSize) return 0;
        return call_block(p9_encodeSnappyBlockAsm8B, dst, dst_len, src, n, 1024);
    case 3:
        if (n > 65536) return call_block(p9_encodeSnappyBetterBlockAsm, dst, dst_len, src, n, 589824);
        if (n >= limit12B) return call_block(p9_encodeSnappyBetterBlockAsm64K, dst, dst_len, src, n, 294912);
        if (n >= limit10B) return call_block(p9_encodeSnappyBetterBlockAsm12B, dst, dst_len, src, n, 81920);
        if (n >= limit8B) return call_block(p9_encodeSnappyBetterBlockAsm10B, dst, dst_len, src, n, 20480);
        if (n < minNonLiteralBlockSize) return 0;
        return call_block(p9_encodeSnappyBetterBlockAsm8B, dst, dst_len, src, n, 5120);
    }
    return -1;
}

// func emitLiteral(dst []byte, lit []byte) int
int64_t s2ref_emit_literal(uint8_t* dst, uint64_t dst_len, const uint8_t* lit, uint64_t n) {
    uint64_t f[7] = {(uint64_t)(uintptr_t)dst, dst_len, dst_len, (uint64_t)(uintptr_t)lit, n, n, 0};
    p9_emitLiteral(f);
    return (int64_t)f[6];
}
// func emitRepeat / emitCopy / emitCopyNoRepeat(dst []byte, offset int, length int) int
static int64_t call_emit(void (*fn)(uint64_t*), uint8_t* dst, uint64_t dst_len, int64_t offset, int64_t length) {
    uint64_t f[6] = {(uint64_t)(uintptr_t)dst, dst_len, dst_len, (uint64_t)offset, (uint64_t)length, 0};
    fn(f);
    return (int64_t)f[5];
}
int64_t s2ref_emit_repeat(uint8_t* dst, uint64_t dst_len, int64_t offset, int64_t length) { return call_emit(p9_emitRepeat, dst, dst_len, offset, length); }
int64_t s2ref_emit_copy(uint8_t* dst, uint64_t dst_len, int64_t offset, int64_t length) { return call_emit(p9_emitCopy, dst, dst_len, offset, length); }
int64_t s2ref_emit_copy_norepeat(uint8_t* dst, uint64_t dst_len, int64_t offset, int64_t length) { return call_emit(p9_emitCopyNoRepeat, dst, dst_len, offset, length); }
// func matchLen(a []byte, b []byte) int
int64_t s2ref_match_len(const uint8_t* a, uint64_t an, const uint8_t* b, uint64_t bn) {
    uint64_t f[7] = {(uint64_t)(uintptr_t)a, an, an, (uint64_t)(uintptr_t)b, bn, bn, 0};
    p9_matchLen(f);
    return (int64_t)f[6];
}

// zstd's matchLen (zstd/matchlen_amd64.s:8; what fastBase.matchlen calls on amd64, zstd/enc_base.go:117-131)
extern void p9_zstd_matchLen(uint64_t* frame);
int64_t zstdref_match_len(const uint8_t* a, uint64_t an, const uint8_t* b, uint64_t bn) {
    uint64_t f[7] = {(uint64_t)(uintptr_t)a, an, an, (uint64_t)(uintptr_t)b, bn, bn, 0};
    p9_zstd_matchLen(f);
    return (int64_t)f[6];
}

static int put_uvarint(uint8_t* dst, uint64_t x) {
    int i = 0;
    while (x >= 0x80) { dst[i++] = (uint8_t)x | 0x80; x >>= 7; }
    dst[i++] = (uint8_t)x;
    return i;
}

// s2.Encode (level 0), EncodeBetter (1), EncodeSnappy (2), EncodeSnappyBetter (3) on amd64.  dst_len >= MaxEncodedLen(n).
int64_t s2ref_encode(int level, uint8_t* dst, uint64_t dst_len, const uint8_t* src, uint64_t n) {
    int64_t d = put_uvarint(dst, n);
    if (n == 0) return d;
    if (n < minNonLiteralBlockSize) return d + s2ref_emit_literal(dst + d, dst_len - (uint64_t)d, src, n);
    const int64_t k = s2ref_encode_block(level, dst + d, dst_len - (uint64_t)d, src, n);
    if (k < 0) return k;
    if (k > 0) return d + k;
    return d + s2ref_emit_literal(dst + d, dst_len - (uint64_t)d, src, n);  // not compressible
}

// Blocks src[off[i] .. off[i+1]) encoded back to back on `threads` host threads is the caller's business (tests/oracle_ref.py);
// this one is the single-call form for timing loops: returns the total bytes, writes nothing but `scratch` (>= MaxEncodedLen of
// the largest block).
int64_t s2ref_encode_blocks_size(int level, const uint8_t* src, const uint64_t* off, uint32_t n, uint8_t* scratch, uint64_t scratch_len) {
    int64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        const int64_t r = s2ref_encode(level, scratch, scratch_len, src + off[i], off[i + 1] - off[i]);
        if (r < 0) return r;
        total += r;
    }
    return total;
}

// N blocks through s2ref_encode on `threads` host threads (the bench's cpu_baseline of kind "reference" and the byte compare of its
// sample): the encoded blocks back to back in dst, out_off[i] = start of block i.  Returns the total, -2 when dst is too small.
#include <pthread.h>
struct blk_job {
    int level;
    const uint8_t* src;
    const uint64_t* off;
    uint32_t n;
    uint8_t** outs;
    int64_t* sizes;
    uint32_t next;
    pthread_mutex_t mu;
    uint64_t maxlen;
};
static void* blk_worker(void* arg) {
    struct blk_job* j = (struct blk_job*)arg;
    const uint64_t cap = j->maxlen + j->maxlen / 6 + 64;
    uint8_t* scratch = (uint8_t*)malloc(cap);
    for (;;) {
        pthread_mutex_lock(&j->mu);
        const uint32_t i = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (i >= j->n) break;
        const uint64_t len = j->off[i + 1] - j->off[i];
        const int64_t r = s2ref_encode(j->level, scratch, cap, j->src + j->off[i], len);
        j->sizes[i] = r;
        if (r > 0) {
            j->outs[i] = (uint8_t*)malloc((size_t)r);
            memcpy(j->outs[i], scratch, (size_t)r);
        }
    }
    free(scratch);
    return 0;
}
int64_t s2ref_encode_blocks(int level, const uint8_t* src, const uint64_t* off, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_off,
                            int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    struct blk_job j;
    j.level = level; j.src = src; j.off = off; j.n = n; j.next = 0; j.maxlen = 32;
    for (uint32_t i = 0; i < n; i++) if (off[i + 1] - off[i] > j.maxlen) j.maxlen = off[i + 1] - off[i];
    j.outs = (uint8_t**)calloc(n ? n : 1, sizeof(uint8_t*));
    j.sizes = (int64_t*)calloc(n ? n : 1, sizeof(int64_t));
    pthread_mutex_init(&j.mu, 0);
    pthread_t th[256];
    for (int t = 0; t < threads; t++) pthread_create(&th[t], 0, blk_worker, &j);
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
    int64_t pos = 0, rc = 0;
    for (uint32_t i = 0; i < n; i++) {
        out_off[i] = (uint64_t)pos;
        if (j.sizes[i] <= 0) rc = -1;
        else if ((uint64_t)(pos + j.sizes[i]) > dst_cap) rc = -2;
        else { memcpy(dst + pos, j.outs[i], (size_t)j.sizes[i]); pos += j.sizes[i]; }
        free(j.outs[i]);
    }
    out_off[n] = (uint64_t)pos;
    free(j.outs); free(j.sizes);
    pthread_mutex_destroy(&j.mu);
    return rc < 0 ? rc : pos;
}


// ---- zstd/internal/xxhash/xxhash_amd64.s: the XXH64 the reference's zstd encoder checksums frames with on amd64 ----
// var primes = [...]uint64{prime1, prime2, prime3, prime4, prime5} (xxhash.go:13-25), read by the assembly as ·primes+off(SB)
__attribute__((visibility("hidden"))) uint64_t p9data_primes[5] = {11400714785074694791ULL, 14029467366897019727ULL, 1609587929392839161ULL, 9650029242287828579ULL, 2870177450012600261ULL};
extern void p9_Sum64(uint64_t* frame);
// func Sum64(b []byte) uint64
uint64_t zref_xxh64_sum(const uint8_t* b, uint64_t n) {
    uint64_t f[4] = {(uint64_t)(uintptr_t)b, n, n, 0};
    p9_Sum64(f);
    return f[3];
}


// ---- s2/decode_amd64.s: the reference's own block decoder (s2.Decode, s2/decode.go:58-76: decodedLen + s2Decode) ----
extern void p9_s2Decode(uint64_t* frame);
// Returns the decoded length, -1 for a corrupt block (the reference's ErrCorrupt), -2 when dst is too small.
int64_t s2ref_decode(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) {
    uint64_t v = 0;
    int shift = 0;
    uint64_t s = 0;
    for (;;) {  // binary.Uvarint
        if (s >= n || shift > 63) return -1;
        const uint8_t b = src[s++];
        v |= (uint64_t)(b & 0x7f) << shift;
        if (b < 0x80) break;
        shift += 7;
    }
    if (v > 0xffffffffULL) return -1;  // decodedLen: ErrTooLarge / ErrCorrupt
    if (v > cap) return -2;
    // func s2Decode(dst, src []byte) int
    uint64_t f[7] = {(uint64_t)(uintptr_t)dst, v, v, (uint64_t)(uintptr_t)(src + s), n - s, n - s, 0};
    p9_s2Decode(f);
    return f[6] != 0 ? -1 : (int64_t)v;
}
