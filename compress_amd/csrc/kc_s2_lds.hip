// kc_s2_lds.hip — S2 block encoder with LDS-resident tables: ONE WAVE PER BLOCK, the latency path.
//
// Same reference functions as kc_s2.hip (s2.Encode / s2.EncodeSnappy: encodeBlockGo / encodeBlockGo64K /
// encodeBlockSnappyGo / ...64K, s2/encode_all.go:72-284, 287-500, 502-889; emit*: s2/encode_go.go:80-290) and the
// same bytes.  kc_s2.hip keeps 8 blocks per wave in flight with their tables in HBM: it needs thousands of blocks to
// cover the latency of its dependent table -> candidate chain and takes ~30 ms for one block.  Here the block's hash
// table (2^14 x u32 = 64 KiB) AND, for blocks up to 64 KiB, the block itself live in the CU's LDS (128 KiB of its
// 160 KiB), so a probe round is LDS-only: ~1 ms per 64 KiB block, whatever the number of blocks in flight.  The
// dispatcher in kc_s2_api.cpp picks the path by blocks in flight (measured crossover, profiles/r03_crossover_s2.csv).
//
// Execution scheme: the 64 lanes evaluate the next W probe steps of the reference's scan (positions follow
// nextS = s + (s-nextEmit)>>skip + 4 exactly, each lane iterating the recurrence up to its own step) against the
// pre-round table; ballot + ctz picks the first step that ends the scan the way the sequential encoder would; steps
// up to it commit their table writes.  A step whose bucket is also touched by another step of the round is detected
// through a marker byte: table entries are `position | marker << 24`; every lane stores its lane id into the marker
// byte of its three buckets (ds_write_b8), then reads the entries back — a lane that does not find its own id in all
// three shares a bucket with the lane whose id it finds, and tells it through a 64-bit mask in LDS (ds_or_b64, only in
// rounds where some lane lost).  The round is cut at the lowest sharing lane other than lane 0 (the first step depends
// on nothing), so no committed step ever saw a table that differs from the sequential encoder's.  No tags: candidates
// are verified on the bytes, which sit in LDS.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_s2_dev.h"

#define S2L_SRC_MAX 65536
#ifdef KC_S2_PROF  // diagnostics (tools/s2_lds_prof.hip): shader clocks and event counts per phase of the fused step, block 0 only
__device__ unsigned long long kc_s2_prof[16];
#define S2PROF(k) do { const unsigned long long t1__ = __builtin_readcyclecounter(); pacc[k] += t1__ - pt0; pcnt[k]++; pt0 = __builtin_readcyclecounter(); } while (0)
#else
#define S2PROF(k) do { } while (0)
#endif
#define S2L_POS_MASK 0xFFFFFFu
#define S2L_TAIL 544  // bytes behind the block in LDS: the fused step loads up to 63 x 8 bytes past a position before it looks at the bounds

// CRC32C of [0, len) read through rd32 / rdb, all 64 lanes cooperating: lane j takes bytes [j*C, (j+1)*C); the raw
// (init 0) remainders are combined left to right, acc = advance(acc, C zero bytes) ^ part[j], with the 32 columns of
// "advance by C zero bytes" computed by lanes 0..31.  The 0xFFFFFFFF initial value equals an XOR into the first word.
template <class RD32, class RDB>
__device__ __forceinline__ uint32_t s2_crc32c_wave(RD32 rd32, RDB rdb, int len, const uint32_t (*T)[256], uint32_t* colM, uint32_t* part, int lane) {
    auto step4 = [&](uint32_t c) -> uint32_t { return T[3][c & 0xFF] ^ T[2][(c >> 8) & 0xFF] ^ T[1][(c >> 16) & 0xFF] ^ T[0][c >> 24]; };
    if (len < 512) {  // short: every lane runs the plain loop (uniform)
        uint32_t c = 0xFFFFFFFFu;
        int i = 0;
        for (; i + 4 <= len; i += 4) c = step4(c ^ rd32(i));
        for (; i < len; i++) c = T[0][(c ^ rdb(i)) & 0xFF] ^ (c >> 8);
        return c ^ 0xFFFFFFFFu;
    }
    const int C = ((len + 63) / 64 + 3) & ~3;
    const int b = lane * C;
    const int e = b + C < len ? b + C : len;
    uint32_t c = 0;
    if (b < len) {
        int i = b;
        for (; i + 4 <= e; i += 4) {
            uint32_t w = rd32(i);
            if (i == 0) w ^= 0xFFFFFFFFu;
            c = step4(c ^ w);
        }
        for (; i < e; i++) c = T[0][(c ^ rdb(i)) & 0xFF] ^ (c >> 8);
    }
    if (lane < 32) {
        uint32_t v = 1u << lane;
        for (int i = 0; i < C; i += 4) v = step4(v);
        colM[lane] = v;
    }
    part[lane] = c;
    KC_WAVE_SYNC();
    const int nfull = len / C;
    uint32_t acc = 0;
    for (int j = 0; j < nfull; j++) {
        uint32_t a = 0;
        for (int k = 0; k < 32; k++) if ((acc >> k) & 1u) a ^= colM[k];
        acc = a ^ part[j];
    }
    const int r = len - nfull * C;
    if (r > 0) {
        int i = 0;
        for (; i + 4 <= r; i += 4) acc = step4(acc);
        for (; i < r; i++) acc = T[0][acc & 0xFF] ^ (acc >> 8);
        acc ^= part[nfull];
    }
    KC_WAVE_SYNC();
    return acc ^ 0xFFFFFFFFu;
}

// LEVEL 0: s2.Encode, 2: s2.EncodeSnappy; SRCLDS: block <= 64 KiB, held in LDS; ONESTEP (with SRCLDS): one probe step at a time,
// every lane the same — with the block in LDS a step is two LDS round trips whether or not other steps run beside it, so the
// speculative round's machinery (positions, markers, ballots, winner selection: ~600 instructions for one sequence) buys nothing
// and a wave-uniform step (scalar hashes, broadcast LDS reads, scalar branches) is what is left on the critical path.
//
// MODE 2 (with SRCLDS; the default for blocks held in LDS), "fused step": the same wave-uniform step, laid out for the fewest
// DEPENDENT LDS round trips and instructions — one wave issues one instruction per ~4 clocks and an LDS round trip costs ~30 of
// them, so both count.  A probe step is two trips: (1) lanes 0 / 1 / 2 hash the bytes at s / s+1 / s+2 (their own 8-byte loads,
// issued one step ahead) and read their buckets, lanes 0 / 1 write s / s+1, lane 2 reads its bucket again behind the writes
// (encode_all.go:401: table[hash2] is read after the two stores); (2) four 16-lane groups — the candidates of s, s+1, s+2 and
// the repeat candidate of s+1 — each load 8 bytes per lane on both sides: lane 0 of a group the 4 bytes before and the 4 bytes
// at the candidate (backward extension + the 4-byte verification), lanes 1..15 the 120 bytes behind them (forward extension).
// ONE ballot then holds which candidates verify and how far each match runs; the reference's priority order picks the winner in
// scalar code.  The immediate re-match test after a copy (:474-488) is the same two trips with one 64-lane group.
template <int LEVEL, bool SRCLDS, int MODE>
__global__ __launch_bounds__(64) void kc_s2_encode_lds_kernel(KcS2Params P) {
    constexpr bool SNAPPY = LEVEL == 2;
    constexpr bool ONESTEP = MODE == 1;
    constexpr bool FUSED = MODE == 2;
    constexpr int FRONT = 16;  // bytes in front of the block in LDS: the fused step reads the 4 bytes before a candidate unconditionally
    __shared__ uint32_t tab[1 << S2_TABLE_BITS];
    __shared__ __attribute__((aligned(16))) uint8_t lbuf[SRCLDS ? FRONT + S2L_SRC_MAX + S2L_TAIL : 16];
    __shared__ uint32_t sink[64];  // where the lanes without a table store of their own write (one store instruction, no exec masking)
    uint8_t* const lsrc = lbuf + (SRCLDS ? FRONT : 0);
    __shared__ uint32_t crcT[4][256];
    __shared__ uint32_t crcM[32];
    __shared__ uint32_t crcP[64];
    __shared__ unsigned long long shareMask;  // lanes that share a table bucket with another lane of the round
    const int lane = (int)threadIdx.x;
    const uint32_t bi = blockIdx.x;
    if (bi >= P.n_blocks) return;
    const uint8_t* __restrict__ src = P.src + P.blk_off[bi];
    const int len = (int)(P.blk_off[bi + 1] - P.blk_off[bi]);
    if ((len <= S2L_SRC_MAX) != SRCLDS) return;  // the other instantiation's block
    uint8_t* __restrict__ slot = P.stage + P.stage_off[bi];
    uint8_t* __restrict__ out = slot + (P.framed ? 8 : 0);

    if (P.framed) {
        for (int i = lane; i < 256; i += 64) {
            uint32_t c = (uint32_t)i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            crcT[0][i] = c;
        }
        KC_WAVE_SYNC();
        for (int i = lane; i < 256; i += 64) {
            uint32_t c = crcT[0][i];
            for (int t = 1; t < 4; t++) { c = crcT[0][c & 0xFF] ^ (c >> 8); crcT[t][i] = c; }
        }
    }
    for (int i = lane * 4; i < (1 << S2_TABLE_BITS); i += 256) *(uint4*)&tab[i] = make_uint4(0, 0, 0, 0);
    if (SRCLDS) {
        const int body = len & ~15;
        for (int o = lane * 16; o < body; o += 1024) *(uint4*)(lsrc + o) = ld128u(src + o);
        for (int o = body + lane; o < len; o += 64) lsrc[o] = src[o];
        for (int o = lane; o < S2L_TAIL; o += 64) lsrc[len + o] = 0;  // reads of whole words around and past the last bytes stay inside the array
        if (lane < FRONT) lbuf[lane] = 0;
    }
    KC_WAVE_SYNC();

    // ---- source access: LDS or global ----
    auto rd32 = [&](int pos) -> uint32_t { return SRCLDS ? ld32(lsrc + pos) : ld32(src + pos); };  // LDS takes unaligned ds_read_b32 / b64
    auto rd64 = [&](int pos) -> uint64_t { return SRCLDS ? ld64(lsrc + pos) : ld64(src + pos); };
    auto rdb = [&](int pos) -> uint32_t { return SRCLDS ? (uint32_t)lsrc[pos] : (uint32_t)src[pos]; };

    // uvarint(len) header (encode.go:39)
    int hdr = 0;
    {
        uint64_t x = (uint64_t)len;
        while (x >= 0x80) { if (lane == 0) out[hdr] = (uint8_t)x | 0x80; hdr++; x >>= 7; }
        if (lane == 0) out[hdr] = (uint8_t)x;
        hdr++;
    }
    uint8_t* __restrict__ dst = out + hdr;
    int d = 0;
    bool stored = false;  // encodeBlock returned 0 -> emit everything as one literal
    if (len == 0 && !P.framed) { if (lane == 0) P.out_size[bi] = (uint32_t)hdr; return; }
    if (len < 32 || P.stored_only) stored = true;  // minNonLiteralBlockSize (also len == 0 in a framed stream); s2.WriterUncompressed

    // emitLiteral(dst[d:], src[from:from+n]) (encode_go.go:80): lane 0 the tag, all lanes the bytes
    // Tag bytes are assembled in a register by every lane (the values are wave-uniform: scalar code) and stored by lane 0 as ONE
    // 8-byte word together with the first literal bytes; the bytes of that word past the emit's end are overwritten by the next
    // emit (emission is strictly sequential, the slot has MaxEncodedLen - dstLimit bytes of slack).
    auto emit_lit = [&](int from, int n) -> int {
        if (n == 0) return 0;
        const uint32_t m = (uint32_t)(n - 1);
        uint8_t* __restrict__ o = dst + d;
        int i;
        uint64_t tag;
        if (m < 60) { i = 1; tag = m << 2; }
        else if (m < (1u << 8)) { i = 2; tag = (60u << 2) | ((uint64_t)m << 8); }
        else if (m < (1u << 16)) { i = 3; tag = (61u << 2) | ((uint64_t)m << 8); }
        else if (m < (1u << 24)) { i = 4; tag = (62u << 2) | ((uint64_t)m << 8); }
        else { i = 5; tag = (63u << 2) | ((uint64_t)m << 8); }
        const int head = n < 8 - i ? n : 8 - i;  // literal bytes that share the tag's word
        if (lane == 0) {
            uint64_t hv = 0;
            if (SRCLDS || from + 8 <= len) hv = rd64(from);
            else for (int k = 0; k < head; k++) hv |= (uint64_t)rdb(from + k) << (8 * k);
            st64(o, tag | (hv << (8 * i)));
        }
        const int rest = n - head;
        const int body = rest & ~7;
        for (int k = lane * 8; k < body; k += 512) st64(o + 8 + k, rd64(from + head + k));
        for (int k = body + lane; k < rest; k += 64) o[8 + k] = (uint8_t)rdb(from + head + k);
        return i + n;
    };
    // forward extension in whole 8-byte steps while a <= limit (encode_all.go:353-360, :435-442); returns the new a
    auto extend = [&](int a, int b, int limit) -> int {
        for (;;) {
            const int pa = a + 8 * lane, pb = b + 8 * lane;
            const bool inb = pa <= limit;
            uint64_t diff = 0;
            if (inb) diff = rd64(pa) ^ rd64(pb);
            const uint64_t oob = ballot64(!inb);
            const uint64_t dm = ballot64(inb && diff != 0);
            const int firstOob = oob ? ctz64(oob) : 64;
            if (dm) {
                const int fl = ctz64(dm);  // necessarily < firstOob
                const uint64_t dd = rdlane64(diff, fl);
                return a + 8 * fl + (ctz64(dd) >> 3);
            }
            if (firstOob < 64) return a + 8 * firstOob;
            a += 512;
            b += 512;
        }
    };
    // the assembly's matchLen (gen.go:2778-2880): the exact common prefix up to the end of the block; returns the new a
    auto extend_exact = [&](int a, int b) -> int {
        for (;;) {
            const int pa = a + 8 * lane, pb = b + 8 * lane;
            const bool inb = pa + 8 <= len;
            uint64_t diff = 0;
            if (inb) diff = rd64(pa) ^ rd64(pb);
            const uint64_t oob = ballot64(!inb);
            const uint64_t dm = ballot64(inb && diff != 0);
            const int firstOob = oob ? ctz64(oob) : 64;
            if (dm) {
                const int fl = ctz64(dm);
                const uint64_t dd = rdlane64(diff, fl);
                return a + 8 * fl + (ctz64(dd) >> 3);
            }
            if (firstOob < 64) {
                const int t = a + 8 * firstOob, tb = b + 8 * firstOob;
                const int rem = len - t;  // 0..7
                const bool ne = lane < rem && rdb(t + lane) != rdb(tb + lane);
                const uint64_t nm = ballot64(ne);
                const int k = nm ? ctz64(nm) : rem;
                return t + (k < rem ? k : rem);
            }
            a += 512;
            b += 512;
        }
    };
    // number of k = 1..kmax with src[t-k] == src[s-k], consecutively (the backward extension loops)
    auto backlen = [&](int sp, int tp, int kmax) -> int {
        if (kmax <= 0 || rdb(tp - 1) != rdb(sp - 1)) return 0;  // the usual case, decided on one (wave-uniform) byte pair
        int cnt = 0;
        while (cnt < kmax) {
            const int k = cnt + lane + 1;
            bool ne = true;
            if (k <= kmax) ne = rdb(tp - k) != rdb(sp - k);
            const uint64_t m = ballot64(ne);
            const int c = m ? ctz64(m) : 64;
            cnt += c;
            if (c < 64) break;
        }
        return cnt < kmax ? cnt : kmax;
    };
    // emitCopy / emitRepeat (encode_go.go:118-234): at most 10 bytes for blocks below 16 MiB, assembled in two registers by every
    // lane (uniform), stored by lane 0 as one or two 8-byte words
    bool smallRep = false;  // the assembly's emitRepeat as generated into encodeBlockAsm8B: no two-byte offset form (kc_s2.hip)
    auto emit_copy_any = [&](int offset, int length, bool asRepeat) -> int {
        if (asRepeat && smallRep && length > 8 && length < 12) {
            if (lane == 0) { dst[d] = (uint8_t)(5 << 2 | 1); dst[d + 1] = 0; dst[d + 2] = (uint8_t)(length - 8); }
            return 3;
        }
        if (SNAPPY) { if (lane == 0) s2_emit_copy_nr1(dst + d, offset, length); return s2_copy_nr_size(offset, length); }  // (can be hundreds of 3-byte operations)
        uint64_t lo = 0, hi = 0;
        auto sink = [&](int k, uint8_t v) { if (k < 8) lo |= (uint64_t)v << (8 * k); else hi |= (uint64_t)v << (8 * (k - 8)); };
        const int n = asRepeat ? s2_put_repeat(sink, offset, length) : s2_put_copy(sink, offset, length);
        if (lane == 0) {
            st64(dst + d, lo);
            if (n > 8) st64(dst + d + 8, hi);
        }
        return n;
    };

    if (!stored) {
        int SKIP = len <= (64 << 10) ? 5 : 6;  // encodeBlockGo64K vs encodeBlockGo (encode_go.go:23-26)
        // P.variant 1: the amd64 assembly encoders' bytes (kc_s2.hip has the list of differences)
        const bool AX = P.variant == 1;
        int HSHL = 16, HSHR = 64 - S2_TABLE_BITS, LITOVH = 0;
        uint64_t HPRIME = KC_PRIME6;
        if (AX) {
            const bool top = SNAPPY ? len > 65536 : len >= (4 << 20);
            if (top || len >= (16 << 10)) { SKIP = 6; LITOVH = top ? 5 : (SNAPPY ? 3 : 4); }
            else if (len >= (4 << 10)) { SKIP = 5; HSHL = 24; HPRIME = KC_PRIME5; HSHR = 64 - 12; LITOVH = 3; }
            else if (len >= 512) { SKIP = 5; HSHL = 32; HPRIME = (uint64_t)KC_PRIME4; HSHR = 64 - 10; LITOVH = 3; }
            else { SKIP = 4; HSHL = 32; HPRIME = (uint64_t)KC_PRIME4; HSHR = 64 - 8; LITOVH = 3; smallRep = !SNAPPY; }
        }
        auto hashOf = [&](uint64_t v) -> uint32_t { return (uint32_t)(((v << HSHL) * HPRIME) >> HSHR); };
        const int sLimit = len - 8;
        const int sLimT = AX ? sLimit - 1 : sLimit;
        const int dstLimit = AX ? (len - 9) - (len >> 5) : len - (len >> 5) - 5;
        const int bailLim = AX ? dstLimit - LITOVH - 1 : dstLimit;
        const int cpLim = AX ? dstLimit - 1 : dstLimit;
        int nextEmit = 0, s = 1, repeat = 1;
        bool fin = false;  // goto emitRemainder
        const int W0 = P.spec_w0 < 1 ? 1 : (P.spec_w0 > 64 ? 64 : P.spec_w0);
        int W = W0;
        uint8_t* const tabB = (uint8_t*)tab;
        uint32_t rounds = 0;
        if (FUSED) {
            const int g4 = lane >> 4, j16 = lane & 15;
            const int off16 = j16 == 0 ? -4 : 8 * j16 - 4;  // a lane's 8 bytes inside its 16-lane group: [-4, +4) then [+4, +124) from the position
            const int soff16 = (g4 == 3 ? 1 : g4) + off16;   // groups 0..2: the candidates of s, s+1, s+2; group 3: the repeat candidate of s+1
            const int off64 = lane == 0 ? -4 : 8 * lane - 4; // the immediate re-match test: one 64-lane group, [+4, +508)
            const int q = g4 < 3 ? g4 : 0;                   // groups 0 / 1 / 2 hash the bytes at s / s+1 / s+2, every lane for itself (group 3 repeats group 0's work)
            const uint32_t hiOnly = j16 == 0 ? 0u : ~0u;     // a group's lane 0 verifies on the upper four bytes of its difference only
            const uint32_t hiOnly64 = lane == 0 ? 0u : ~0u;
            uint32_t* const sinkL = &sink[lane];
            const int lenM8 = len - 8;
            // emitLiteral in one LDS trip: lane 0 stores the tag and the literal bytes that share its 8-byte word (what it stores past a
            // short literal's end is overwritten by the next emit, which lane 0 starts too); lane k >= 1 the 8 output bytes at 8k, the
            // last of them moved back to end exactly on the literal's end — it overlaps its neighbour with the same bytes, so no lane
            // ever stores a byte that a LATER emit's other lane has to overwrite.  Literals above ~500 bytes loop.
            auto emit_lit2 = [&](int from, int n) -> int {
                if (n == 0) return 0;
                const uint32_t m = (uint32_t)(n - 1);
                int i;
                uint64_t tag;
                if (m < 60) { i = 1; tag = m << 2; }
                else if (m < (1u << 8)) { i = 2; tag = (60u << 2) | ((uint64_t)m << 8); }
                else if (m < (1u << 16)) { i = 3; tag = (61u << 2) | ((uint64_t)m << 8); }
                else if (m < (1u << 24)) { i = 4; tag = (62u << 2) | ((uint64_t)m << 8); }
                else { i = 5; tag = (63u << 2) | ((uint64_t)m << 8); }
                uint8_t* __restrict__ o = dst + d;
                const int end = i + n;  // output bytes of this emit
                for (int ob = 0; ob < end; ob += 512) {
                    int oo = ob + 8 * lane;                // this lane's 8 output bytes
                    const bool act = oo < end;
                    if (oo + 8 > end && oo != 0) oo = end - 8;
                    const bool tagw = oo == 0;             // (lane 0 of the first pass)
                    uint64_t v = rd64(from + (tagw ? 0 : oo - i));
                    if (tagw) v = tag | (v << (8 * i));
                    if (act) st64(o + oo, v);
                }
                return i + n;
            };
            // emitCopy / emitRepeat / emitCopyNoRepeat: the two- and three-byte forms (offset below 64 KiB, length up to 64 — nearly all of
            // them) in one word; the rest through the byte sinks
            auto emit_copy2 = [&](int offset, int length, bool asRepeat) -> int {
                if (SNAPPY || !asRepeat) {
                    if (length <= 64 && offset < 65536) {
                        uint32_t w;
                        int n;
                        if (length >= 12 || offset >= 2048) { w = ((uint32_t)(length - 1) << 2 | 2u) | (uint32_t)offset << 8; n = 3; }
                        else { w = ((uint32_t)(offset >> 8) << 5 | (uint32_t)(length - 4) << 2 | 1u) | ((uint32_t)offset & 0xFFu) << 8; n = 2; }
                        if (lane == 0) st32(dst + d, w);
                        return n;
                    }
                } else if (!smallRep || length <= 8 || length >= 12) {
                    const int l4 = length - 4;
                    uint32_t w = 0;
                    int n = 0;
                    if (l4 <= 4) { w = (uint32_t)l4 << 2 | 1u; n = 2; }
                    else if (l4 < 8 && offset < 2048) { w = ((uint32_t)(offset >> 8) << 5 | (uint32_t)l4 << 2 | 1u) | ((uint32_t)offset & 0xFFu) << 8; n = 2; }
                    else if (l4 < (1 << 8) + 4) { w = (5u << 2 | 1u) | (uint32_t)(l4 - 4) << 16; n = 3; }
                    if (n) {
                        if (lane == 0) st32(dst + d, w);
                        return n;
                    }
                }
                return emit_copy_any(offset, length, asRepeat);
            };
            uint64_t cvL = rd64(s + q);
#ifdef KC_S2_PROF
            unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pcnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_readcyclecounter();
#endif
            while (!stored) {
                if (++rounds > (uint32_t)len + 16u) { stored = true; break; }  // every round advances s: cannot happen; never spin on the device
                // ---------------- probe steps (encode_all.go:318-412) up to the first one that ends the scan ----------------
                uint64_t B, diff;
                uint32_t* tabL;
                int cL;
                for (;;) {
                    S2PROF(0);  // loop overhead + whatever ran since the last mark
                    const int nextS = s + ((s - nextEmit) >> SKIP) + 4;
                    if (nextS > sLimT) { fin = true; break; }
                    // trip 1, the table
                    tabL = &tab[hashOf(cvL)];
                    const uint32_t e1 = *tabL;
                    KC_WAVE_SYNC();
                    *(lane == 0 ? tabL : sinkL) = (uint32_t)s;         // table[hash0] = s
                    KC_WAVE_SYNC();
                    *(lane == 16 ? tabL : sinkL) = (uint32_t)(s + 1);  // table[hash1] = s + 1 (behind the first store: one bucket for both keeps s + 1)
                    KC_WAVE_SYNC();
                    const uint32_t e2 = *tabL;               // group 2: table[hash2] behind the two stores (:401)
                    const uint64_t cvN = rd64(nextS + q);    // the next step's bytes travel with this step's table entries
                    // trip 2: the four candidates, verification and both extensions at once
                    cL = g4 == 3 ? s + 1 - repeat : (int)((g4 == 2 ? e2 : e1) & S2L_POS_MASK);
                    diff = rd64(cL + off16) ^ rd64(s + soff16);
                    B = ballot64((((uint32_t)diff & hiOnly) | (uint32_t)(diff >> 32)) != 0u);
                    S2PROF(1);  // the probe step: hashes, table trip, candidate trip, ballot
                    if ((~B & 0x0001000100010001ull) != 0ull) break;   // a candidate verified
                    *(lane == 32 ? tabL : sinkL) = (uint32_t)(s + 2);  // table[hash2] = s + 2
                    KC_WAVE_SYNC();
                    s = nextS;
                    cvL = cvN;
                }
                if (fin) break;
                int kind, gs;  // 1 repeat at s+1, 2 match at s, 3 match at s+1, 4 match at s+2 — in the reference's order
                if (!((B >> 48) & 1ull)) { kind = 1; gs = 3; }
                else if (!(B & 1ull)) { kind = 2; gs = 0; }
                else if (!((B >> 16) & 1ull)) { kind = 3; gs = 1; }
                else { kind = 4; gs = 2; }
                // table[hash2] = s+2 is skipped when the step ends on the repeat or on the match at s
                if (kind > 2) { *(lane == 32 ? tabL : sinkL) = (uint32_t)(s + 2); }
                KC_WAVE_SYNC();
                const int p = s + (gs == 3 ? 1 : gs);
                const int gb = 16 * gs;
                const int csel = (int)rdlane32((uint32_t)cL, gb);
                int back;
                {   // backward extension: the 4 bytes in front of both positions are in the group's lane 0; beyond them (rare) bytewise
                    const uint32_t dlo = rdlane32((uint32_t)diff, gb);
                    const int kmax = csel < p - nextEmit ? csel : p - nextEmit;
                    const int nb = dlo == 0u ? 4 : (__builtin_clz(dlo) >> 3);
                    back = nb < kmax ? nb : kmax;
                    if (nb == 4 && kmax > 4) back = 4 + backlen(p - 4, csel - 4, kmax - 4);
                }
                // Forward extension.  The reference compares 8 bytes at a time from (match start + 4) while the position is <= len - 8
                // (:353-360, :441-448; the assembly: the exact common prefix): it ends on the first differing byte M, or on aK, the first
                // 8-byte position behind len - 8 in ITS phase — the match start moved by `back`, the lanes' chunks did not.  Bytes past the
                // end of the block are zero padding: a difference found there is >= len >= aK.
                auto fwd_end = [&](int ms, int pp, int cc, uint64_t fw, uint64_t dfw, int span) -> int {
                    int aK = len;
                    if (!AX) aK = ms + 4 > lenM8 ? ms + 4 : ms + 4 + 8 * (((lenM8 - ms - 4) >> 3) + 1);
                    if (fw != 0ull) {
                        const int M = pp + 4 + 8 * ctz64(fw) + (ctz64(dfw) >> 3);
                        return M < aK ? M : aK;
                    }
                    if (pp + 4 + span >= aK) return aK;
                    const int sh = (pp - ms) & 7;  // back into the reference's phase (re-reads up to 7 bytes known to be equal)
                    return AX ? extend_exact(pp + 4 + span, cc + 4 + span) : extend(pp + 4 + span - sh, cc + 4 + span - sh, lenM8);
                };
                int send;  // where the match ends
                {
                    const uint64_t fwd = (uint64_t)(((uint32_t)(B >> gb) >> 1) & 0x7FFFu);
                    const uint64_t dfw = fwd != 0ull ? rdlane64(diff, gb + 1 + ctz64(fwd)) : 0ull;
                    send = fwd_end(kind == 1 ? p : p - back, p, csel, fwd, dfw, 120);  // (the repeat's backward extension moves only its base, :349-352)
                }
                S2PROF(2);  // the match's two ends
                if (kind == 1) {
                    // ---------------- repeat at s+1 (:336-384) ----------------
                    const int base = p - back;
                    if (d + (base - nextEmit) > bailLim) { stored = true; break; }
                    s = send;
                    cvL = rd64(s + q);  // travels with the literal bytes
                    d += emit_lit2(nextEmit, base - nextEmit);
                    d += emit_copy2(repeat, s - base, nextEmit > 0);
                    nextEmit = s;
                    S2PROF(3);  // emission (repeat)
                    if (s >= sLimit) { fin = true; break; }
                    continue;
                }
                // ---------------- regular match (:387-489) ----------------
                int cand = csel - back, ms = p - back;
                if (d + (ms - nextEmit) > bailLim) { stored = true; break; }
                bool first = true;
                for (;;) {
                    const int base = ms;
                    repeat = base - cand;
                    s = send;
                    // the bytes of the immediate re-match test (lane 0: s-2, lane 1: s) and of the probe step behind it travel with the literal bytes
                    const uint64_t xL = rd64(s - 2 + (lane == 1 ? 2 : 0));
                    const uint64_t cvP = rd64(s + 1 + q);
                    if (first) { d += emit_lit2(nextEmit, base - nextEmit); first = false; }
                    d += emit_copy2(repeat, s - base, false);
                    nextEmit = s;
                    S2PROF(4);  // emission (literals + copy)
                    if (s >= sLimit) { fin = true; break; }
                    if (d > cpLim) { stored = true; break; }
                    // check for an immediate match, otherwise start the search at s+1 (:474-488)
                    uint32_t* const tabX = &tab[hashOf(xL)];
                    const uint32_t ec = *tabX;
                    KC_WAVE_SYNC();
                    *(lane == 0 ? tabX : sinkL) = (uint32_t)(s - 2);  // table[m2Hash] = s - 2
                    KC_WAVE_SYNC();
                    *(lane == 1 ? tabX : sinkL) = (uint32_t)s;        // table[currHash] = s
                    KC_WAVE_SYNC();
                    cand = (int)(rdlane32(ec, 1) & S2L_POS_MASK);
                    const uint64_t dx = rd64(cand + off64) ^ rd64(s + off64);
                    const uint64_t BX = ballot64((((uint32_t)dx & hiOnly64) | (uint32_t)(dx >> 32)) != 0u);
                    S2PROF(5);  // the immediate re-match test
                    if (BX & 1ull) { s++; cvL = cvP; break; }
                    ms = s;
                    const uint64_t fw = BX >> 1;
                    const uint64_t dfw = fw != 0ull ? rdlane64(dx, 1 + ctz64(fw)) : 0ull;
                    send = fwd_end(s, s, cand, fw, dfw, 504);
                    S2PROF(6);  // an immediate match's end
                }
                if (fin) break;
            }
#ifdef KC_S2_PROF
            if (lane == 0 && bi == 0) for (int k = 0; k < 8; k++) { kc_s2_prof[k] = pacc[k]; kc_s2_prof[8 + k] = pcnt[k]; }
#endif
        } else
        while (!fin && !stored) {
            if (++rounds > (uint32_t)len + 16u) { stored = true; break; }  // every round advances s: cannot happen; never spin on the device
            int mkind = 0, candidate = 0, ps = 0;
            if (ONESTEP) {
                // ---------------- one probe step, wave-uniform (encode_all.go:318-412) ----------------
                const int nextS = s + ((s - nextEmit) >> SKIP) + 4;
                if (nextS > sLimT) { fin = true; continue; }
                const uint64_t cv = rdlane64(rd64(s), 0);
                const uint32_t h0 = hashOf(cv), h1 = hashOf(cv >> 8), h2 = hashOf(cv >> 16);
                const uint32_t e0 = rdlane32(tab[h0], 0), e1 = rdlane32(tab[h1], 0);
                const uint32_t wr = rdlane32(rd32(s - repeat + 1), 0);
                KC_WAVE_SYNC();
                if (lane == 0) { tab[h0] = (uint32_t)s; tab[h1] = (uint32_t)(s + 1); }
                KC_WAVE_SYNC();
                const int c0 = (int)(e0 & S2L_POS_MASK), c1 = (int)(e1 & S2L_POS_MASK);
                const int c2 = (int)(rdlane32(tab[h2], 0) & S2L_POS_MASK);  // read after the s / s+1 buckets were written (encode_all.go:401)
                const uint32_t w0 = rdlane32(rd32(c0), 0), w1 = rdlane32(rd32(c1), 0), w2 = rdlane32(rd32(c2), 0);
                int kind = 0;  // 1 repeat at s+1, 2 match at s, 3 match at s+1, 4 match at s+2
                if ((uint32_t)(cv >> 8) == wr) kind = 1;
                else if ((uint32_t)cv == w0) { kind = 2; candidate = c0; }
                else if ((uint32_t)(cv >> 8) == w1) { kind = 3; candidate = c1; }
                else if ((uint32_t)(cv >> 16) == w2) { kind = 4; candidate = c2; }
                KC_WAVE_SYNC();
                // table[hash2] = s+2 is skipped when the step ends on the repeat or on the match at s
                if (lane == 0 && kind != 1 && kind != 2) tab[h2] = (uint32_t)(s + 2);
                KC_WAVE_SYNC();
                if (kind == 0) { s = nextS; continue; }
                mkind = kind;
                ps = s;
            } else {
                // ---------------- probe positions of this round: lane i = the i-th step from s ----------------
                int p;
                {
                    const int d0 = s - nextEmit, step0 = 4 + (d0 >> SKIP);
                    if (((d0 + (W - 1) * step0) >> SKIP) == (d0 >> SKIP)) {
                        p = s + lane * step0;  // all W steps inside one skip segment
                    } else {
                        p = s;
                        for (int k = 0; k + 1 < W; k++) if (k < lane) p += ((p - nextEmit) >> SKIP) + 4;
                    }
                }
                const int nextS = p + ((p - nextEmit) >> SKIP) + 4;
                const bool inW = lane < W;
                const bool valid = inW && nextS <= sLimT;  // a prefix of the lanes: nextS grows with the lane
                const bool term = inW && nextS > sLimT;    // this step would `goto emitRemainder`
                uint64_t cv = 0;
                uint32_t h0 = 0, h1 = 0, h2 = 0, e0 = 0, e1 = 0, e2 = 0;
                if (valid) {
                    cv = rd64(p);
                    h0 = hashOf(cv); h1 = hashOf(cv >> 8); h2 = hashOf(cv >> 16);
                    tabB[4 * h0 + 3] = (uint8_t)lane;
                    tabB[4 * h1 + 3] = (uint8_t)lane;
                    tabB[4 * h2 + 3] = (uint8_t)lane;
                }
                KC_WAVE_SYNC();
                if (valid) { e0 = tab[h0]; e1 = tab[h1]; e2 = tab[h2]; }
                // a lane that finds another lane's id in one of its buckets shares that bucket with it; it also tells that lane
                const uint32_t m0 = e0 >> 24, m1 = e1 >> 24, m2 = e2 >> 24;
                const bool lost = valid && (m0 != (uint32_t)lane || m1 != (uint32_t)lane || m2 != (uint32_t)lane);
                bool dep = lost;
                if (ballot64(lost) != 0) {  // rare: two steps of the round in one bucket
                    if (lane == 0) shareMask = 0ull;
                    KC_WAVE_SYNC();
                    if (lost) {
                        if (m0 != (uint32_t)lane) atomicOr(&shareMask, 1ull << m0);
                        if (m1 != (uint32_t)lane) atomicOr(&shareMask, 1ull << m1);
                        if (m2 != (uint32_t)lane) atomicOr(&shareMask, 1ull << m2);
                    }
                    KC_WAVE_SYNC();
                    dep = lost || ((shareMask >> lane) & 1ull) != 0;
                }
                if (lane == 0) dep = false;  // the first step has no earlier step to depend on: every round commits at least one step
                int kind = 0, cand = 0;  // 1 repeat at s+1, 2 match at s, 3 match at s+1, 4 match at s+2
                if (valid) {
                    // the s+2 bucket is read after the s / s+1 buckets were written (encode_all.go:401)
                    int c2 = (int)(e2 & S2L_POS_MASK);
                    if (h2 == h1) c2 = p + 1;
                    else if (h2 == h0) c2 = p;
                    const int c0 = (int)(e0 & S2L_POS_MASK), c1 = (int)(e1 & S2L_POS_MASK);
                    const uint32_t wr = rd32(p - repeat + 1);
                    const uint32_t w0 = rd32(c0), w1 = rd32(c1), w2 = rd32(c2);
                    if ((uint32_t)(cv >> 8) == wr) kind = 1;
                    else if ((uint32_t)cv == w0) { kind = 2; cand = c0; }
                    else if ((uint32_t)(cv >> 8) == w1) { kind = 3; cand = c1; }
                    else if ((uint32_t)(cv >> 16) == w2) { kind = 4; cand = c2; }
                }
                const uint64_t vm = ballot64(valid);
                const uint64_t tm = ballot64(term);
                const uint64_t depm = ballot64(dep);
                const uint64_t hm = ballot64(kind != 0);
                const int nvalid = __popcll(vm);
                const int c = depm ? ctz64(depm) : 64;
                const uint64_t hmc = c >= 64 ? hm : (hm & ((1ull << c) - 1ull));
                const bool found = hmc != 0;
                const int f = found ? ctz64(hmc) : 0;
                const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
                if (valid && lane <= commitUpTo) {
                    const bool winner = found && lane == f;
                    tab[h0] = (uint32_t)p;
                    tab[h1] = (uint32_t)(p + 1);
                    // table[hash2] = s+2 is skipped when the step ends on the repeat or on the match at s
                    if (!(winner && (kind == 1 || kind == 2))) tab[h2] = (uint32_t)(p + 2);
                }
                KC_WAVE_SYNC();
                if (!found) {
                    W = 2 * W < 64 ? 2 * W : 64;
                    if (c < nvalid) {
                        s = (int)rdlane32((uint32_t)p, c);  // the first dependent step restarts as lane 0
                    } else if (nvalid < 64 && ((tm >> nvalid) & 1ull)) {
                        fin = true;                        // the step after the last committed one hits `nextS > sLimit`
                    } else {
                        s = (int)rdlane32((uint32_t)nextS, nvalid - 1);  // nvalid >= 1: lane 0 is valid or terminates
                    }
                    continue;
                }
                W = W0;
                mkind = (int)rdlane32((uint32_t)kind, f);
                candidate = (int)rdlane32((uint32_t)cand, f);
                ps = (int)rdlane32((uint32_t)p, f);
            }
            if (mkind == 1) {
                // ---------------- repeat at s+1 (encode_all.go:336-384) ----------------
                int base = ps + 1;
                {
                    const int i0 = base - repeat;
                    int kmax = base - nextEmit;
                    if (i0 < kmax) kmax = i0;
                    base -= backlen(base, i0, kmax);
                }
                if (d + (base - nextEmit) > bailLim) { stored = true; continue; }
                d += emit_lit(nextEmit, base - nextEmit);
                s = AX ? extend_exact(ps + 4 + 1, ps - repeat + 4 + 1) : extend(ps + 4 + 1, ps - repeat + 4 + 1, sLimit);
                d += emit_copy_any(repeat, s - base, nextEmit > 0);
                nextEmit = s;
                if (s >= sLimit) fin = true;
                continue;
            }
            // ---------------- regular match (encode_all.go:387-489) ----------------
            s = ps + (mkind - 2);
            {
                int kmax = candidate;
                if (s - nextEmit < kmax) kmax = s - nextEmit;
                const int back = backlen(s, candidate, kmax);
                candidate -= back;
                s -= back;
            }
            if (d + (s - nextEmit) > bailLim) { stored = true; continue; }
            d += emit_lit(nextEmit, s - nextEmit);
            for (;;) {
                const int base = s;
                repeat = base - candidate;
                s = AX ? extend_exact(s + 4, candidate + 4) : extend(s + 4, candidate + 4, len - 8);
                d += emit_copy_any(repeat, s - base, false);
                nextEmit = s;
                if (s >= sLimit) { fin = true; break; }
                if (d > cpLim) { stored = true; break; }
                // check for an immediate match, otherwise start the search at s+1 (:474-488)
                const uint64_t x = rd64(s - 2);
                const uint32_t m2Hash = hashOf(x), currHash = hashOf(x >> 16);
                const uint32_t ec = tab[currHash];
                KC_WAVE_SYNC();
                if (lane == 0) { tab[m2Hash] = (uint32_t)(s - 2); tab[currHash] = (uint32_t)s; }
                KC_WAVE_SYNC();
                candidate = (int)(ec & S2L_POS_MASK);
                if ((uint32_t)(x >> 16) != rd32(candidate)) { s++; break; }
            }
        }
        if (!stored) {
            // emitRemainder (:491-499)
            if (AX || nextEmit < len) {  // (the assembly tests the bail-out even when nothing is left to emit)
                if (d + len - nextEmit > bailLim) stored = true;
                else if (nextEmit < len) d += emit_lit(nextEmit, len - nextEmit);
            }
        }
    }
    if (stored) {
        // the block is emitted again from the start of the slot, by other lanes than those that wrote the abandoned attempt:
        // order the two (same wave, different lanes, same addresses)
#ifndef KC_HIPEMU
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
        KC_WAVE_SYNC();
    }
    if (!P.framed) {
        if (stored) { d = 0; d = emit_lit(0, len); }  // encode.go:44-55: not compressible -> one literal
        if (lane == 0) P.out_size[bi] = (uint32_t)(hdr + d);
        return;
    }
    // ---- s2.Writer chunk (s2/writer.go:414-451) ----
    uint32_t chunkLen;
    uint8_t chunkType;
    if (stored) {  // encodeBlock returned 0: uncompressed chunk, raw copy
        const int body = len & ~7;
        for (int k = lane * 8; k < body; k += 512) st64(out + k, rd64(k));
        for (int k = body + lane; k < len; k += 64) out[k] = (uint8_t)rdb(k);
        chunkType = 0x01;
        chunkLen = 4u + (uint32_t)len;
    } else {
        chunkType = 0x00;
        chunkLen = 4u + (uint32_t)hdr + (uint32_t)d;
    }
    const uint32_t cc = s2_crc32c_wave(rd32, rdb, len, crcT, crcM, crcP, lane);
    if (lane == 0) {
        const uint32_t checksum = ((cc >> 15) | (cc << 17)) + 0xa282ead8u;
        slot[0] = chunkType;
        slot[1] = (uint8_t)chunkLen; slot[2] = (uint8_t)(chunkLen >> 8); slot[3] = (uint8_t)(chunkLen >> 16);
        slot[4] = (uint8_t)checksum; slot[5] = (uint8_t)(checksum >> 8); slot[6] = (uint8_t)(checksum >> 16); slot[7] = (uint8_t)(checksum >> 24);
        P.out_size[bi] = 4u + chunkLen;
    }
}

void kc_launch_s2_encode_lds(const KcS2Params& P, bool any_small, bool any_big, hipStream_t st) {
    if (P.n_blocks == 0) return;
    // blocks held in LDS: spec_w0 0 (the default) the fused step, 1 one wave-uniform step at a time in its first form, above that speculative rounds
    const int mode = P.spec_w0 <= 0 ? 2 : (P.spec_w0 == 1 ? 1 : 0);
    if (P.level == 2) {
        if (any_small && mode == 2) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<2, true, 2>), dim3(P.n_blocks), dim3(64), 0, st, P);
        if (any_small && mode == 1) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<2, true, 1>), dim3(P.n_blocks), dim3(64), 0, st, P);
        if (any_small && mode == 0) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<2, true, 0>), dim3(P.n_blocks), dim3(64), 0, st, P);
        if (any_big) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<2, false, 0>), dim3(P.n_blocks), dim3(64), 0, st, P);
    } else {
        if (any_small && mode == 2) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<0, true, 2>), dim3(P.n_blocks), dim3(64), 0, st, P);
        if (any_small && mode == 1) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<0, true, 1>), dim3(P.n_blocks), dim3(64), 0, st, P);
        if (any_small && mode == 0) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<0, true, 0>), dim3(P.n_blocks), dim3(64), 0, st, P);
        if (any_big) hipLaunchKernelGGL((kc_s2_encode_lds_kernel<0, false, 0>), dim3(P.n_blocks), dim3(64), 0, st, P);
    }
}
