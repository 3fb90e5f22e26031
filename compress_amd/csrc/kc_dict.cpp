// kc_dict.cpp — host-side loader for full-format zstd dictionaries (WithEncoderDict).
//
// Replaces loadDict (zstd/dict.go:71-150) + huff0.ReadTable (huff0/decompress.go:29-168) as far as the ENCODER
// uses their result: dictionary ID, the literal Huffman code as huff0 "prevTable" (blockenc.go:518-522), the three
// repeat offsets (enc_base.go:189-195) and the content (initial history).  The three FSE tables of the dictionary
// are only consumed by decoders; they are parsed here to find where they end and to reject malformed input.
//
// Written against the Zstandard format description (Dictionary_Format, Huffman_Tree_Description, FSE_Table_Description)
// with a forward bit cursor for table descriptions and a backward one for the FSE-compressed weights.
#include <stdint.h>
#include <string.h>

#include "../../include/kcgpu.h"

namespace {

struct FwdBits {  // little-endian bit cursor, zero padded past the end
    const uint8_t* p;
    size_t n;
    size_t bit = 0;
    uint32_t peek(int nb) const {
        uint64_t v = 0;
        const size_t b0 = bit >> 3;
        for (int k = 0; k < 5; k++)
            if (b0 + (size_t)k < n) v |= (uint64_t)p[b0 + (size_t)k] << (8 * k);
        return (uint32_t)((v >> (bit & 7)) & ((nb >= 32) ? 0xFFFFFFFFull : ((1ull << nb) - 1ull)));
    }
    uint32_t take(int nb) { const uint32_t v = peek(nb); bit += (size_t)nb; return v; }
    bool overrun() const { return bit > n * 8; }
    size_t bytes() const { return (bit + 7) >> 3; }
};

inline int hibit(uint32_t v) { return 31 - __builtin_clz(v); }

// FSE_Table_Description -> normalized counts.  Returns bytes consumed, 0 on error.
size_t read_ncount(const uint8_t* p, size_t n, int maxSym, int maxLog, int16_t* norm, int* nSym, int* tableLog) {
    FwdBits b{p, n};
    const int tl = (int)b.take(4) + 5;
    if (tl > maxLog) return 0;
    int remaining = 1 << tl;
    int sym = 0;
    while (remaining > 0 && sym <= maxSym) {
        const int maxv = remaining + 1;
        const int bits = hibit((uint32_t)maxv) + 1;
        const int lowThreshold = (1 << bits) - 1 - maxv;
        int v = (int)b.peek(bits - 1);
        if (v < lowThreshold) {
            b.bit += (size_t)(bits - 1);
        } else {
            v = (int)b.take(bits);
            if (v >= (1 << (bits - 1))) v -= lowThreshold;
        }
        const int prob = v - 1;
        norm[sym++] = (int16_t)prob;
        remaining -= prob < 0 ? 1 : prob;
        if (prob == 0) {
            for (;;) {
                const int rep = (int)b.take(2);
                for (int k = 0; k < rep && sym <= maxSym; k++) norm[sym++] = 0;
                if (rep != 3) break;
                if (b.overrun()) return 0;
            }
        }
        if (b.overrun()) return 0;
    }
    if (remaining != 0 || sym <= 1 || sym > maxSym + 1) return 0;
    *nSym = sym;
    *tableLog = tl;
    return b.bytes();
}

struct DState { uint8_t sym, nb; uint16_t base; };

bool build_dtable(const int16_t* norm, int nSym, int tl, DState* dt) {
    const int size = 1 << tl;
    uint16_t next[256];
    int high = size - 1;
    for (int s = 0; s < nSym; s++) {
        if (norm[s] == -1) { dt[high--].sym = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nSym; s++)
        for (int k = 0; k < norm[s]; k++) {
            dt[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    if (pos != 0) return false;
    for (int u = 0; u < size; u++) {
        const uint16_t nx = next[dt[u].sym]++;
        if (nx == 0) return false;
        const int nb = tl - hibit(nx);
        dt[u].nb = (uint8_t)nb;
        dt[u].base = (uint16_t)((nx << nb) - size);
    }
    return true;
}

// FSE-compressed Huffman weights: two interleaved states over a backward bitstream.  Returns the weight count, -1 on error.
int fse_decode_weights(const uint8_t* p, size_t n, uint8_t* out, int cap) {
    int16_t norm[256];
    int nSym = 0, tl = 0;
    const size_t hdr = read_ncount(p, n, 255, 12, norm, &nSym, &tl);
    if (hdr == 0 || hdr >= n) return -1;
    static thread_local DState dt[1 << 12];
    if (!build_dtable(norm, nSym, tl, dt)) return -1;
    const uint8_t* s = p + hdr;
    const size_t sn = n - hdr;
    if (s[sn - 1] == 0) return -1;
    long off = (long)sn * 8 - (8 - hibit(s[sn - 1]));  // bits available below the end mark
    auto rd = [&](int nb) -> uint32_t {                // read nb bits ending at `off`, zero-filled below bit 0
        off -= nb;
        uint32_t v = 0;
        for (int k = 0; k < nb; k++) {
            const long bp = off + k;
            if (bp >= 0 && ((s[bp >> 3] >> (bp & 7)) & 1)) v |= 1u << k;
        }
        return v;
    };
    uint32_t s1 = rd(tl), s2 = rd(tl);
    if (off < 0) return -1;
    int w = 0;
    // fse/decompress.go:310-330: a state may only run dry exactly at the end of the stream — a read that starts with bits left
    // and needs more than remain is an over-read (bitReader.close -> io.ErrUnexpectedEOF), not the end.
    for (;;) {
        if (w + 2 > cap) return -1;
        out[w++] = dt[s1].sym;
        if (off > 0 && off < (long)dt[s1].nb) return -1;
        s1 = dt[s1].base + rd(dt[s1].nb);
        if (off < 0) { out[w++] = dt[s2].sym; break; }
        if (w + 2 > cap) return -1;
        out[w++] = dt[s2].sym;
        if (off > 0 && off < (long)dt[s2].nb) return -1;
        s2 = dt[s2].base + rd(dt[s2].nb);
        if (off < 0) { out[w++] = dt[s1].sym; break; }
    }
    return w;
}

}  // namespace

extern "C" int kc_zstd_opts_dict(kc_zstd_opts* o, const uint8_t* blob, uint64_t len) {
    if (o == nullptr || blob == nullptr) return -1;
    if (len <= 8 + 12) return -1;                                                   // dict.go:73
    if (!(blob[0] == 0x37 && blob[1] == 0xA4 && blob[2] == 0x30 && blob[3] == 0xEC)) return -1;  // dictMagic
    uint32_t id;
    memcpy(&id, blob + 4, 4);
    if (id == 0) return -1;  // "dictionaries cannot have ID 0"
    const uint8_t* p = blob + 8;
    size_t n = (size_t)len - 8;

    // ---- Huffman_Tree_Description -> weights ----
    uint8_t wt[256];
    memset(wt, 0, sizeof(wt));
    int nw = 0;
    if (n <= 1) return -1;
    const int hb = p[0];
    p++; n--;
    if (hb >= 128) {
        nw = hb - 127;
        const size_t nbytes = (size_t)(nw + 1) / 2;
        if (nbytes > n) return -1;
        for (int k = 0; k < nw; k++) wt[k] = (k & 1) ? (p[k >> 1] & 15) : (p[k >> 1] >> 4);
        p += nbytes; n -= nbytes;
    } else {
        if ((size_t)hb > n || hb == 0) return -1;
        nw = fse_decode_weights(p, (size_t)hb, wt, 255);
        if (nw <= 0) return -1;
        p += hb; n -= (size_t)hb;
    }
    // implied last weight; canonical code values in the reference's order (rank start >> (w-1), symbols ascending)
    uint32_t rankCount[16] = {0};
    uint32_t total = 0;
    for (int k = 0; k < nw; k++) {
        if (wt[k] > 11) return -1;
        rankCount[wt[k]]++;
        total += (1u << wt[k]) >> 1;
    }
    if (total == 0) return -1;
    const int tableLog = hibit(total) + 1;
    if (tableLog > 11) return -1;
    const uint32_t rest = (1u << tableLog) - total;
    if (rest == 0 || (rest & (rest - 1)) != 0) return -1;  // last weight must be a clean power of two
    const int lastW = hibit(rest) + 1;
    wt[nw++] = (uint8_t)lastW;
    rankCount[lastW]++;
    if (rankCount[1] < 2 || (rankCount[1] & 1)) return -1;
    uint32_t rankStart[16] = {0};
    {
        uint32_t nxt = 0;
        for (int r = 1; r <= tableLog; r++) { rankStart[r] = nxt; nxt += rankCount[r] << (r - 1); }
    }
    memset(o->dict_huf_val, 0, sizeof(o->dict_huf_val));
    memset(o->dict_huf_nbits, 0, sizeof(o->dict_huf_nbits));
    for (int s = 0; s < nw; s++) {
        const int w = wt[s];
        if (w == 0) continue;
        o->dict_huf_val[s] = (uint16_t)(rankStart[w] >> (w - 1));
        o->dict_huf_nbits[s] = (uint8_t)(tableLog + 1 - w);
        rankStart[w] += (1u << w) >> 1;
    }
    o->dict_huf_len = nw;
    o->dict_huf_log = tableLog;

    // ---- three FSE table descriptions: offsets (maxOffsetLengthSymbol = 30, zstd/fse_predefined.go:45), match lengths (52), literal lengths (35) ----
    const int maxSyms[3] = {30, 52, 35};
    for (int t = 0; t < 3; t++) {
        int16_t norm[256];
        int ns = 0, tl = 0;
        if (n < 4) return -1;
        const size_t used = read_ncount(p, n, maxSyms[t], 9, norm, &ns, &tl);
        if (used == 0 || used > n) return -1;
        static thread_local DState dt[1 << 9];
        if (!build_dtable(norm, ns, tl, dt)) return -1;
        p += used; n -= used;
    }
    if (n < 12) return -1;
    uint32_t offs[3];
    memcpy(offs, p, 12);
    p += 12; n -= 12;
    for (int k = 0; k < 3; k++) {
        if (offs[k] == 0 || (int32_t)offs[k] < 0) return -1;  // "invalid offset in dictionary"
        if ((uint64_t)offs[k] > (uint64_t)n) return -1;       // "initial offset bigger than dictionary content size"
        o->dict_offsets[k] = offs[k];
    }
    o->dict_id = id;
    o->dict = p;
    o->dict_len = n;
    return 0;
}
