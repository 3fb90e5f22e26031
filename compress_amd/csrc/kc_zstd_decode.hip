// kc_zstd_decode.hip — zstd frame decoder on the device: the verifier half of SURVEY.md §8f N1 for zstd
// (zstd/framedec.go:65-330 -> blockdec.go:227-690 -> seqdec_generic.go:16,161, fse_decoder.go, huff0/decompress.go).
// One wave per frame.  Serial format parsing (headers, FSE/Huffman table descriptions, the sequence bitstream) runs on lane 0
// with the tables in LDS; Huffman streams decode on one lane per stream; literal and match copies use all 64 lanes, 64
// decoded sequences at a time.  Raw-content dictionaries are supported as history; dictionary entropy tables are not (status 8/12).  Not a throughput kernel: it exists so that a
// device-resident encode can be verified (decode + XXH64 compare) without leaving the GPU.
#include "kc_dev.h"
#include "kc_kernels.h"

namespace {

struct ZdSym { uint16_t base; uint8_t sym; uint8_t nb; };  // FSE decoding cell: newState base, symbol, bits to read

struct ZdShared {
    uint16_t huf[1 << 11];  // symbol << 8 | nBits, index = next tableLog bits
    ZdSym ll[1 << 9], of[1 << 8], ml[1 << 9];
    ZdSym wt[1 << 7];       // FSE table of the Huffman weights
    uint8_t weights[256];
    int16_t norm[64];
    uint16_t next[64];
    uint32_t seqLL[64], seqML[64], seqOF[64];
    int iv[16];
};
enum { V_ERR = 0, V_HUFLOG, V_LLLOG, V_OFLOG, V_MLLOG, V_LLOK, V_OFOK, V_MLOK, V_HUFOK, V_NBATCH };

__device__ __forceinline__ int zd_hibit(uint32_t v) { return 31 - __builtin_clz(v); }

// backward bit reader (zstd/bitreader.go) over global memory: `pos` = unread bits
struct ZdRBits {
    const uint8_t* p;
    long pos;
    __device__ bool init(const uint8_t* d, int n) {
        if (n <= 0 || d[n - 1] == 0) return false;
        p = d;
        pos = (long)n * 8 - (8 - zd_hibit(d[n - 1]));
        return true;
    }
    __device__ uint32_t peek(int nb) const {  // next nb (<= 24) bits, most significant first, zeros below bit 0
        if (nb == 0) return 0;
        const long lo = pos - nb;  // lowest bit index wanted (may be negative)
        uint64_t w = 0;
        const long b0 = (lo < 0 ? 0 : lo) >> 3;
        for (int k = 0; k < 5; k++) {
            const long bi = b0 + k;
            if (bi * 8 < pos) w |= (uint64_t)p[bi] << (8 * k);
        }
        if (lo >= 0) return (uint32_t)((w >> (lo & 7)) & ((1u << nb) - 1u));
        const int have = (int)pos;  // fewer than nb bits left: they are the high part, zeros fill the rest
        if (have <= 0) return 0;
        return (uint32_t)((w & ((1ull << have) - 1ull)) << (nb - have)) & ((1u << nb) - 1u);
    }
    __device__ uint32_t read(int nb) { const uint32_t v = peek(nb); pos -= nb; return v; }
};

// forward bit cursor for FSE table descriptions (zero padded past the end)
struct ZdFBits {
    const uint8_t* p;
    int n;
    int bit;
    __device__ uint32_t peek(int nb) const {
        uint64_t v = 0;
        const int b0 = bit >> 3;
        for (int k = 0; k < 5; k++) if (b0 + k < n) v |= (uint64_t)p[b0 + k] << (8 * k);
        return (uint32_t)((v >> (bit & 7)) & ((1ull << nb) - 1ull));
    }
    __device__ uint32_t take(int nb) { const uint32_t v = peek(nb); bit += nb; return v; }
};

// FSE_Table_Description -> norm[] (fse_decoder.go:52-184).  Returns bytes consumed, 0 on error.  Lane 0 only.
__device__ int zd_read_ncount(const uint8_t* p, int n, int maxSym, int maxLog, int16_t* norm, int* nSym, int* tableLog) {
    if (n < 1) return 0;
    ZdFBits b{p, n, 0};
    const int tl = (int)b.take(4) + 5;
    if (tl > maxLog) return 0;
    int remaining = 1 << tl, sym = 0;
    while (remaining > 0 && sym <= maxSym) {
        const int maxv = remaining + 1;
        const int bits = zd_hibit((uint32_t)maxv) + 1;
        const int lowThreshold = (1 << bits) - 1 - maxv;
        int v = (int)b.peek(bits - 1);
        if (v < lowThreshold) b.bit += bits - 1;
        else { v = (int)b.take(bits); if (v >= (1 << (bits - 1))) v -= lowThreshold; }
        const int prob = v - 1;
        norm[sym++] = (int16_t)prob;
        remaining -= prob < 0 ? 1 : prob;
        if (prob == 0) {
            for (;;) {
                const int rep = (int)b.take(2);
                for (int k = 0; k < rep && sym <= maxSym; k++) norm[sym++] = 0;
                if (rep != 3) break;
                if (b.bit > n * 8) return 0;
            }
        }
        if (b.bit > n * 8) return 0;
    }
    if (remaining != 0 || sym <= 1) return 0;
    *nSym = sym;
    *tableLog = tl;
    return (b.bit + 7) >> 3;
}

// fse_decoder.go buildDtable: norm -> decoding cells.  Lane 0 only.
__device__ bool zd_build_fse(const int16_t* norm, int nSym, int tl, ZdSym* dt, uint16_t* next) {
    const int size = 1 << tl;
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
        const int nb = tl - zd_hibit(nx);
        dt[u].nb = (uint8_t)nb;
        dt[u].base = (uint16_t)((nx << nb) - size);
    }
    return true;
}

__constant__ int16_t kLLNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t kOFNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__constant__ int16_t kMLNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512,
                                     1024, 2048, 4096, 8192, 16384, 32768, 65536};
__constant__ uint32_t kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32,
                                     33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};

// One sequence table according to its mode (blockdec.go:560-640).  Lane 0 only.  Returns bytes consumed (>= 0) or -1.
__device__ int zd_seq_table(int mode, int kind, const uint8_t* p, int n, ZdShared& S) {
    ZdSym* dt = kind == 0 ? S.ll : (kind == 1 ? S.of : S.ml);
    const int maxSym = kind == 0 ? 35 : (kind == 1 ? 30 : 52);  // maxOffsetLengthSymbol = 30 (zstd/fse_predefined.go:45)
    const int maxLog = kind == 1 ? 8 : 9;
    if (mode == 0) {
        const int16_t* src = kind == 0 ? kLLNorm : (kind == 1 ? kOFNorm : kMLNorm);
        const int ns = kind == 0 ? 36 : (kind == 1 ? 29 : 53);
        const int tl = kind == 1 ? 5 : 6;
        for (int i = 0; i < ns; i++) S.norm[i] = src[i];
        if (!zd_build_fse(S.norm, ns, tl, dt, S.next)) return -1;
        S.iv[V_LLLOG + kind] = tl;
        S.iv[V_LLOK + kind] = 1;
        return 0;
    }
    if (mode == 1) {
        if (n < 1 || p[0] > maxSym) return -1;
        dt[0].sym = p[0]; dt[0].nb = 0; dt[0].base = 0;
        S.iv[V_LLLOG + kind] = 0;
        S.iv[V_LLOK + kind] = 1;
        return 1;
    }
    if (mode == 2) {
        int ns = 0, tl = 0;
        const int used = zd_read_ncount(p, n, maxSym, maxLog, S.norm, &ns, &tl);
        if (used == 0 || used > n) return -1;
        if (!zd_build_fse(S.norm, ns, tl, dt, S.next)) return -1;
        S.iv[V_LLLOG + kind] = tl;
        S.iv[V_LLOK + kind] = 1;
        return used;
    }
    return S.iv[V_LLOK + kind] ? 0 : -1;  // repeat
}

// FSE-compressed Huffman weights (huff0/decompress.go:57-70 -> fse.Decompress).  Lane 0.  Returns count or -1.
__device__ int zd_fse_weights(const uint8_t* p, int n, ZdShared& S, uint8_t* out) {
    int ns = 0, tl = 0;
    const int hdr = zd_read_ncount(p, n, 255 > 63 ? 63 : 255, 7, S.norm, &ns, &tl);  // weights are < 16: 64 norm slots are plenty
    if (hdr == 0 || hdr >= n) return -1;
    ZdSym* dt = S.wt;  // the sequence tables must survive: a later block may use them in repeat mode
    if (!zd_build_fse(S.norm, ns, tl, dt, S.next)) return -1;
    ZdRBits br;
    if (!br.init(p + hdr, n - hdr)) return -1;
    uint32_t s1 = br.read(tl), s2 = br.read(tl);
    if (br.pos < 0) return -1;
    int w = 0;
    for (;;) {
        if (w + 2 > 255) return -1;
        out[w++] = dt[s1].sym;
        s1 = dt[s1].base + br.read(dt[s1].nb);
        if (br.pos < 0) { out[w++] = dt[s2].sym; break; }
        if (w + 2 > 255) return -1;
        out[w++] = dt[s2].sym;
        s2 = dt[s2].base + br.read(dt[s2].nb);
        if (br.pos < 0) { out[w++] = dt[s1].sym; break; }
    }
    return w;
}

}  // namespace

__global__ __launch_bounds__(64) void kc_zstd_decode_kernel(KcZstdDecParams P) {
    __shared__ ZdShared S;
    const int lane = (int)threadIdx.x;
    const uint32_t u = blockIdx.x;
    if (u >= P.n_units) return;
    const uint8_t* __restrict__ in = P.enc + P.enc_off[u];
    const int n = (int)(P.enc_off[u + 1] - P.enc_off[u]);
    uint8_t* __restrict__ out = P.dst + P.dst_off[u];
    const uint64_t want = P.dst_off[u + 1] - P.dst_off[u];
    uint8_t* __restrict__ lits = P.lits + (size_t)u * P.lit_stride;
    if (lane < 16) S.iv[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    int err = 0;
    int p = 0;
    bool checksum = false;
    // ---- frame header (framedec.go:65-200), identical on all lanes ----
    if (n == 0) {  // nothing was emitted for an empty unit (WithZeroFrames(false))
        if (lane == 0) { P.status[u] = want == 0 ? 0u : 2u; P.crc_stored[u] = 0xFFFFFFFFu; P.has_crc[u] = 0u; }
        return;
    }
    if (n < 6 || ld32(in) != 0xFD2FB528u) err = 1;
    uint64_t fcs = 0;
    int fcsSize = 0;
    if (!err) {
        const uint8_t fhd = in[4];
        p = 5;
        const bool single = (fhd >> 5) & 1;
        checksum = (fhd >> 2) & 1;
        if (fhd & 8) err = 1;
        if (!single) p++;
        const int dsz = (fhd & 3) == 3 ? 4 : (fhd & 3);
        if (dsz && P.dict == nullptr) err = 20;  // a dictionary frame needs the dictionary content (raw content only: no entropy tables)
        p += dsz;
        fcsSize = (fhd >> 6) == 0 ? (single ? 1 : 0) : (1 << (fhd >> 6));
        if (p + fcsSize > n) err = 1;
        else {
            for (int k = 0; k < fcsSize; k++) fcs |= (uint64_t)in[p + k] << (8 * k);
            if (fcsSize == 2) fcs += 256;
            p += fcsSize;
        }
        if (!err && fcsSize > 0 && fcs != want) err = 2;
    }
    uint64_t d = 0;  // bytes produced
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    bool last = false;
    while (!err && !last) {
        if (p + 3 > n) { err = 3; break; }
        const uint32_t bh = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
        p += 3;
        last = bh & 1;
        const int type = (bh >> 1) & 3;
        const int size = (int)(bh >> 3);
        if (type == 0) {  // raw
            if (p + size > n || d + (uint64_t)size > want) { err = 4; break; }
            for (int k = lane; k < size; k += 64) out[d + k] = in[p + k];
            d += (uint64_t)size; p += size;
            continue;
        }
        if (type == 1) {  // RLE
            if (p + 1 > n || d + (uint64_t)size > want) { err = 4; break; }
            const uint8_t v = in[p];
            for (int k = lane; k < size; k += 64) out[d + k] = v;
            d += (uint64_t)size; p += 1;
            continue;
        }
        if (type == 3 || p + size > n || size > (128 << 10)) { err = 5; break; }
        // ================= compressed block =================
        const uint8_t* b = in + p;
        const int bn = size;
        p += size;
        // ---- literals section (blockdec.go:275-460) ----
        if (bn < 2) { err = 6; break; }  // ErrBlockTooSmall (blockdec.go:233): nothing below is read past the block
        const int ltype = b[0] & 3, sf = (b[0] >> 2) & 3;
        {
            const int need = ltype < 2 ? ((sf & 1) == 0 ? 1 : (sf == 1 ? 2 : 3)) : (sf < 2 ? 3 : (sf == 2 ? 4 : 5));
            if (need > bn) { err = 6; break; }
        }
        int hdr = 0, regen = 0, comp = 0;
        bool four = false;
        const uint8_t* L = nullptr;  // where the literals of this block can be read
        int litRle = -1;
        if (ltype < 2) {
            if ((sf & 1) == 0) { hdr = 1; regen = b[0] >> 3; }
            else if (sf == 1) { hdr = 2; regen = (b[0] >> 4) | ((int)b[1] << 4); }
            else { hdr = 3; regen = (b[0] >> 4) | ((int)b[1] << 4) | ((int)b[2] << 12); }
            if (ltype == 0) { if (hdr + regen > bn) { err = 6; break; } L = b + hdr; comp = regen; }
            else { if (hdr + 1 > bn) { err = 6; break; } litRle = b[hdr]; comp = 1; }
        } else {
            if (sf < 2) { const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16); hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; four = sf == 1; }
            else if (sf == 2) { const uint32_t v = ld32(b); hdr = 4; regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; four = true; }
            else { const uint64_t v = (uint64_t)ld32(b) | ((uint64_t)b[4] << 32); hdr = 5; regen = (int)((v >> 4) & 0x3FFFF); comp = (int)((v >> 22) & 0x3FFFF); four = true; }
            if (hdr + comp > bn || regen > (int)P.lit_stride) { err = 6; break; }
            const uint8_t* q = b + hdr;
            int left = comp;
            if (ltype == 2) {
                // Huffman_Tree_Description (huff0/decompress.go:29-168): weights on lane 0, table fill on all lanes
                if (lane == 0) {
                    int e2 = 0, used = 0, nw = 0;
                    const int hb = left > 0 ? q[0] : 0;
                    if (left < 2) e2 = 7;
                    else if (hb >= 128) {
                        nw = hb - 127;
                        used = 1 + (nw + 1) / 2;
                        if (used > left) e2 = 7;
                        else for (int k = 0; k < nw; k++) S.weights[k] = (k & 1) ? (q[1 + (k >> 1)] & 15) : (q[1 + (k >> 1)] >> 4);
                    } else {
                        used = 1 + hb;
                        if (hb == 0 || used > left) e2 = 7;
                        else { nw = zd_fse_weights(q + 1, hb, S, S.weights); if (nw <= 0) e2 = 7; }
                    }
                    int tableLog = 0;
                    if (!e2) {
                        uint32_t total = 0;
                        for (int k = 0; k < nw; k++) { if (S.weights[k] > 11) e2 = 7; total += (1u << S.weights[k]) >> 1; }
                        if (!e2 && total == 0) e2 = 7;
                        if (!e2) {
                            tableLog = zd_hibit(total) + 1;
                            const uint32_t rest = (1u << tableLog) - total;
                            if (tableLog > 11 || rest == 0 || (rest & (rest - 1)) != 0) e2 = 7;
                            else { S.weights[nw++] = (uint8_t)(zd_hibit(rest) + 1); for (int k = nw; k < 256; k++) S.weights[k] = 0; }
                        }
                    }
                    S.iv[V_HUFLOG] = tableLog;
                    S.iv[V_HUFOK] = e2 ? 0 : 1;
                    S.iv[V_ERR] = e2;
                    S.iv[V_NBATCH] = used;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (S.iv[V_ERR]) { err = S.iv[V_ERR]; break; }
                const int used = S.iv[V_NBATCH];
                const int tableLog = S.iv[V_HUFLOG];
                // start of each symbol's cell range: cells are ordered by (weight asc, symbol asc)
                for (int s0 = 0; s0 < 256; s0 += 64) {
                    const int sy = s0 + lane;
                    const int w = S.weights[sy];
                    if (w) {
                        uint32_t start = 0;
                        for (int t = 0; t < 256; t++) {
                            const int wt = S.weights[t];
                            if (wt && (wt < w || (wt == w && t < sy))) start += (1u << wt) >> 1;
                        }
                        const uint32_t len = (1u << w) >> 1;
                        const uint16_t e = (uint16_t)((sy << 8) | (tableLog + 1 - w));
                        for (uint32_t k = 0; k < len; k++) S.huf[start + k] = e;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                q += used; left -= used;
            } else if (!S.iv[V_HUFOK]) { err = 8; break; }
            // streams: one lane each (decompress.go Decompress1X / Decompress4X)
            const int hlog = S.iv[V_HUFLOG];
            int sOff[4] = {0, 0, 0, 0}, sLen[4] = {left, 0, 0, 0}, oOff[4] = {0, 0, 0, 0}, oLen[4] = {regen, 0, 0, 0};
            int nstreams = 1;
            if (four) {
                if (left < 6) { err = 9; break; }
                const int s1 = q[0] | (q[1] << 8), s2 = q[2] | (q[3] << 8), s3 = q[4] | (q[5] << 8);
                if (6 + s1 + s2 + s3 > left) { err = 9; break; }
                const int seg = (regen + 3) / 4;
                if (seg * 3 > regen) { err = 9; break; }
                sOff[0] = 6; sLen[0] = s1; sOff[1] = 6 + s1; sLen[1] = s2; sOff[2] = 6 + s1 + s2; sLen[2] = s3;
                sOff[3] = 6 + s1 + s2 + s3; sLen[3] = left - sOff[3];
                for (int k = 0; k < 4; k++) { oOff[k] = k * seg; oLen[k] = k < 3 ? seg : regen - 3 * seg; }
                nstreams = 4;
            }
            int serr = 0;
            if (lane < nstreams) {
                ZdRBits br;
                if (!br.init(q + sOff[lane], sLen[lane])) serr = 10;
                else {
                    uint8_t* o = lits + oOff[lane];
                    for (int i = 0; i < oLen[lane]; i++) {
                        const uint16_t e = S.huf[br.peek(hlog)];
                        o[i] = (uint8_t)(e >> 8);
                        br.pos -= (e & 0xFF);
                    }
                    if (br.pos != 0) serr = 10;
                }
            }
            if (__ballot(serr != 0)) { err = 10; break; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            L = lits;
        }
        // ---- sequences section (blockdec.go:505-690) ----
        const uint8_t* sp = b + hdr + comp;
        int sn = bn - hdr - comp;
        if (sn < 1) { err = 11; break; }
        int nSeq = sp[0];
        int sh = 1;
        if (nSeq >= 128) {
            if (nSeq < 255) { if (sn < 2) { err = 11; break; } nSeq = ((nSeq - 128) << 8) + sp[1]; sh = 2; }
            else { if (sn < 3) { err = 11; break; } nSeq = sp[1] + (sp[2] << 8) + 0x7F00; sh = 3; }
        }
        sp += sh; sn -= sh;
        if (nSeq == 0) {
            if (sn != 0 || d + (uint64_t)regen > want) { err = 11; break; }
            if (litRle >= 0) { for (int k = lane; k < regen; k += 64) out[d + k] = (uint8_t)litRle; }
            else { for (int k = lane; k < regen; k += 64) out[d + k] = L[k]; }
            d += (uint64_t)regen;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            continue;
        }
        if (lane == 0) {
            int e2 = 0;
            int used = 0;
            if (sn < 1) e2 = 12;
            else {
                const uint8_t modes = sp[0];
                if (modes & 3) e2 = 12;
                int q2 = 1;
                for (int kind = 0; kind < 3 && !e2; kind++) {
                    const int mode = (modes >> (6 - 2 * kind)) & 3;
                    const int r = zd_seq_table(mode, kind, sp + q2, sn - q2, S);
                    if (r < 0) e2 = 12; else q2 += r;
                }
                used = q2;
            }
            S.iv[V_ERR] = e2;
            S.iv[V_NBATCH] = used;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (S.iv[V_ERR]) { err = S.iv[V_ERR]; break; }
        sp += S.iv[V_NBATCH]; sn -= S.iv[V_NBATCH];
        // decode 64 sequences on lane 0, then execute them on all lanes (seqdec_generic.go)
        ZdRBits br;
        uint32_t llS = 0, ofS = 0, mlS = 0;
        bool brOk = true;
        if (lane == 0) {
            brOk = br.init(sp, sn);
            if (brOk) {
                llS = br.read(S.iv[V_LLLOG]); ofS = br.read(S.iv[V_OFLOG]); mlS = br.read(S.iv[V_MLLOG]);
                if (br.pos < 0) brOk = false;
            }
        }
        if (__ballot(lane == 0 && !brOk)) { err = 13; break; }
        int lp = 0;  // literals consumed
        for (int s0 = 0; s0 < nSeq && !err; s0 += 64) {
            const int cnt = nSeq - s0 < 64 ? nSeq - s0 : 64;
            if (lane == 0) {
                int e2 = 0;
                for (int i = 0; i < cnt && !e2; i++) {
                    const ZdSym cl = S.ll[llS], co = S.of[ofS], cm = S.ml[mlS];
                    if (cl.sym > 35 || cm.sym > 52 || co.sym > 30) { e2 = 14; break; }
                    uint32_t ofVal;
                    if (co.sym <= 24) ofVal = (1u << co.sym) + br.read(co.sym);
                    else { const uint32_t hi = br.read(co.sym - 16); const uint32_t lo = br.read(16); ofVal = (1u << co.sym) + ((hi << 16) | lo); }
                    const uint32_t mlen = kMLBase[cm.sym] + br.read(kMLBits[cm.sym]);
                    const uint32_t llen = kLLBase[cl.sym] + br.read(kLLBits[cl.sym]);
                    uint32_t off;
                    if (ofVal > 3) { off = ofVal - 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
                    else {
                        const uint32_t idx = ofVal + (llen == 0 ? 1u : 0u);
                        if (idx == 1) off = rep0;
                        else {
                            off = idx == 4 ? rep0 - 1 : (idx == 2 ? rep1 : rep2);
                            if (off == 0) { e2 = 15; break; }
                            if (idx != 2) rep2 = rep1;
                            rep1 = rep0;
                            rep0 = off;
                        }
                    }
                    if (s0 + i + 1 < nSeq) {
                        llS = cl.base + br.read(cl.nb);
                        mlS = cm.base + br.read(cm.nb);
                        ofS = co.base + br.read(co.nb);
                    }
                    if (br.pos < 0) { e2 = 16; break; }
                    S.seqLL[i] = llen; S.seqML[i] = mlen; S.seqOF[i] = off;
                }
                if (!e2 && s0 + cnt >= nSeq && br.pos != 0) e2 = 17;  // "extra bits on stream"
                S.iv[V_ERR] = e2;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (S.iv[V_ERR]) { err = S.iv[V_ERR]; break; }
            for (int i = 0; i < cnt; i++) {
                const uint32_t llen = S.seqLL[i], mlen = S.seqML[i], off = S.seqOF[i];
                if ((uint64_t)lp + llen > (uint64_t)regen || d + llen + mlen > want || (uint64_t)off > d + llen + P.dict_len) { err = 18; break; }
                if (litRle >= 0) { for (uint32_t k = (uint32_t)lane; k < llen; k += 64) out[d + k] = (uint8_t)litRle; }
                else { for (uint32_t k = (uint32_t)lane; k < llen; k += 64) out[d + k] = L[lp + k]; }
                lp += (int)llen;
                d += llen;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if ((uint64_t)off > d) {  // the match starts in the dictionary (history in front of the frame, dict.go / history.go)
                    const uint32_t inDict = (uint32_t)((uint64_t)off - d);  // bytes of the match source that lie in the dictionary
                    for (uint32_t k = (uint32_t)lane; k < mlen; k += 64) {
                        const uint32_t j = off >= mlen ? k : k % off;  // an overlapping match repeats its first `off` bytes
                        out[d + k] = j < inDict ? P.dict[P.dict_len - inDict + j] : out[j - inDict];
                    }
                } else if (off >= mlen) { for (uint32_t k = (uint32_t)lane; k < mlen; k += 64) out[d + k] = out[d - off + k]; }
                else { for (uint32_t k = (uint32_t)lane; k < mlen; k += 64) out[d + k] = out[d - off + (k % off)]; }
                d += mlen;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
        if (err) break;
        // trailing literals
        const int tail = regen - lp;
        if (d + (uint64_t)tail > want) { err = 18; break; }
        if (litRle >= 0) { for (int k = lane; k < tail; k += 64) out[d + k] = (uint8_t)litRle; }
        else { for (int k = lane; k < tail; k += 64) out[d + k] = L[lp + k]; }
        d += (uint64_t)tail;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (!err && d != want) err = 2;
    uint32_t stored = 0xFFFFFFFFu;
    if (!err && checksum) {
        if (p + 4 > n) err = 19;
        else { stored = ld32(in + p); p += 4; }
    }
    if (!err && p != n) err = 19;  // one frame per unit
    if (lane == 0) { P.status[u] = (uint32_t)err; P.crc_stored[u] = (!err && checksum) ? stored : 0xFFFFFFFFu; P.has_crc[u] = (!err && checksum) ? 1u : 0u; }
}

void kc_launch_zstd_decode(const KcZstdDecParams& P, hipStream_t st) {
    if (P.n_units == 0) return;
    hipLaunchKernelGGL(kc_zstd_decode_kernel, dim3(P.n_units), dim3(64), 0, st, P);
}
