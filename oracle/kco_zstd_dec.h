// TEST INFRASTRUCTURE ONLY (see kco_common.h).  CPU restatement of the reference's zstd DECODE path, used as an
// in-repo verifier for frames the encoders produce so that parity tests do not depend on a system libzstd:
//   zstd/framedec.go:65-330      frame header, block loop, checksum
//   zstd/blockdec.go:227-690     block header, decodeLiterals (:275), prepareSequences (:505), sequence modes
//   zstd/seqdec.go + seqdec_generic.go:16,161   sequence decoding + execution, repeat-offset rules
//   zstd/fse_decoder.go:52-260   readNCount / buildDtable / transform (shared with kco_dict.h)
//   huff0/decompress.go:29-1095  ReadTable + 1X / 4X stream decoding
//   zstd/dict.go                 dictionary content / tables / offsets as initial state
// Written against the format description; every stage names the reference code it stands for.
#pragma once
#include <string>
#include <stdio.h>
#include "kco_common.h"
#include "kco_xxhash.h"
#include "kco_zstd_fse.h"
#include "kco_dict.h"

namespace kco {
namespace zdec {

// zstd/bitreader.go: backward bit reader over [p, p+n) whose last byte carries the end mark.
struct RBits {
    const uint8_t* p = nullptr;
    int64_t pos = 0;  // bits still unread (bit index of the next bit to deliver, exclusive)
    bool init(const uint8_t* d, size_t n) {
        if (n == 0 || d[n - 1] == 0) return false;
        p = d;
        pos = (int64_t)n * 8 - (8 - dictload::highBits(d[n - 1]));
        return true;
    }
    uint64_t peek(int nb) const {  // the next nb bits (most significant first), zero below bit 0
        uint64_t v = 0;
        for (int k = 0; k < nb; k++) {
            const int64_t b = pos - 1 - k;
            v <<= 1;
            if (b >= 0) v |= (uint64_t)((p[b >> 3] >> (b & 7)) & 1);
        }
        return v;
    }
    uint64_t read(int nb) { const uint64_t v = peek(nb); pos -= nb; return v; }
    bool overread() const { return pos < 0; }
};

struct SeqTable {  // one of LL / OF / ML for the current block (fseDecoder after transform)
    std::vector<dictload::DecSymbol> dt;
    int log = 0;
    bool valid = false;
    void setRLE(uint8_t sym) { dt.assign(1, dictload::DecSymbol{0, sym, 0}); log = 0; valid = true; }
    bool fromNorm(const int16_t* norm, int symbolLen, int tableLog) {
        dictload::NCount nc;
        memset(nc.norm, 0, sizeof(nc.norm));
        memcpy(nc.norm, norm, sizeof(int16_t) * (size_t)symbolLen);
        nc.symbolLen = (uint16_t)symbolLen;
        nc.actualTableLog = (uint8_t)tableLog;
        bool zb;
        if (!dictload::buildDtable(nc, &dt, &zb)) return false;
        log = tableLog;
        valid = true;
        return true;
    }
};

struct DictState {  // dict.go: what a decoder takes from a dictionary
    Bytes content;
    uint32_t rep[3] = {1, 4, 8};
    SeqTable ll, of, ml;
    std::vector<uint16_t> huf;  // literal decode table (empty: none)
    int hufLog = 0;
    uint32_t id = 0;
};

// huff0 decoding table (decompress.go:129-165): 1 << tableLog entries of (symbol << 8 | nBits), filled in rank order.
static inline bool buildHufTable(const uint8_t* weights, int symbolLen, int tableLog, std::vector<uint16_t>* dt) {
    uint32_t rankStart[16] = {0};
    uint32_t rankCount[16] = {0};
    for (int i = 0; i < symbolLen; i++) rankCount[weights[i] & 15]++;
    uint32_t next = 0;
    for (int r = 1; r <= tableLog; r++) { rankStart[r] = next; next += rankCount[r] << (r - 1); }
    if (next != (1u << tableLog)) return false;
    dt->assign((size_t)1 << tableLog, 0);
    for (int s = 0; s < symbolLen; s++) {
        const int w = weights[s];
        if (w == 0) continue;
        const uint32_t len = (1u << w) >> 1;
        const uint16_t e = (uint16_t)((s << 8) | (tableLog + 1 - w));
        for (uint32_t k = 0; k < len; k++) (*dt)[rankStart[w] + k] = e;
        rankStart[w] += len;
    }
    return true;
}

// huff0.ReadTable -> weights + tableLog (decompress.go:29-127), then the decoding table.  Returns bytes consumed, 0 on error.
static inline int readHufTable(const uint8_t* in, int n, std::vector<uint16_t>* dt, int* tableLog) {
    huff0::Scratch tmp;
    int used = 0;
    if (!dictload::ReadTable(in, n, &tmp, &used)) return 0;
    *tableLog = tmp.actualTableLog;
    if (!buildHufTable(tmp.huffWeight, tmp.symbolLen, tmp.actualTableLog, dt)) return 0;
    return used;
}

static inline bool hufDecodeStream(const uint8_t* in, size_t n, const std::vector<uint16_t>& dt, int log, uint8_t* out, size_t count) {
    RBits br;
    if (!br.init(in, n)) return false;
    for (size_t i = 0; i < count; i++) {
        const uint16_t e = dt[(size_t)br.peek(log)];
        out[i] = (uint8_t)(e >> 8);
        br.pos -= (e & 0xFF);
    }
    return br.pos == 0;  // a stream must be consumed exactly (decompress.go: "corruption detected: stream should be fully read")
}

struct FrameDec {
    const DictState* dict = nullptr;
    // per-frame state
    uint32_t rep[3];
    SeqTable ll, of, ml;
    std::vector<uint16_t> huf;
    int hufLog = 0;
    Bytes out;          // regenerated content of the frame
    size_t histBase = 0;  // dictionary bytes logically in front of `out`
    size_t repSeen[4] = {0, 0, 0, 0};  // inspection: sequences carrying repeat-offset code 1 / 2 / 3 in the whole frame
    std::string* trace = nullptr;  // inspection (kco_zstd_inspect): one line per block with its modes and first sequences

    static uint32_t llBase(int c) { uint32_t b = 0; for (int i = 0; i < c; i++) b += 1u << zfse::llBitsTable[i]; return b; }
    static uint32_t mlBase(int c) { uint32_t b = 3; for (int i = 0; i < c; i++) b += 1u << zfse::mlBitsTable[i]; return b; }

    // blockdec.go:275 decodeLiterals.  Returns bytes consumed (0 = error).
    size_t decodeLiterals(const uint8_t* b, size_t n, Bytes* lits) {
        if (n < 1) return 0;
        const int type = b[0] & 3, sf = (b[0] >> 2) & 3;
        size_t hdr, regen, comp = 0;
        bool four = false;
        if (type < 2) {  // raw / RLE (blockdec.go:288-340)
            if ((sf & 1) == 0) { hdr = 1; regen = b[0] >> 3; }
            else if (sf == 1) { if (n < 2) return 0; hdr = 2; regen = (b[0] >> 4) | ((size_t)b[1] << 4); }
            else { if (n < 3) return 0; hdr = 3; regen = (b[0] >> 4) | ((size_t)b[1] << 4) | ((size_t)b[2] << 12); }
            if (type == 0) {
                if (hdr + regen > n) return 0;
                lits->assign(b + hdr, b + hdr + regen);
                return hdr + regen;
            }
            if (hdr + 1 > n) return 0;
            lits->assign(regen, b[hdr]);
            return hdr + 1;
        }
        // compressed / treeless (blockdec.go:341-460)
        if (sf < 2) {
            if (n < 3) return 0;
            const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
            hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; four = sf == 1;
        } else if (sf == 2) {
            if (n < 4) return 0;
            const uint32_t v = load32(b, 0);
            hdr = 4; regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; four = true;
        } else {
            if (n < 5) return 0;
            const uint64_t v = (uint64_t)load32(b, 0) | ((uint64_t)b[4] << 32);
            hdr = 5; regen = (size_t)((v >> 4) & 0x3FFFF); comp = (size_t)((v >> 22) & 0x3FFFF); four = true;
        }
        if (hdr + comp > n) return 0;
        const uint8_t* p = b + hdr;
        size_t left = comp;
        if (type == 2) {
            const int used = readHufTable(p, (int)left, &huf, &hufLog);
            if (used == 0) return 0;
            p += used; left -= (size_t)used;
        } else if (huf.empty()) {
            return 0;  // treeless without a previous table (blockdec.go:395 "literal table was not found")
        }
        lits->assign(regen, 0);
        if (!four) {
            if (!hufDecodeStream(p, left, huf, hufLog, lits->data(), regen)) return 0;
        } else {
            if (left < 6) return 0;
            const size_t s1 = p[0] | ((size_t)p[1] << 8), s2 = p[2] | ((size_t)p[3] << 8), s3 = p[4] | ((size_t)p[5] << 8);
            if (6 + s1 + s2 + s3 > left) return 0;
            const size_t s4 = left - 6 - s1 - s2 - s3;
            const size_t seg = (regen + 3) / 4;
            if (seg * 3 > regen) return 0;
            const uint8_t* q = p + 6;
            if (!hufDecodeStream(q, s1, huf, hufLog, lits->data(), seg)) return 0;
            if (!hufDecodeStream(q + s1, s2, huf, hufLog, lits->data() + seg, seg)) return 0;
            if (!hufDecodeStream(q + s1 + s2, s3, huf, hufLog, lits->data() + 2 * seg, seg)) return 0;
            if (!hufDecodeStream(q + s1 + s2 + s3, s4, huf, hufLog, lits->data() + 3 * seg, regen - 3 * seg)) return 0;
        }
        return hdr + comp;
    }

    // blockdec.go:560-640: one sequence table according to its compression mode.  Advances *pp.
    bool readSeqTable(int mode, int kind, const uint8_t** pp, const uint8_t* end, SeqTable* t) {
        static const int maxSym[3] = {35, 30, 52}, maxLog[3] = {9, 8, 9};  // maxOffsetLengthSymbol = 30 (zstd/fse_predefined.go:45)
        if (mode == 0) {  // predefined
            zfse::Predef& pd = zfse::predef();
            return t->fromNorm(pd.enc[kind].norm, pd.enc[kind].symbolLen, pd.enc[kind].actualTableLog);
        }
        if (mode == 1) {  // RLE
            if (*pp >= end) return false;
            const uint8_t s = **pp;
            if (s > maxSym[kind]) return false;
            (*pp)++;
            t->setRLE(s);
            return true;
        }
        if (mode == 2) {  // FSE compressed
            dictload::ByteReader br{*pp, (int)(end - *pp)};
            dictload::NCount nc;
            if (!dictload::readNCount(&br, &nc, 9, maxSym[kind], true)) return false;
            if (nc.actualTableLog > maxLog[kind]) return false;
            for (int i = maxSym[kind] + 1; i < nc.symbolLen; i++)
                if (nc.norm[i] != 0) return false;  // transform: "symbol >= max" (fse_decoder.go:279)
            if (!t->fromNorm(nc.norm, nc.symbolLen, nc.actualTableLog)) return false;
            *pp += br.off;
            return true;
        }
        return t->valid;  // repeat
    }

    // blockdec.go:505 prepareSequences + seqdec_generic.go decode/execute for one compressed block
    bool decodeCompressed(const uint8_t* b, size_t n) {
        Bytes lits;
        const size_t lsz = decodeLiterals(b, n, &lits);
        if (lsz == 0) return false;
        const uint8_t* p = b + lsz;
        const uint8_t* end = b + n;
        if (p >= end) return false;
        size_t nSeq = *p++;
        if (nSeq >= 128) {
            if (nSeq < 255) { if (p >= end) return false; nSeq = ((nSeq - 128) << 8) + *p++; }
            else { if (p + 2 > end) return false; nSeq = (size_t)p[0] + ((size_t)p[1] << 8) + 0x7F00; p += 2; }
        }
        if (trace) {
            char tb[128];
            snprintf(tb, sizeof(tb), " litType=%d litBytes=%zu nSeq=%zu", b[0] & 3, lits.size(), nSeq);
            *trace += tb;
        }
        if (nSeq == 0) {
            if (p != end) return false;
            out.insert(out.end(), lits.begin(), lits.end());
            return true;
        }
        if (p >= end) return false;
        const uint8_t modes = *p++;
        if (trace) {
            char tb[96];
            snprintf(tb, sizeof(tb), " modes(ll,of,ml)=%d,%d,%d first:", (modes >> 6) & 3, (modes >> 4) & 3, (modes >> 2) & 3);
            *trace += tb;
        }
        if (modes & 3) return false;  // reserved bits
        if (!readSeqTable((modes >> 6) & 3, 0, &p, end, &ll)) return false;
        if (!readSeqTable((modes >> 4) & 3, 1, &p, end, &of)) return false;
        if (!readSeqTable((modes >> 2) & 3, 2, &p, end, &ml)) return false;
        RBits br;
        if (!br.init(p, (size_t)(end - p))) return false;
        uint32_t llS = (uint32_t)br.read(ll.log), ofS = (uint32_t)br.read(of.log), mlS = (uint32_t)br.read(ml.log);
        size_t lp = 0;
        for (size_t i = 0; i < nSeq; i++) {
            const int lc = ll.dt[llS].symbol, oc = of.dt[ofS].symbol, mc = ml.dt[mlS].symbol;
            if (lc > 35 || mc > 52 || oc > 30) return false;
            const uint64_t ofVal = ((uint64_t)1 << oc) + br.read(oc);
            const uint32_t mlen = mlBase(mc) + (uint32_t)br.read(zfse::mlBitsTable[mc]);
            const uint32_t llen = llBase(lc) + (uint32_t)br.read(zfse::llBitsTable[lc]);
            if (trace && ofVal <= 3) repSeen[ofVal]++;
            if (trace && i < 4) {
                char tb[96];
                snprintf(tb, sizeof(tb), " [ll=%u ml=%u %s%llu]", llen, mlen, ofVal > 3 ? "off=" : "REP", (unsigned long long)(ofVal > 3 ? ofVal - 3 : ofVal));
                *trace += tb;
            }
            uint32_t off;
            if (ofVal > 3) {  // seqdec.go: new offset
                off = (uint32_t)(ofVal - 3);
                rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off;
            } else {
                uint32_t idx = (uint32_t)ofVal + (llen == 0 ? 1u : 0u);
                if (idx == 1) off = rep[0];
                else {
                    off = idx == 4 ? rep[0] - 1 : rep[idx - 1];
                    if (off == 0) return false;  // "corrupt: offset 0"
                    if (idx != 2) rep[2] = rep[1];
                    rep[1] = rep[0];
                    rep[0] = off;
                }
            }
            if (i + 1 < nSeq) {  // state updates: LL, ML, OF (seqdec_generic.go:97-117)
                llS = (uint32_t)ll.dt[llS].newState + (uint32_t)br.read(ll.dt[llS].nbBits);
                mlS = (uint32_t)ml.dt[mlS].newState + (uint32_t)br.read(ml.dt[mlS].nbBits);
                ofS = (uint32_t)of.dt[ofS].newState + (uint32_t)br.read(of.dt[ofS].nbBits);
            }
            if (br.overread()) return false;
            // execute (seqdec.go:119 execute): literals then the match, possibly reaching into the dictionary
            if (lp + llen > lits.size()) return false;
            out.insert(out.end(), lits.begin() + (long)lp, lits.begin() + (long)(lp + llen));
            lp += llen;
            if ((uint64_t)off > (uint64_t)out.size() + histBase) return false;
            for (uint32_t k = 0; k < mlen; k++) {
                const int64_t src = (int64_t)out.size() - (int64_t)off;
                out.push_back(src >= 0 ? out[(size_t)src] : dict->content[(size_t)((int64_t)dict->content.size() + src)]);
            }
        }
        if (br.pos != 0) return false;  // "extra bits on stream"
        out.insert(out.end(), lits.begin() + (long)lp, lits.end());
        return true;
    }

    // framedec.go:65 reset + :290 runDecoder.  Returns bytes consumed from `in` (0 = error); *content receives the frame.
    size_t decodeFrame(const uint8_t* in, size_t n, const DictState* d, Bytes* content) {
        if (n < 6 || load32(in, 0) != 0xFD2FB528u) return 0;
        const uint8_t fhd = in[4];
        size_t p = 5;
        const bool single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1;
        if (fhd & 8) return 0;  // reserved bit
        if (!single) { if (p >= n) return 0; p++; }  // window descriptor: this verifier keeps the whole frame
        const int dsz = (fhd & 3) == 3 ? 4 : (fhd & 3);
        uint32_t dictID = 0;
        if (p + (size_t)dsz > n) return 0;
        for (int k = 0; k < dsz; k++) dictID |= (uint32_t)in[p + (size_t)k] << (8 * k);
        p += (size_t)dsz;
        int fcsSize = (fhd >> 6) == 0 ? (single ? 1 : 0) : (1 << (fhd >> 6));
        uint64_t fcs = 0;
        if (p + (size_t)fcsSize > n) return 0;
        for (int k = 0; k < fcsSize; k++) fcs |= (uint64_t)in[p + (size_t)k] << (8 * k);
        if (fcsSize == 2) fcs += 256;
        p += (size_t)fcsSize;
        dict = (d != nullptr && (dictID == 0 || dictID == d->id)) ? d : nullptr;
        if (dictID != 0 && dict == nullptr) return 0;  // ErrUnknownDictionary
        static const DictState none;
        if (dict == nullptr) dict = &none;
        histBase = dict->content.size();
        rep[0] = dict->rep[0]; rep[1] = dict->rep[1]; rep[2] = dict->rep[2];
        ll = dict->ll; of = dict->of; ml = dict->ml;
        huf = dict->huf; hufLog = dict->hufLog;
        out.clear();
        for (;;) {
            if (p + 3 > n) return 0;
            const uint32_t bh = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
            p += 3;
            const bool last = bh & 1;
            const int type = (bh >> 1) & 3;
            const size_t size = bh >> 3;
            if (trace) {
                char tb[96];
                snprintf(tb, sizeof(tb), "%sblock type=%d size=%zu last=%d", trace->empty() ? "" : "\n", type, size, (int)last);
                *trace += tb;
            }
            if (type == 0) {
                if (p + size > n) return 0;
                out.insert(out.end(), in + p, in + p + size);
                p += size;
            } else if (type == 1) {
                if (p + 1 > n) return 0;
                out.insert(out.end(), size, in[p]);
                p += 1;
            } else if (type == 2) {
                if (p + size > n || size > (128u << 10)) return 0;
                if (!decodeCompressed(in + p, size)) return 0;
                p += size;
            } else {
                return 0;
            }
            if (last) break;
        }
        if (fcsSize > 0 && fcs != out.size()) return 0;
        if (checksum) {
            if (p + 4 > n) return 0;
            XXH64 h;
            h.Reset();
            h.Write(out.data(), out.size());
            if ((uint32_t)h.Sum64() != load32(in, (int64_t)p)) return 0;
            p += 4;
        }
        *content = out;
        return p;
    }
};

// dict.go loadDict, decoder side: content, offsets, literal table, the three sequence tables ("repeat" state of block 0).
static inline bool loadDictState(const uint8_t* blob, size_t len, DictState* ds) {
    DictO d;
    huff0::Scratch lit;
    if (!dictload::loadDict(blob, len, &d, &lit)) return false;
    ds->id = d.id;
    ds->content = d.content;
    for (int k = 0; k < 3; k++) ds->rep[k] = (uint32_t)d.offsets[k];
    ds->hufLog = lit.prevTableLog;
    if (!buildHufTable(lit.huffWeight, lit.symbolLen, lit.actualTableLog, &ds->huf)) return false;
    // re-read the three tables (loadDict only validated them)
    int used = 0;
    huff0::Scratch tmp;
    if (!dictload::ReadTable(blob + 8, (int)(len - 8), &tmp, &used)) return false;
    dictload::ByteReader br{blob + 8 + used, (int)(len - 8) - used};
    const int maxSym[3] = {31, 52, 35};
    SeqTable* tabs[3] = {&ds->of, &ds->ml, &ds->ll};
    for (int t = 0; t < 3; t++) {
        dictload::NCount nc;
        if (!dictload::readNCount(&br, &nc, 9, maxSym[t], true)) return false;
        if (!tabs[t]->fromNorm(nc.norm, nc.symbolLen, nc.actualTableLog)) return false;
    }
    return true;
}

static inline void rawDictState(uint32_t id, const uint8_t* content, size_t len, DictState* ds) {
    ds->id = id;
    ds->content.assign(content, content + len);
}

}  // namespace zdec
}  // namespace kco
