// oracle/kco_api.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// C entry points (ctypes-friendly) over the restated reference encoders.
// zstd: restates (*Encoder).encodeAll, zstd/encoder.go:731-839, MaxEncodedSize :843-873,
// frameHeader.appendTo zstd/frameenc.go:25-92.
#include "kco_common.h"
#include "kco_xxhash.h"
#include "kco_huff0.h"
#include "kco_zstd_fse.h"
#include "kco_zstd_block.h"
#include "kco_zstd_fast.h"
#include "kco_zstd_dfast.h"
#include "kco_zstd_better.h"
#include "kco_s2.h"
#include "kco_s2_asm.h"
#include "kco_dict.h"
#include "kco_zstd_best.h"
#include "kco_zstd_dec.h"
#include <thread>
#include <atomic>
#include <memory>

using namespace kco;

extern "C" {

// Resolved encoder options (after applying the reference's EOption functions in order;
// the option-resolution logic itself lives in the product's host layer and is tested
// against zstd/encoder_options.go).
typedef struct {
    int32_t level;            // 1 SpeedFastest, 2 SpeedDefault, 3 SpeedBetterCompression
    int32_t window_size;      // o.windowSize
    int32_t block_size;       // o.blockSize
    int32_t crc;              // o.crc
    int32_t single;           // -1: o.single == nil; else 0/1
    int32_t full_zero;        // o.fullZero
    int32_t no_entropy;       // o.noEntropy
    int32_t all_lit_entropy;  // o.allLitEntropy
    int32_t low_mem;          // o.lowMem
    uint32_t dict_id;         // 0 = no dict (raw-content dictionary, WithEncoderDictRaw)
    const uint8_t* dict;      // dict content
    uint64_t dict_len;
    int32_t dict_full;        // 1: dict/dict_len is a full-format dictionary blob (WithEncoderDict -> loadDict)
    int32_t concurrent;       // o.concurrent (WithEncoderConcurrency); 0 = the default (GOMAXPROCS > 1).  Only "1 or not" changes bytes,
                              // and only of dictionary streams (encodeStream)
} kco_zstd_opts;

}  // extern "C"

namespace {

// zstd/frameenc.go:25 appendTo
void frameHeaderAppend(Bytes* dst, uint64_t ContentSize, uint32_t WindowSize, bool SingleSegment, bool Checksum, uint32_t DictID) {
    dst->push_back(0x28); dst->push_back(0xb5); dst->push_back(0x2f); dst->push_back(0xfd);
    uint8_t fhd = 0;
    if (Checksum) fhd |= 1 << 2;
    if (SingleSegment) fhd |= 1 << 5;
    uint8_t dictIDContent[4];
    int dictIDLen = 0;
    if (DictID > 0) {
        if (DictID < 256) { fhd |= 1; dictIDContent[0] = (uint8_t)DictID; dictIDLen = 1; }
        else if (DictID < (1u << 16)) { fhd |= 2; dictIDContent[0] = (uint8_t)DictID; dictIDContent[1] = (uint8_t)(DictID >> 8); dictIDLen = 2; }
        else { fhd |= 3; for (int i = 0; i < 4; i++) dictIDContent[i] = (uint8_t)(DictID >> (8 * i)); dictIDLen = 4; }
    }
    uint8_t fcs = 0;
    if (ContentSize >= 256) fcs++;
    if (ContentSize >= 65536 + 256) fcs++;
    if (ContentSize >= 0xffffffffULL) fcs++;
    fhd |= (uint8_t)(fcs << 6);
    dst->push_back(fhd);
    if (!SingleSegment) {
        const int winLogMin = 10;
        int windowLog = (bitsLen32(WindowSize - 1) - winLogMin) << 3;
        dst->push_back((uint8_t)windowLog);
    }
    for (int i = 0; i < dictIDLen; i++) dst->push_back(dictIDContent[i]);
    switch (fcs) {
    case 0:
        if (SingleSegment) dst->push_back((uint8_t)ContentSize);
        break;
    case 1:
        ContentSize -= 256;
        dst->push_back((uint8_t)ContentSize); dst->push_back((uint8_t)(ContentSize >> 8));
        break;
    case 2:
        for (int i = 0; i < 4; i++) dst->push_back((uint8_t)(ContentSize >> (8 * i)));
        break;
    case 3:
        for (int i = 0; i < 8; i++) dst->push_back((uint8_t)(ContentSize >> (8 * i)));
        break;
    }
}

struct OracleEncoder {
    kco_zstd_opts o;
    DictO dict;
    bool hasDict = false;
    bool bad = false;  // loadDict returned an error (the EOption would have failed)
    huff0::Scratch dictLit;
    std::unique_ptr<FastBase> enc;

    explicit OracleEncoder(const kco_zstd_opts& opts) : o(opts) {
        if (o.dict_full) {
            hasDict = true;
            if (!dictload::loadDict(o.dict, (size_t)o.dict_len, &dict, &dictLit)) bad = true;
        } else if (o.dict_id != 0 || (o.dict != nullptr && o.dict_len > 0)) {
            hasDict = true;
            dict.id = o.dict_id;
            dict.content.assign(o.dict, o.dict + o.dict_len);
            dict.offsets[0] = 1; dict.offsets[1] = 4; dict.offsets[2] = 8;  // WithEncoderDictRaw, encoder_options.go:398
            dict.litEnc = nullptr;
        }
        // encoderOptions.encoder(), encoder_options.go:51
        switch (o.level) {
        case 1: if (hasDict) enc.reset(new FastEncoderDict()); else enc.reset(new FastEncoder()); break;
        case 2: if (hasDict) enc.reset(new DoubleFastEncoderDict()); else enc.reset(new DoubleFastEncoder()); break;
        case 3: if (hasDict) enc.reset(new BetterFastEncoderDict()); else enc.reset(new BetterFastEncoder()); break;
        case 4: enc.reset(new BestFastEncoder()); break;
        default: enc.reset(new FastEncoder());
        }
        enc->setup(o.window_size, o.low_mem != 0);
    }

    // zstd/encoder.go:731 encodeAll — appends to dst.
    int encodeAll(const uint8_t* src, size_t n, Bytes* dst) {
        const DictO* d = hasDict ? &dict : nullptr;
        if (n == 0) {
            if (o.full_zero) {
                frameHeaderAppend(dst, 0, MinWindowSize, true, false, 0);
                BlockHeader blk;
                blk.setSize(0);
                blk.setType(blockTypeRaw);
                blk.setLast(true);
                blk.appendTo(dst);
            }
            return 0;
        }
        bool single = (int64_t)n <= (int64_t)o.window_size && (int64_t)n > MinWindowSize;
        if (o.single >= 0) single = o.single != 0;
        frameHeaderAppend(dst, (uint64_t)n, (uint32_t)enc->WindowSize((int64_t)n), single, o.crc != 0, d ? d->id : 0);
        if ((int64_t)n <= (int64_t)o.block_size) {
            enc->Reset(d, true);
            if (o.crc) enc->crc.Write(src, n);
            BlockEnc* blk = &enc->blk;
            blk->last = true;
            if (d == nullptr) enc->EncodeNoHist(blk, src, n);
            else enc->Encode(blk, src, n);
            Bytes oldout;
            oldout.swap(blk->output);
            blk->output.swap(*dst);
            int err = blk->encode(src, n, o.no_entropy != 0, !o.all_lit_entropy);
            blk->output.swap(*dst);
            blk->output.swap(oldout);
            if (err != 0) return -1;
        } else {
            enc->Reset(d, false);
            BlockEnc* blk = &enc->blk;
            while (n > 0) {
                size_t todo = n;
                if (todo > (size_t)o.block_size) todo = (size_t)o.block_size;
                const uint8_t* tp = src;
                src += todo;
                n -= todo;
                if (o.crc) enc->crc.Write(tp, todo);
                blk->pushOffsets();
                enc->Encode(blk, tp, todo);
                if (n == 0) blk->last = true;
                int err = blk->encode(tp, todo, o.no_entropy != 0, !o.all_lit_entropy);
                if (err != 0) return -1;
                dst->insert(dst->end(), blk->output.begin(), blk->output.end());
                blk->reset(nullptr);
            }
        }
        if (o.crc) enc->AppendCRC(dst);
        return 0;
    }

    // Streaming use of the encoder (zstd/encoder.go: Write :154-253 -> nextBlock :257-428 -> Close :567-649).  `cuts` lists the
    // input positions at which Flush was called (ascending, may be empty); full blocks are cut every blockSize bytes after the
    // last cut, as writeBlocks does.  nextBlock has a synchronous form (o.concurrent == 1, :364-391) and an asynchronous one
    // (:393-428, two blockEnc objects that swap their entropy coders).  Without a dictionary they differ only in which stale repeat
    // offsets a block starts with, and no matcher reads them before it has found three sequences of its own
    // (`canRepeat := len(blk.sequences) > 2`), so both forms give the same bytes.  With a dictionary they differ in one more
    // thing: the synchronous form calls blk.reset(nil) before the FIRST Encode too (:371), which clears blk.dictLitEnc
    // (blockenc.go:97), so the dictionary's literal table never reaches a block; the asynchronous form takes enc.Block() as
    // Reset left it (enc_base.go:196), and the first block's literals start from the dictionary table like EncodeAll's
    // (blockenc.go:518-522; swapEncoders :103-106 exchanges litEnc, not dictLitEnc).  o.concurrent selects the form.
    int encodeStream(const uint8_t* src, size_t n, const uint64_t* cuts, size_t n_cuts, Bytes* dst) {
        const DictO* d = hasDict ? &dict : nullptr;
        const bool syncForm = o.concurrent == 1;
        // block boundaries
        std::vector<size_t> ends;
        {
            size_t pos = 0, ci = 0;
            while (pos < n) {
                size_t e = pos + (size_t)o.block_size;
                while (ci < n_cuts && cuts[ci] <= pos) ci++;
                if (ci < n_cuts && cuts[ci] < e) e = (size_t)cuts[ci];
                if (e > n) e = n;
                ends.push_back(e);
                pos = e;
            }
        }
        // Did Close find bytes in `filling`?  Not if a Flush drained it at the very end or the tail filled a whole block.
        bool tailBuffered = false;
        if (!ends.empty()) {
            const size_t lastStart = ends.size() > 1 ? ends[ends.size() - 2] : 0;
            const bool flushedAtEnd = n_cuts > 0 && cuts[n_cuts - 1] >= n;
            tailBuffered = !flushedAtEnd && (n - lastStart) < (size_t)o.block_size;
        }
        if (n == 0 || (ends.size() == 1 && tailBuffered)) {
            if (n == 0) {
                if (!o.full_zero) return 0;  // :266-271
            } else {
                return encodeAll(src, n, dst);  // :272-288 single block: a complete EncodeAll frame
            }
        }
        enc->Reset(d, false);  // Encoder.Reset, encoder.go:138
        frameHeaderAppend(dst, 0, (uint32_t)enc->WindowSize(0), false, o.crc != 0, d ? d->id : 0);  // :290-297
        BlockEnc* blk = &enc->blk;
        size_t pos = 0;
        for (size_t bi = 0; bi < ends.size(); bi++) {
            const size_t todo = ends[bi] - pos;
            if (o.crc) enc->crc.Write(src + pos, todo);
            if (bi > 0 || syncForm) blk->reset(nullptr);  // asynchronous form: the first block is enc.Block() as Reset left it
            enc->Encode(blk, src + pos, todo);
            blk->last = (bi + 1 == ends.size()) && tailBuffered;
            if (blk->encode(src + pos, todo, o.no_entropy != 0, !o.all_lit_entropy) != 0) return -1;
            dst->insert(dst->end(), blk->output.begin(), blk->output.end());
            pos = ends[bi];
        }
        if (!tailBuffered) {  // :315-329 final block without data
            BlockHeader bh;
            bh.setSize(0);
            bh.setType(blockTypeRaw);
            bh.setLast(true);
            bh.appendTo(dst);
        }
        if (o.crc) enc->AppendCRC(dst);
        return 0;
    }

    // WithConcurrentBlocks(true) (and concurrency > 1, no dictionary): NewWriter(w); Write(src[...]) with Flush at each cuts[i];
    // Close().  Restates writeJobs (encoder.go:214-247), flushJobs (:585-597), closeJobs (:652-700), dispatchJob and compressJob
    // (enc_jobs.go:251-352, 88-124) for that call sequence: jobs of jobSize = max(4 * window, 512 KiB) input bytes
    // (encoder_options.go:356-359), each encoded on a freshly reset encoder whose history is the last overlapSize bytes of the
    // previous job's input (:362-371; ResetPrefix), the job outputs concatenated behind one frame header.
    int encodeJobs(const uint8_t* src, size_t n, const uint64_t* cuts, size_t n_cuts, Bytes* dst) {
        if (hasDict) return -1;  // the reference switches the option off with a dictionary (encoder.go:81, :174)
        const size_t jobSize = std::max<size_t>((size_t)o.window_size * 4, (size_t)512 << 10);
        const size_t overlapSize = o.level == 4 ? (size_t)o.window_size / 2 : (o.level == 3 ? (size_t)o.window_size / 4 : (size_t)o.window_size / 8);
        // non-final jobs: dispatched when `filling` reaches jobSize, or by a Flush that finds bytes in it
        std::vector<std::pair<size_t, size_t>> jobs;
        size_t pos = 0, ci = 0;
        for (;;) {
            while (ci < n_cuts && cuts[ci] <= pos) ci++;
            size_t e = pos + jobSize;
            bool dispatched = e <= n;  // filled up during Write
            if (ci < n_cuts && cuts[ci] < e && cuts[ci] <= n) { e = (size_t)cuts[ci]; dispatched = true; }
            if (!dispatched) break;
            jobs.push_back({pos, e});
            pos = e;
        }
        const size_t tail = n - pos;  // what Close finds in `filling`
        bool headerWritten = !jobs.empty();
        if (!headerWritten) {  // dispatchJob(true), :263-289
            if (tail > 0 && tail <= (size_t)o.block_size) return encodeAll(src, n, dst);
            if (tail == 0 && !o.full_zero) return 0;
        }
        enc->Reset(nullptr, false);  // (the encoder of the Encoder state: only its WindowSize and CRC are used below)
        frameHeaderAppend(dst, 0, (uint32_t)enc->WindowSize(0), false, o.crc != 0, 0);
        jobs.push_back({pos, n});  // the final job, possibly empty
        XXH64 crcAll;
        if (o.crc) crcAll.Write(src, n);
        size_t prevLo = 0, prevHi = 0;
        for (size_t j = 0; j < jobs.size(); j++) {
            const bool last = j + 1 == jobs.size();
            const size_t lo = jobs[j].first, hi = jobs[j].second;
            // compressJob (enc_jobs.go:88)
            if (j > 0 && prevHi > prevLo) {
                const size_t ov = std::min(overlapSize, prevHi - prevLo);
                enc->ResetPrefix(src + prevHi - ov, ov);
            } else {
                enc->Reset(nullptr, false);
            }
            BlockEnc* blk = &enc->blk;
            if (hi == lo && last) {
                blk->reset(nullptr);
                blk->last = true;
                blk->encodeRawTo(0, src, 0);  // blockenc.go:310 encodeRaw(nil): an empty raw block with the last flag
                dst->insert(dst->end(), blk->output.begin(), blk->output.end());
            } else {
                size_t p = lo;
                while (p < hi) {
                    const size_t todo = std::min<size_t>(hi - p, (size_t)o.block_size);
                    blk->pushOffsets();
                    enc->Encode(blk, src + p, todo);
                    blk->last = (p + todo == hi) && last;
                    if (blk->encode(src + p, todo, o.no_entropy != 0, !o.all_lit_entropy) != 0) return -1;
                    dst->insert(dst->end(), blk->output.begin(), blk->output.end());
                    blk->reset(nullptr);
                    p += todo;
                }
            }
            prevLo = lo;
            prevHi = hi;
        }
        if (o.crc) {
            const uint64_t h = crcAll.Sum64();
            for (int k = 0; k < 4; k++) dst->push_back((uint8_t)(h >> (8 * k)));
        }
        return 0;
    }
};

// zstd/encoder.go:843 MaxEncodedSize (pad==0)
int64_t maxEncodedSize(const kco_zstd_opts* o, int64_t size) {
    int64_t frameHeader = 4 + 2;
    if (o->dict_id != 0 || o->dict_len != 0) frameHeader += 4;
    if (size < 256) frameHeader++;
    else if (size < 65536 + 256) frameHeader += 2;
    else if (size < 0x7fffffff) frameHeader += 4;
    else frameHeader += 8;
    if (o->crc) frameHeader += 4;
    int64_t blocks = (size + o->block_size) / o->block_size;
    return frameHeader + 3 * blocks + size;
}

}  // namespace

extern "C" {

// s2.Index: reset(block_size); add(comp[i], uncomp[i]) for each output write; appendTo(nil, uncomp_total, comp_total).
int64_t kco_s2_index(int32_t block_size, const int64_t* comp, const int64_t* uncomp, uint64_t n, int64_t uncomp_total, int64_t comp_total,
                     uint8_t* dst, uint64_t cap) {
    s2::Index ix;
    ix.reset(block_size);
    for (uint64_t i = 0; i < n; i++)
        if (ix.add(comp[i], uncomp[i]) != 0) return -1;
    Bytes out;
    ix.appendTo(&out, uncomp_total, comp_total);
    if (out.size() > cap) return -2;
    memcpy(dst, out.data(), out.size());
    return (int64_t)out.size();
}

// NewWriter(w).Write(src[...]); Flush at each cuts[i]; Close() on a persistent encoder state.  Returns bytes or -1 / -2.
int64_t kco_zstd_encode_stream(void* e, const uint8_t* src, uint64_t n, const uint64_t* cuts, uint64_t n_cuts, uint8_t* dst, uint64_t cap) {
    Bytes out;
    if (((OracleEncoder*)e)->encodeStream(src, (size_t)n, cuts, (size_t)n_cuts, &out) != 0) return -1;
    if (out.size() > cap) return -2;
    memcpy(dst, out.data(), out.size());
    return (int64_t)out.size();
}

// The same with WithConcurrentBlocks(true) (OracleEncoder::encodeJobs).
int64_t kco_zstd_encode_jobs(void* e, const uint8_t* src, uint64_t n, const uint64_t* cuts, uint64_t n_cuts, uint8_t* dst, uint64_t cap) {
    Bytes out;
    if (((OracleEncoder*)e)->encodeJobs(src, (size_t)n, cuts, (size_t)n_cuts, &out) != 0) return -1;
    if (out.size() > cap) return -2;
    memcpy(dst, out.data(), out.size());
    return (int64_t)out.size();
}

// loadDict result as the encoder sees it (zstd/dict.go:71-150): returns 0, or -1 when loadDict errors.
int kco_zstd_load_dict(const uint8_t* blob, uint64_t len, uint32_t* id, int32_t* offsets, uint16_t* val, uint8_t* nbits,
                       int32_t* huf_len, int32_t* huf_log, uint64_t* content_off) {
    DictO d;
    huff0::Scratch lit;
    if (!dictload::loadDict(blob, (size_t)len, &d, &lit)) return -1;
    *id = d.id;
    for (int k = 0; k < 3; k++) offsets[k] = d.offsets[k];
    for (int k = 0; k < 256; k++) { val[k] = k < lit.prevTable.len ? lit.prevTable.e[k].val : 0; nbits[k] = k < lit.prevTable.len ? lit.prevTable.e[k].nBits : 0; }
    *huf_len = lit.prevTable.len;
    *huf_log = lit.prevTableLog;
    *content_off = len - d.content.size();
    return 0;
}

// In-repo zstd decoder (kco_zstd_dec.h): decodes the concatenated frames in [enc, enc+n) into dst.
// dict/dict_len: raw dictionary content (dict_full = 0, any id accepted) or a full-format dictionary blob (dict_full = 1).
// Returns the decoded size, -1 on a format error, -2 if dst is too small.
int64_t kco_zstd_decode(const uint8_t* enc, uint64_t n, uint8_t* dst, uint64_t cap, const uint8_t* dict, uint64_t dict_len, int dict_full) {
    zdec::DictState ds;
    const zdec::DictState* dp = nullptr;
    if (dict != nullptr && dict_len > 0) {
        if (dict_full) { if (!zdec::loadDictState(dict, (size_t)dict_len, &ds)) return -1; }
        else zdec::rawDictState(0, dict, (size_t)dict_len, &ds);
        dp = &ds;
    }
    uint64_t pos = 0, outp = 0;
    zdec::FrameDec fd;
    while (pos < n) {
        Bytes content;
        if (n - pos >= 8 && (load32(enc, (int64_t)pos) & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame (framedec.go:79-110)
            const uint64_t sz = load32(enc, (int64_t)pos + 4);
            if (pos + 8 + sz > n) return -1;
            pos += 8 + sz;
            continue;
        }
        if (dict != nullptr && !dict_full) {  // raw content dictionaries match whatever id the frame carries
            const uint8_t fhd = n - pos > 5 ? enc[pos + 4] : 0;
            const int dsz = (fhd & 3) == 3 ? 4 : (fhd & 3);
            const size_t at = pos + 5 + (((fhd >> 5) & 1) ? 0 : 1);
            uint32_t id = 0;
            for (int k = 0; k < dsz && at + (size_t)k < n; k++) id |= (uint32_t)enc[at + (size_t)k] << (8 * k);
            ds.id = id;
        }
        const size_t used = fd.decodeFrame(enc + pos, (size_t)(n - pos), dp, &content);
        if (used == 0) return -1;
        if (outp + content.size() > cap) return -2;
        memcpy(dst + outp, content.data(), content.size());
        outp += content.size();
        pos += used;
    }
    return (int64_t)outp;
}

void* kco_zstd_encoder_new(const kco_zstd_opts* opts) {
    OracleEncoder* e = new OracleEncoder(*opts);
    if (e->bad) { delete e; return nullptr; }
    return e;
}
void kco_zstd_encoder_free(void* e) { delete (OracleEncoder*)e; }

// EncodeAll(src, nil) on a persistent encoder state (like one pooled Go encoder).
// Returns bytes written or -1 (error) / -2 (dst too small).
int64_t kco_zstd_encode_all(void* e, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) {
    Bytes out;
    out.reserve((size_t)n / 2 + 64);
    if (((OracleEncoder*)e)->encodeAll(src, (size_t)n, &out) != 0) return -1;
    if (out.size() > cap) return -2;
    memcpy(dst, out.data(), out.size());
    return (int64_t)out.size();
}

int64_t kco_zstd_max_encoded_size(const kco_zstd_opts* o, int64_t size) { return maxEncodedSize(o, size); }

// Encode many independent units (each == EncodeAll(unit, nil)) with `threads` host threads,
// one encoder state per thread (the reference's pool of `concurrent` encoders,
// zstd/encoder.go:90-99).  out_off has n_units+1 entries; units are written at
// dst + unit_index*stride and compacted afterwards when `compact` != 0.
int64_t kco_zstd_encode_units(const kco_zstd_opts* opts, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                              uint8_t* dst, uint64_t dst_cap, uint64_t* out_off, int threads) {
    if (threads < 1) threads = 1;
    std::vector<Bytes> outs(n_units);
    std::atomic<uint32_t> next(0);
    std::atomic<int> fail(0);
    auto worker = [&]() {
        OracleEncoder enc(*opts);
        if (enc.bad) { fail = 1; return; }
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= n_units) break;
            if (enc.encodeAll(src + unit_off[i], (size_t)(unit_off[i + 1] - unit_off[i]), &outs[i]) != 0) fail = 1;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    if (fail) return -1;
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n_units; i++) {
        out_off[i] = pos;
        pos += outs[i].size();
    }
    out_off[n_units] = pos;
    if (pos > dst_cap) return -2;
    // gather the frames with the same threads (first touch of dst in parallel, like the encode itself)
    std::atomic<uint32_t> nextc(0);
    auto copier = [&]() {
        for (;;) {
            uint32_t i = nextc.fetch_add(64);
            if (i >= n_units) break;
            const uint32_t e = i + 64 < n_units ? i + 64 : n_units;
            for (uint32_t k = i; k < e; k++) {
                memcpy(dst + out_off[k], outs[k].data(), outs[k].size());
                Bytes().swap(outs[k]);
            }
        }
    };
    th.clear();
    for (int t = 0; t < threads; t++) th.emplace_back(copier);
    for (auto& t : th) t.join();
    return (int64_t)pos;
}

// Debug/inspection: run only the match finder for one unit and return the sequence list
// and literal bytes of each block (used to diff GPU intermediates against the oracle).
// seqs out: triples (litLen, matchLen, offset) per sequence; blk_nseq[b], blk_nlit[b] per block.
int64_t kco_zstd_parse_unit(const kco_zstd_opts* opts, const uint8_t* src, uint64_t n, uint32_t* seqs, uint64_t seq_cap,
                            uint8_t* lits, uint64_t lit_cap, uint32_t* blk_nseq, uint32_t* blk_nlit, uint32_t max_blocks) {
    OracleEncoder E(*opts);
    const DictO* d = E.hasDict ? &E.dict : nullptr;
    uint64_t ns = 0, nl = 0;
    uint32_t nb = 0;
    auto dump = [&](BlockEnc* blk) -> bool {
        if (nb >= max_blocks) return false;
        if (ns + blk->sequences.size() > seq_cap || nl + blk->literals.size() > lit_cap) return false;
        for (auto& s : blk->sequences) { seqs[3 * ns] = s.litLen; seqs[3 * ns + 1] = s.matchLen; seqs[3 * ns + 2] = s.offset; ns++; }
        memcpy(lits + nl, blk->literals.data(), blk->literals.size());
        nl += blk->literals.size();
        blk_nseq[nb] = (uint32_t)blk->sequences.size();
        blk_nlit[nb] = (uint32_t)blk->literals.size();
        nb++;
        return true;
    };
    if ((int64_t)n <= (int64_t)E.o.block_size) {
        E.enc->Reset(d, true);
        BlockEnc* blk = &E.enc->blk;
        blk->last = true;
        if (d == nullptr) E.enc->EncodeNoHist(blk, src, (size_t)n);
        else E.enc->Encode(blk, src, (size_t)n);
        if (!dump(blk)) return -2;
    } else {
        E.enc->Reset(d, false);
        BlockEnc* blk = &E.enc->blk;
        while (n > 0) {
            size_t todo = (size_t)n;
            if (todo > (size_t)E.o.block_size) todo = (size_t)E.o.block_size;
            blk->pushOffsets();
            E.enc->Encode(blk, src, todo);
            if (!dump(blk)) return -2;
            if (n == todo) blk->last = true;
            if (blk->encode(src, todo, E.o.no_entropy != 0, !E.o.all_lit_entropy) != 0) return -1;
            blk->reset(nullptr);
            src += todo;
            n -= todo;
        }
    }
    return (int64_t)nb;
}

// The per-code bit costs match.estBits reads from the predefined FSE tables (enc_best.go:48-53): cost[0..31] offset codes,
// cost[32..95] match-length codes.  (For the emulator test of the device kernel, which takes them as an input.)
void kco_zstd_best_costs(int32_t* cost) {
    for (int i = 0; i < 96; i++) cost[i] = 0;
    for (int i = 0; i <= zfse::maxOffsetLengthSymbol; i++) {
        const zfse::SymbolTransform t = zfse::predef().enc[1].symbolTT[i];
        cost[i] = (int32_t)t.outBits + (int32_t)(t.deltaNbBits >> 16);
    }
    for (int i = 0; i <= zfse::maxMatchLengthSymbol; i++) {
        const zfse::SymbolTransform t = zfse::predef().enc[2].symbolTT[i];
        cost[32 + i] = (int32_t)t.outBits + (int32_t)(t.deltaNbBits >> 16);
    }
}

// compress.ShannonEntropyBits (compressible.go:68-85)
int64_t kco_shannon_entropy_bits(const uint8_t* p, uint64_t n) { return (int64_t)ShannonEntropyBits(p, (size_t)n); }
double kco_go_log2(double x) { return golog::log2(x); }

uint64_t kco_xxh64(const uint8_t* p, uint64_t n) {
    XXH64 h;
    h.Write(p, (size_t)n);
    return h.Sum64();
}
// chunked variant to exercise Digest.Write buffering (xxhash_test.go testDigest)
uint64_t kco_xxh64_chunked(const uint8_t* p, uint64_t n, uint64_t chunk) {
    XXH64 h;
    for (uint64_t i = 0; i < n; i += chunk) h.Write(p + i, (size_t)std::min(chunk, n - i));
    return h.Sum64();
}

int32_t kco_zstd_matchlen(const uint8_t* a, uint64_t alen, const uint8_t* b) { return matchLen(a, (size_t)alen, b); }
uint32_t kco_zstd_hashlen(uint64_t u, uint32_t length, uint32_t mls) { return hashLen(u, (uint8_t)length, (uint8_t)mls); }

// huff0.Compress1X / Compress4X on a fresh Scratch (WantLogLess as given).
// Returns: >=0 output size; -1 ErrIncompressible; -2 ErrUseRLE; -3 ErrTooBig; -4 internal.
int64_t kco_huff0_compress(const uint8_t* in, uint64_t n, int four, int want_log_less, uint8_t* out, uint64_t cap) {
    huff0::Scratch s;
    s.WantLogLess = (uint8_t)want_log_less;
    bool reUsed = false;
    huff0::Err e = huff0::compress(in, (size_t)n, &s, four != 0, &reUsed);
    if (e != huff0::OK) return -(int64_t)e;
    if (s.Out.size() > cap) return -5;
    memcpy(out, s.Out.data(), s.Out.size());
    return (int64_t)s.Out.size();
}

// fse.Compress (byte FSE) on a fresh Scratch. Returns >=0 size, -1 incompressible, -2 RLE, -3 internal.
int64_t kco_fse_compress(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap) {
    std::unique_ptr<fseb::Scratch> s(new fseb::Scratch());
    fseb::Err e = fseb::Compress(in, (size_t)n, s.get());
    if (e != fseb::OK) return -(int64_t)e;
    if (s->Out.size() > cap) return -5;
    memcpy(out, s->Out.data(), s->Out.size());
    return (int64_t)s->Out.size();
}

// ---- S2 ----
// Inspection of a zstd frame (analysis tooling, e.g. the provenance of zstd/testdata/z000028.zst): one text line per block
// with its type, literal type, sequence count, compression modes and the first sequences (REPn = repeat-offset code n).
int64_t kco_zstd_inspect(const uint8_t* enc, uint64_t n, char* out, uint64_t cap) {
    zdec::FrameDec fd;
    std::string tr;
    fd.trace = &tr;
    Bytes content;
    const size_t used = fd.decodeFrame(enc, (size_t)n, nullptr, &content);
    char tb[192];
    snprintf(tb, sizeof(tb), "\nframe: consumed=%zu decoded=%zu repeat codes REP1=%zu REP2=%zu REP3=%zu", used, content.size(), fd.repSeen[1], fd.repSeen[2], fd.repSeen[3]);
    tr += tb;
    if (tr.size() + 1 > cap) return -2;
    memcpy(out, tr.c_str(), tr.size() + 1);
    return (int64_t)tr.size();
}
int64_t kco_s2_max_encoded_len(int64_t n) { return s2::MaxEncodedLen(n); }
int64_t kco_s2_encode(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::Encode(dst, cap, src, (size_t)n); }
// encodeBlock only (no varint header); 0 == incompressible.  The WriterCustomEncoder contract.
int64_t kco_s2_encode_better(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::EncodeBetter(dst, cap, src, (size_t)n); }
int64_t kco_s2_encode_snappy(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::EncodeSnappy(dst, cap, src, (size_t)n); }
// test diagnostics: number of late raw fallbacks (blockenc.go:811-817) on non-last blocks that changed the carried offsets since the
// last reset — the situation the device path's speculation re-run handles
uint64_t kco_debug_late_raw_pops(int reset) { const uint64_t v = kco::lateRawPops().load(); if (reset) kco::lateRawPops().store(0); return v; }
int64_t kco_s2_encode_snappy_best(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::EncodeSnappyBest(dst, cap, src, (size_t)n); }
int64_t kco_s2_encode_best(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::EncodeBest(dst, cap, src, (size_t)n); }
int64_t kco_s2_encode_snappy_better(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::EncodeSnappyBetter(dst, cap, src, (size_t)n); }
int64_t kco_s2_encode_block(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) {
    if (cap < (uint64_t)s2::MaxEncodedLen((int64_t)n)) return -2;
    return s2::encodeBlock(dst, src, (size_t)n);
}
int64_t kco_s2_emit_literal(uint8_t* dst, const uint8_t* lit, uint64_t n) { return s2::emitLiteral(dst, lit, (size_t)n); }
int64_t kco_s2_emit_copy(uint8_t* dst, int64_t offset, int64_t length) { return s2::emitCopy(dst, (int)offset, (int)length); }
int64_t kco_s2_emit_repeat(uint8_t* dst, int64_t offset, int64_t length) { return s2::emitRepeat(dst, (int)offset, (int)length); }
int64_t kco_s2_decode(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::Decode(dst, cap, src, (size_t)n); }
// s2.Encode (snappy 0) / s2.EncodeSnappy (1) as an amd64 build of the reference writes them (kco_s2_asm.h)
int64_t kco_s2_encode_asm(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, int snappy) {
    return s2::EncodeAsm(dst, cap, src, (size_t)n, snappy != 0);
}
// level: 0 s2.Encode, 1 EncodeBetter, 2 EncodeSnappy, 3 EncodeSnappyBetter
int64_t kco_s2_encode_asm_level(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, int level) {
    return s2::EncodeAsm(dst, cap, src, (size_t)n, level == 2 || level == 3, level == 1 || level == 3);
}

uint32_t kco_s2_crc(const uint8_t* p, uint64_t n) { return s2::crc(p, (size_t)n); }

// s2.Writer output for the blocks (stream identifier optional): chunks back to back; out_off[i] = start of chunk i.
int64_t kco_s2_encode_stream(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                             uint64_t* out_off, int with_stream_id) {
    static const uint8_t magic[10] = {0xff, 0x06, 0x00, 0x00, 'S', '2', 's', 'T', 'w', 'O'};
    uint64_t pos = 0;
    if (with_stream_id) { if (dst_cap < 10) return -2; memcpy(dst, magic, 10); pos = 10; }
    std::vector<uint8_t> tmp;
    for (uint32_t i = 0; i < n_blocks; i++) {
        const size_t n = (size_t)(blk_off[i + 1] - blk_off[i]);
        tmp.resize(n + 32);
        const int64_t r = s2::EncodeChunk(tmp.data(), src + blk_off[i], n);
        out_off[i] = pos;
        if (pos + (uint64_t)r > dst_cap) return -2;
        memcpy(dst + pos, tmp.data(), (size_t)r);
        pos += (uint64_t)r;
    }
    out_off[n_blocks] = pos;
    return (int64_t)pos;
}
// The same for a Writer at any level: 0 default, 1 WriterBetterCompression, 2 WriterSnappyCompat, 3 both, 4 WriterBestCompression,
// 5 best + Snappy compatible — (*Writer).encodeBlock, s2/writer.go:1053-1091, inside the chunk framing of :414-451.
int64_t kco_s2_encode_stream_level(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                                   uint64_t* out_off, int with_stream_id, int level) {
    static const uint8_t magic[10] = {0xff, 0x06, 0x00, 0x00, 'S', '2', 's', 'T', 'w', 'O'};
    uint64_t pos = 0;
    if (with_stream_id) { if (dst_cap < 10) return -2; memcpy(dst, magic, 10); pos = 10; }
    std::vector<uint8_t> tmp;
    for (uint32_t i = 0; i < n_blocks; i++) {
        const size_t n = (size_t)(blk_off[i + 1] - blk_off[i]);
        const uint8_t* p = src + blk_off[i];
        tmp.assign((size_t)s2::MaxEncodedLen((int64_t)n) + 32, 0);
        uint8_t* o = tmp.data();
        const uint32_t checksum = s2::crc(p, n);
        uint8_t chunkType = 0x01;
        size_t chunkLen = 4 + n;
        const int v = s2::putUvarint(o + 8, (uint64_t)n);
        int n2 = 0;
        switch (level) {
        case 1: n2 = n < (size_t)s2::minNonLiteralBlockSize ? 0 : s2::encodeBlockBetter(o + 8 + v, p, n); break;
        case 2: n2 = n < (size_t)s2::minNonLiteralBlockSize ? 0 : (n <= ((size_t)64 << 10) ? s2::encodeBlockGoT<uint16_t, 5, true>(o + 8 + v, p, n) : s2::encodeBlockGoT<uint32_t, 6, true>(o + 8 + v, p, n)); break;
        case 3: n2 = n < (size_t)s2::minNonLiteralBlockSize ? 0 : s2::encodeBlockBetterSnappy(o + 8 + v, p, n); break;
        case 4: n2 = n < (size_t)s2::minNonLiteralBlockSize ? 0 : s2::encodeBlockBest(o + 8 + v, p, n); break;
        case 5: n2 = n < (size_t)s2::minNonLiteralBlockSize ? 0 : s2::encodeBlockBestSnappy(o + 8 + v, p, n); break;
        default: n2 = s2::encodeBlock(o + 8 + v, p, n);
        }
        if (n2 > 0) { chunkType = 0x00; chunkLen = 4 + (size_t)v + (size_t)n2; }
        else memcpy(o + 8, p, n);
        o[0] = chunkType;
        o[1] = (uint8_t)chunkLen; o[2] = (uint8_t)(chunkLen >> 8); o[3] = (uint8_t)(chunkLen >> 16);
        o[4] = (uint8_t)checksum; o[5] = (uint8_t)(checksum >> 8); o[6] = (uint8_t)(checksum >> 16); o[7] = (uint8_t)(checksum >> 24);
        const uint64_t r = 4 + chunkLen;
        out_off[i] = pos;
        if (pos + r > dst_cap) return -2;
        memcpy(dst + pos, o, (size_t)r);
        pos += r;
    }
    out_off[n_blocks] = pos;
    return (int64_t)pos;
}
int64_t kco_s2_decode_stream(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) { return s2::DecodeStream(dst, cap, src, (size_t)n); }

static int64_t s2_encode_blocks_impl(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                                     uint64_t* out_off, int threads, int level) {
    if (threads < 1) threads = 1;
    std::vector<Bytes> outs(n_blocks);
    std::atomic<uint32_t> next(0);
    auto worker = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= n_blocks) break;
            size_t n = (size_t)(blk_off[i + 1] - blk_off[i]);
            outs[i].resize((size_t)s2::MaxEncodedLen((int64_t)n));
            int64_t r = level == 1 ? s2::EncodeBetter(outs[i].data(), outs[i].size(), src + blk_off[i], n)
                      : level == 2 ? s2::EncodeSnappy(outs[i].data(), outs[i].size(), src + blk_off[i], n)
                      : level == 3 ? s2::EncodeSnappyBetter(outs[i].data(), outs[i].size(), src + blk_off[i], n)
                      : level == 4 ? s2::EncodeBest(outs[i].data(), outs[i].size(), src + blk_off[i], n)
                      : level == 5 ? s2::EncodeSnappyBest(outs[i].data(), outs[i].size(), src + blk_off[i], n)
                                   : s2::Encode(outs[i].data(), outs[i].size(), src + blk_off[i], n);
            outs[i].resize((size_t)r);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n_blocks; i++) {
        out_off[i] = pos;
        if (pos + outs[i].size() > dst_cap) return -2;
        memcpy(dst + pos, outs[i].data(), outs[i].size());
        pos += outs[i].size();
    }
    out_off[n_blocks] = pos;
    return (int64_t)pos;
}
// N x s2.Encode(nil, block) / N x s2.EncodeBetter(nil, block) on `threads` host threads
int64_t kco_s2_encode_blocks(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                             uint64_t* out_off, int threads) {
    return s2_encode_blocks_impl(src, blk_off, n_blocks, dst, dst_cap, out_off, threads, 0);
}
int64_t kco_s2_encode_blocks_better(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                                    uint64_t* out_off, int threads) {
    return s2_encode_blocks_impl(src, blk_off, n_blocks, dst, dst_cap, out_off, threads, 1);
}
int64_t kco_s2_encode_blocks_snappy(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                                    uint64_t* out_off, int threads) {
    return s2_encode_blocks_impl(src, blk_off, n_blocks, dst, dst_cap, out_off, threads, 2);
}
int64_t kco_s2_encode_blocks_snappy_better(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                                           uint64_t* out_off, int threads) {
    return s2_encode_blocks_impl(src, blk_off, n_blocks, dst, dst_cap, out_off, threads, 3);
}
// the same at any level 0..5 (4: s2.EncodeBest, 5: s2.EncodeSnappyBest)
int64_t kco_s2_encode_blocks_level(const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks, uint8_t* dst, uint64_t dst_cap,
                                   uint64_t* out_off, int threads, int level) {
    return s2_encode_blocks_impl(src, blk_off, n_blocks, dst, dst_cap, out_off, threads, level);
}

}  // extern "C"
