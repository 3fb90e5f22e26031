// kc_s2_api.cpp — S2 entry points (kernels in kc_s2.hip, kc_s2_lds.hip, kc_s2_best.hip) and the device decoders used as verifiers.
#include "kc_hostpipe.h"

extern "C" {

// ---------------------------------------------------------------------------------------
// S2 (kernels in kc_s2.hip)
// ---------------------------------------------------------------------------------------
int64_t kc_s2_max_encoded_len(int64_t srcLen) {  // s2/encode.go:389-418 (64-bit int)
    uint64_t n = (uint64_t)srcLen;
    if (n > 0xffffffffULL) return -1;
    int lb = n == 0 ? 0 : 64 - __builtin_clzll(n);
    n = n + (uint64_t)((lb + 7) / 7);
    int64_t extra = srcLen == 0 ? 0 : (srcLen < 60 ? 1 : (srcLen < (1 << 8) ? 2 : (srcLen < (1 << 16) ? 3 : (srcLen < (1 << 24) ? 4 : 5))));
    n += (uint64_t)extra;
    if (n > 0xffffffffULL) return -1;
    return (int64_t)n;
}

// phase 0: the whole call.  phase 1 (_begin): everything up to and including the encoder kernel, no wait — d_dst / dst_cap / out_off unused;
// phase 2 (_end_at): sizes -> offsets -> compaction to the d_dst named NOW, offsets to the host, wait (the blocks of a batch leave their
// staging slots only here, which is what lets one batch run as several launches with one contiguous output).
static kc_status s2_encode_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                               uint64_t dst_cap, uint64_t* out_off, int framed, int with_stream_id, int level = KC_S2_LEVEL_DEFAULT,
                               ChunkFeed* feed = nullptr, int phase = 0) {
    if (phase == 2) {
        if (!c || !out_off || !d_dst) return KC_ERR_BAD_ARG;
        if (c->s2_pend_n == 0) { c->err = "no S2 batch in flight on this context"; return KC_ERR_BAD_ARG; }
        HIPCHK(c, hipSetDevice(c->device));
        n = c->s2_pend_n;
        if (c->plan.stage_off[n] > dst_cap) { c->err = "dst_cap smaller than the sum of MaxEncodedLen(block)"; return KC_ERR_DST_TOO_SMALL; }
        c->s2_pend_n = 0;
        hipStream_t st = c->stream;
        kc_launch_scan_sizes((const uint32_t*)c->out_size.p, n, (uint64_t*)c->out_off.p, st);
        kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p, (const uint32_t*)c->out_size.p,
                          (const uint64_t*)c->out_off.p, d_dst, n, st);
        HIPCHK(c, hipEventRecord(c->ev[2], st));
        HIPCHK(c, hipMemcpyAsync(out_off, c->out_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, hipGetLastError());
        float t01 = 0, t12 = 0;
        (void)hipEventElapsedTime(&t01, c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
        c->last.match_ms = t01;
        c->last.other_ms = t12;
        c->last.total_ms = t01 + t12;
        return KC_OK;
    }
    if (!c || !blk_off || (phase == 0 && !out_off) || (n && (!d_src || (phase == 0 && !d_dst)))) return KC_ERR_BAD_ARG;
    if (phase == 1 && (framed || feed || n == 0 || level >= KC_S2_LEVEL_BEST)) { c->err = "begin / end: bare blocks of one batch below the best levels"; return KC_ERR_UNSUPPORTED; }
    if (c->s2_pend_n != 0) { c->err = "an S2 batch is in flight on this context (kc_s2_encode_blocks_lvl_dev_begin without _end_at)"; return KC_ERR_BAD_ARG; }  // its staging slots and sizes are this context's
    if (feed && (framed || n == 0)) { c->err = "chunk feed: bare blocks only"; return KC_ERR_INTERNAL; }
    // s2.WriterUncompressed: a level of the writer (writer.go:951; encodeBlock returns 0 for it, :455-480): framed only, every block one
    // uncompressed chunk — served by the LDS-table kernels' stored path (wave-parallel CRC32C + copy), whatever the batch
    const bool stored_only = level == KC_S2_LEVEL_UNCOMPRESSED;
    if (stored_only) {
        if (!framed) { c->err = "KC_S2_LEVEL_UNCOMPRESSED is a level of the framed stream (s2.WriterUncompressed)"; return KC_ERR_BAD_ARG; }
        level = KC_S2_LEVEL_DEFAULT;
    }
    if (level < KC_S2_LEVEL_DEFAULT || level > KC_S2_LEVEL_SNAPPY_BEST) { c->err = "unknown S2 level"; return KC_ERR_UNSUPPORTED; }
    if (level >= KC_S2_LEVEL_BEST && feed) { c->err = "the best levels are not chunk-fed"; return KC_ERR_UNSUPPORTED; }
    c->err.clear();
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t lead = (framed && with_stream_id) ? 10 : 0;
    if (lead) {
        static const uint8_t magic[10] = {0xff, 0x06, 0x00, 0x00, 'S', '2', 's', 'T', 'w', 'O'};  // magicChunk, s2/s2.go:79
        if (dst_cap < 10) return KC_ERR_DST_TOO_SMALL;
        HIPCHK(c, hipMemcpyAsync(d_dst, magic, 10, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (n == 0) { out_off[0] = lead; return KC_OK; }
    hipStream_t st = c->stream;
    std::vector<uint64_t>& rel = c->plan.rel_off;   // in the context: the asynchronous copies below outlive this call when chunk-fed
    std::vector<uint64_t>& so = c->plan.stage64;     // staging slots: whole 64-byte lines (the kernel stores its output line by line)
    std::vector<uint64_t>& reg = c->plan.stage_off;  // 16-byte aligned bounds: what dst_cap is checked against, chunk regions when chunk-fed
    rel.resize(n + 1);
    so.resize(n + 1);
    reg.resize(n + 1);
    uint64_t acc = 0, acc16 = 0, maxLen = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (blk_off[i + 1] < blk_off[i]) { c->err = "blk_off not ascending"; return KC_ERR_BAD_ARG; }
        const uint64_t len = blk_off[i + 1] - blk_off[i];
        maxLen = std::max(maxLen, len);
        if (len > KC_S2_MAX_BLOCK) { c->err = "S2 block larger than 1 GiB not served by the device path"; return KC_ERR_UNSUPPORTED; }
        // a framed block is a chunk of the stream format: its header holds a 24-bit length and the reference's Reader refuses chunks of
        // blocks above s2.maxBlockSize (4 MiB; s2/s2.go, writer.go:986-989 WriterBlockSize) — the 1 GiB bound is for bare blocks only
        if (framed && len > KC_S2_MAX_FRAMED_BLOCK) { c->err = "framed S2 block larger than 4 MiB (s2.maxBlockSize): not a valid chunk of the stream format"; return KC_ERR_BAD_ARG; }
        rel[i] = blk_off[i] - blk_off[0];
        so[i] = acc;
        reg[i] = acc16;
        acc += ((uint64_t)kc_s2_max_encoded_len((int64_t)len) + (framed ? 8 : 0) + 63) & ~(uint64_t)63;
        acc16 += ((uint64_t)kc_s2_max_encoded_len((int64_t)len) + (framed ? 8 : 0) + 15) & ~(uint64_t)15;
    }
    rel[n] = blk_off[n] - blk_off[0];
    so[n] = acc;
    reg[n] = acc16;
    if (phase == 0 && acc16 + lead > dst_cap) { c->err = "dst_cap smaller than the sum of MaxEncodedLen(block)"; return KC_ERR_DST_TOO_SMALL; }
    // s2.Encode / s2.EncodeSnappy: the LDS-table kernel (one wave per block, ~1 ms per 64 KiB block whatever the batch) while the
    // blocks in flight cannot cover the HBM-table kernel's latency (measured crossover: profiles/r03_crossover_s2.csv)
    // The best levels are pure Go in the reference — one form on every platform (s2/encode_best.go) — so the variant does not
    // apply to them: an amd64 context (the Go shim's default on amd64 builds) encodes them like any other.
    const int s2var = level >= KC_S2_LEVEL_BEST ? KC_S2_VARIANT_GO : (int)c->cfg.s2_variant;
    // (the LDS kernel keeps positions in 24 bits: a batch with a block of 16 MiB or more goes through the HBM-table kernel whole)
    const bool lds = stored_only || ((level == KC_S2_LEVEL_DEFAULT || level == KC_S2_LEVEL_SNAPPY) && feed == nullptr && c->cfg.match_path != KC_PATH_HBM &&
                     maxLen < ((uint64_t)1 << 24) && (c->cfg.match_path == KC_PATH_LDS || (int64_t)n <= c->cfg.s2_lds_max_blocks));
    c->last_path = lds ? KC_PATH_LDS : KC_PATH_HBM;
    kc_status s;
    if ((s = ensure(c, c->unit_off, (n + 1) * 8)) || (s = ensure(c, c->stage_off, (n + 1) * 8)) ||
        (s = ensure(c, c->out_off, (n + 1 + (feed ? feed->cut.size() : 0)) * 8)) || (s = ensure(c, c->stage, acc + 64)) || (s = ensure(c, c->out_size, (size_t)n * 4)) ||
        (!lds && (s = ensure(c, c->tables, (size_t)n * kc_s2_table_bytes(level, maxLen, s2var)))))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->stage_off.p, so.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    const size_t tab_bytes = lds ? 0 : (size_t)n * kc_s2_table_bytes(level, maxLen, s2var);
    if (!lds) {
        c->tab_owner = 0;
        if (c->preclear_bytes >= tab_bytes && c->preclear_ptr == c->tables.p) {  // zeroed behind the previous batch's encoder (below)
            c->preclear_bytes = 0;
            HIPCHK(c, hipStreamWaitEvent(st, c->ev_preclear, 0));
        } else {
            kc_status sc = tables_claim(c, st);
            if (sc != KC_OK) return sc;
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, tab_bytes, st));
        }
    }
    KcS2Params P;
    P.src = d_src + blk_off[0];
    P.blk_off = (const uint64_t*)c->unit_off.p;
    P.stage_off = (const uint64_t*)c->stage_off.p;
    P.stage = (uint8_t*)c->stage.p;
    P.out_size = (uint32_t*)c->out_size.p;
    P.tables = (uint32_t*)c->tables.p;
    P.n_blocks = n;
    P.framed = framed;
    P.level = level;
    P.spec_w0 = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : 2;
    P.spec_w0b = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : 4;
    P.spec_grow = c->cfg.spec_grow >= 0 ? (int)c->cfg.spec_grow : 1;
    if (P.spec_w0 < 1) P.spec_w0 = 1;
    if (P.spec_w0b < 1) P.spec_w0b = 1;
    P.table_stride = (uint32_t)(kc_s2_table_bytes(level, maxLen, s2var) / 4);
    P.variant = (int32_t)s2var;
    P.stored_only = stored_only ? 1 : 0;
    if (feed) {
        // the source is still arriving: per chunk, encode + compaction on the chunk's stream behind its H2D copy; frames of chunk k
        // at d_dst + reg[cut[k]], local offsets in out_off[cut[k] + k ...].  The caller synchronises (s2_feed_finish).
        HIPCHK(c, hipEventRecord(c->ev[6], st));
        feed->loc_off = (uint64_t*)c->out_off.p;
        const size_t nchunk = feed->cut.size() - 1;
        for (size_t k = 0; k < nchunk; k++) {
            if (!feed->wait_recorded(k)) { c->err = "host pipeline: staging failed"; return KC_ERR_HIP; }
            hipStream_t sk = feed->streams[k % feed->streams.size()];
            const uint32_t u0 = feed->cut[k], nk = feed->cut[k + 1] - u0;
            HIPCHK(c, hipStreamWaitEvent(sk, c->ev[6], 0));
            HIPCHK(c, hipStreamWaitEvent(sk, feed->landed[k], 0));
            KcS2Params Pk = P;
            Pk.blk_off += u0;
            Pk.stage_off += u0;
            Pk.out_size += u0;
            Pk.tables += (size_t)u0 * P.table_stride;
            Pk.n_blocks = nk;
            kc_launch_s2_encode(Pk, sk);
            kc_launch_scan_sizes(Pk.out_size, nk, feed->loc_off + u0 + k, sk);
            kc_launch_compact((const uint8_t*)c->stage.p, Pk.stage_off, Pk.out_size, feed->loc_off + u0 + k, d_dst + reg[u0], nk, sk);
            HIPCHK(c, hipEventRecord(feed->done[k], sk));
        }
        for (size_t k = 0; k < nchunk; k++) HIPCHK(c, hipStreamWaitEvent(st, feed->done[k], 0));
        HIPCHK(c, hipGetLastError());
        return KC_OK;
    }
    if (level >= KC_S2_LEVEL_BEST) {
        kc_launch_s2_best(P, st);
    } else if (lds) {
        bool any_small = false, any_big = false;
        for (uint32_t i = 0; i < n; i++) ((blk_off[i + 1] - blk_off[i]) <= ((uint64_t)64 << 10) ? any_small : any_big) = true;
        P.spec_w0 = (int32_t)c->cfg.s2_lds_spec_w0;
        kc_launch_s2_encode_lds(P, any_small, any_big, st);
    } else {
        kc_launch_s2_encode(P, st);
    }
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    if (!lds && c->lane_preclear && c->stream2 != nullptr && level < KC_S2_LEVEL_BEST) {  // the arena for the next batch of this lane (kc_host.h)
        if (c->ev_preclear == nullptr) HIPCHK(c, hipEventCreateWithFlags(&c->ev_preclear, hipEventDisableTiming));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[1], 0));
        HIPCHK(c, hipMemsetAsync(c->tables.p, 0, tab_bytes, c->stream2));
        HIPCHK(c, hipEventRecord(c->ev_preclear, c->stream2));
        c->preclear_ptr = c->tables.p;
        c->preclear_bytes = tab_bytes;
    }
    if (phase == 1) {  // the rest is _end_at's
        HIPCHK(c, hipGetLastError());
        c->s2_pend_n = n;
        return KC_OK;
    }
    kc_launch_scan_sizes((const uint32_t*)c->out_size.p, n, (uint64_t*)c->out_off.p, st);
    kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p, (const uint32_t*)c->out_size.p,
                      (const uint64_t*)c->out_off.p, d_dst + lead, n, st);
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    HIPCHK(c, hipMemcpyAsync(out_off, c->out_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    for (uint32_t i = 0; i <= n; i++) out_off[i] += lead;
    float t01 = 0, t12 = 0;
    (void)hipEventElapsedTime(&t01, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
    c->last.match_ms = t01;
    c->last.other_ms = t12;
    c->last.total_ms = t01 + t12;
    return KC_OK;
}

// zstd frame decode over N units on the device (verifier): decode, then XXH64 of the output against the stored checksum.
kc_status kc_zstd_decode_units_dev(kc_ctx* c, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n, uint8_t* d_dst,
                                   const uint64_t* dst_off, uint32_t* status) {
    return kc_zstd_decode_units_dict_dev(c, d_enc, enc_off, n, d_dst, dst_off, status, nullptr, 0);
}

kc_status kc_zstd_decode_units_dict_dev(kc_ctx* c, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n, uint8_t* d_dst,
                                        const uint64_t* dst_off, uint32_t* status, const uint8_t* dict, uint64_t dict_len) {
    if (!c || !enc_off || !dst_off || !status || (n && (!d_enc || !d_dst))) return KC_ERR_BAD_ARG;
    if (dict_len > ((uint64_t)1 << 30)) return KC_ERR_BAD_ARG;
    c->err.clear();
    if (n == 0) return KC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    kc_status s;
    const uint32_t lit_stride = (128u << 10) + 64u;
    if ((s = ensure(c, c->unit_off, (size_t)(n + 1) * 8)) || (s = ensure(c, c->stage_off, (size_t)(n + 1) * 8)) ||
        (s = ensure(c, c->out_size, (size_t)n * 4)) || (s = ensure(c, c->redo, (size_t)n * 4)) || (s = ensure(c, c->popmask, (size_t)n * 4)) ||
        (s = ensure(c, c->xxh, (size_t)n * 8)) || (s = ensure(c, c->lits, (size_t)n * lit_stride)))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, enc_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->stage_off.p, dst_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    KcZstdDecParams P;
    P.enc = d_enc;
    P.enc_off = (const uint64_t*)c->unit_off.p;
    P.dst = d_dst;
    P.dst_off = (const uint64_t*)c->stage_off.p;
    P.lits = (uint8_t*)c->lits.p;
    P.lit_stride = lit_stride;
    P.status = (uint32_t*)c->out_size.p;
    P.crc_stored = (uint32_t*)c->redo.p;
    P.has_crc = (uint32_t*)c->popmask.p;
    P.n_units = n;
    P.dict = nullptr;
    P.dict_len = 0;
    if (dict != nullptr && dict_len > 0) {
        if ((s = ensure(c, c->dictbuf, (size_t)dict_len + 64))) return s;
        HIPCHK(c, hipMemcpyAsync(c->dictbuf.p, dict, (size_t)dict_len, hipMemcpyHostToDevice, st));
        P.dict = (const uint8_t*)c->dictbuf.p;
        P.dict_len = (uint32_t)dict_len;
    }
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    kc_launch_zstd_decode(P, st);
    kc_launch_xxh64(d_dst, (const uint64_t*)c->stage_off.p, n, (uint64_t*)c->xxh.p, st);
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    std::vector<uint32_t> stored(n), has(n);
    std::vector<uint64_t> hashes(n);
    HIPCHK(c, hipMemcpyAsync(status, c->out_size.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(stored.data(), c->redo.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(has.data(), c->popmask.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hashes.data(), c->xxh.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    for (uint32_t i = 0; i < n; i++)
        if (status[i] == 0 && has[i] && (uint32_t)hashes[i] != stored[i]) status[i] = 30;  // checksum mismatch (framedec.go:310-325)
    float t = 0;
    (void)hipEventElapsedTime(&t, c->ev[0], c->ev[1]);
    c->last = kc_timings{t, t, 0, 0, 0};
    return KC_OK;
}

// s2.Decode over N blocks on the device (verifier).  status[i] (host) receives 0 or the first error of block i.
kc_status kc_s2_decode_blocks_dev(kc_ctx* c, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n, uint8_t* d_dst,
                                  const uint64_t* dst_off, uint32_t* status) {
    if (!c || !enc_off || !dst_off || !status || (n && (!d_enc || !d_dst))) return KC_ERR_BAD_ARG;
    c->err.clear();
    if (n == 0) return KC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    kc_status s;
    if ((s = ensure(c, c->unit_off, (size_t)(n + 1) * 8)) || (s = ensure(c, c->stage_off, (size_t)(n + 1) * 8)) ||
        (s = ensure(c, c->out_size, (size_t)n * 4)))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, enc_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->stage_off.p, dst_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    KcS2DecParams P;
    P.enc = d_enc;
    P.enc_off = (const uint64_t*)c->unit_off.p;
    P.dst = d_dst;
    P.dst_off = (const uint64_t*)c->stage_off.p;
    P.status = (uint32_t*)c->out_size.p;
    P.n_blocks = n;
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    kc_launch_s2_decode(P, st);
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipMemcpyAsync(status, c->out_size.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float t = 0;
    (void)hipEventElapsedTime(&t, c->ev[0], c->ev[1]);
    c->last = kc_timings{t, t, 0, 0, 0};
    return KC_OK;
}

// s2_encode_dev in batches that fit the scratch budget: the HBM path keeps a table per block (64 KiB default / snappy, 288-576 KiB
// better), whatever the block's length, plus a MaxEncodedLen staging slot — half a million small blocks would ask for > 100 GiB.
static kc_status s2_encode_dev_budgeted(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                        uint64_t dst_cap, uint64_t* out_off, int framed, int with_stream_id, int level) {
    if (!c || !blk_off || !out_off) return KC_ERR_BAD_ARG;
    if (level < KC_S2_LEVEL_DEFAULT || level > KC_S2_LEVEL_SNAPPY_BEST || n == 0) return s2_encode_dev(c, d_src, blk_off, n, d_dst, dst_cap, out_off, framed, with_stream_id, level);
    uint64_t maxLen = 0;
    for (uint32_t i = 0; i < n; i++) if (blk_off[i + 1] >= blk_off[i]) maxLen = std::max(maxLen, blk_off[i + 1] - blk_off[i]);
    const uint64_t tb = kc_s2_table_bytes(level, maxLen, level >= KC_S2_LEVEL_BEST ? KC_S2_VARIANT_GO : (int)c->cfg.s2_variant);
    uint64_t budget = scratch_budget(c);
    std::vector<uint64_t> tmp;
    uint64_t pos = 0;
    uint32_t i0 = 0;
    while (i0 < n) {
        uint32_t i1 = i0;
        uint64_t scratch = 0;
        while (i1 < n) {
            const uint64_t len = blk_off[i1 + 1] >= blk_off[i1] ? blk_off[i1 + 1] - blk_off[i1] : 0;
            const uint64_t us = tb + (((uint64_t)std::max<int64_t>(0, kc_s2_max_encoded_len((int64_t)len)) + 8 + 63) & ~(uint64_t)63);
            if (i1 > i0 && (scratch + us) + ((scratch + us) >> 3) > budget) break;
            scratch += us;
            i1++;
        }
        if (i0 == 0 && i1 == n) {  // the usual case: one batch
            const kc_status s1 = s2_encode_dev(c, d_src, blk_off, n, d_dst, dst_cap, out_off, framed, with_stream_id, level);
            c->last_batches = s1 == KC_OK ? 1 : 0;  // (0: nothing was encoded on the device — the Go shim's tests tell a fallback by it)
            return s1;
        }
        if (i0 == 0) c->last_batches = 0;
        const uint32_t nb = i1 - i0;
        tmp.resize(nb + 1);
        c->oom = false;
        kc_status s = s2_encode_dev(c, d_src, blk_off + i0, nb, d_dst + pos, dst_cap - pos, tmp.data(), framed, i0 == 0 ? with_stream_id : 0, level);
        if (s == KC_ERR_UNSUPPORTED && c->oom && nb > 1 && budget > ((uint64_t)64 << 20)) {
            budget /= 2;  // another process took device memory since hipMemGetInfo
            c->err.clear();
            continue;
        }
        if (s != KC_OK) return s;
        for (uint32_t k = 0; k <= nb; k++) out_off[i0 + k] = pos + tmp[k];
        pos += tmp[nb];
        i0 = i1;
        c->last_batches++;
    }
    return KC_OK;
}

kc_status kc_s2_encode_blocks_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                  uint64_t dst_cap, uint64_t* out_off) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 0, 0, KC_S2_LEVEL_DEFAULT);
}

kc_status kc_s2_encode_stream_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                  uint64_t dst_cap, uint64_t* out_off, int with_stream_id) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 1, with_stream_id, KC_S2_LEVEL_DEFAULT);
}

kc_status kc_s2_encode_blocks_lvl_dev(kc_ctx* c, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                      uint64_t dst_cap, uint64_t* out_off) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 0, 0, level);
}

kc_status kc_s2_encode_blocks_lvl_dev_begin(kc_ctx* c, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n) {
    if (!c || !blk_off || !d_src || n == 0) return KC_ERR_BAD_ARG;
    // one device batch: what the scratch budget would cut goes through the blocking call
    uint64_t maxLen = 0;
    for (uint32_t i = 0; i < n; i++) if (blk_off[i + 1] >= blk_off[i]) maxLen = std::max(maxLen, blk_off[i + 1] - blk_off[i]);
    if (level >= KC_S2_LEVEL_DEFAULT && level < KC_S2_LEVEL_BEST) {
        const uint64_t tb = kc_s2_table_bytes(level, maxLen, (int)c->cfg.s2_variant);
        uint64_t scratch = 0;
        for (uint32_t i = 0; i < n; i++) scratch += tb + (((uint64_t)std::max<int64_t>(0, kc_s2_max_encoded_len((int64_t)(blk_off[i + 1] >= blk_off[i] ? blk_off[i + 1] - blk_off[i] : 0))) + 8 + 63) & ~(uint64_t)63);
        if (scratch + (scratch >> 3) > scratch_budget(c)) { c->err = "begin / end serves one device batch; use kc_s2_encode_blocks_lvl_dev for larger inputs"; return KC_ERR_UNSUPPORTED; }
    }
    return s2_encode_dev(c, d_src, blk_off, n, nullptr, 0, nullptr, 0, 0, level, nullptr, 1);
}

kc_status kc_s2_encode_blocks_lvl_dev_end_at(kc_ctx* c, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c) return KC_ERR_BAD_ARG;
    return s2_encode_dev(c, nullptr, nullptr, 0, d_dst, dst_cap, out_off, 0, 0, KC_S2_LEVEL_DEFAULT, nullptr, 2);
}

kc_status kc_s2_encode_stream_lvl_dev(kc_ctx* c, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                      uint64_t dst_cap, uint64_t* out_off, int with_stream_id) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 1, with_stream_id, level);
}

kc_status kc_s2_encode_blocks(kc_ctx* c, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* dst, uint64_t dst_cap,
                              uint64_t* out_off) {
    return kc_s2_encode_blocks_lvl(c, KC_S2_LEVEL_DEFAULT, src, blk_off, n, dst, dst_cap, out_off);
}

kc_status kc_s2_encode_blocks_lvl(kc_ctx* c, int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* dst, uint64_t dst_cap,
                                  uint64_t* out_off) {
    if (!c || !blk_off || !out_off || (n && (!src || !dst))) return KC_ERR_BAD_ARG;
    if (level < KC_S2_LEVEL_DEFAULT || level > KC_S2_LEVEL_SNAPPY_BEST) { c->err = "unknown S2 level"; return KC_ERR_UNSUPPORTED; }
    c->err.clear();
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) { out_off[0] = 0; return KC_OK; }
    for (uint32_t i = 0; i < n; i++) {  // before any byte moves
        if (blk_off[i + 1] < blk_off[i]) { c->err = "blk_off not ascending"; return KC_ERR_BAD_ARG; }
        if (blk_off[i + 1] - blk_off[i] > KC_S2_MAX_BLOCK) { c->err = "S2 block larger than 1 GiB not served by the device path"; return KC_ERR_UNSUPPORTED; }
    }
    const uint64_t total = blk_off[n] - blk_off[0];
    const uint64_t ov_min = c->cfg.host_overlap_min_mib >= 0 ? (uint64_t)c->cfg.host_overlap_min_mib << 20 : (uint64_t)512 << 20;
    if (total >= ov_min && !c->cfg.host_serial && c->cfg.host_pipe_mib < 16 && level < KC_S2_LEVEL_BEST && c->cfg.host_roll) {
        RollEncFn enc = [level](kc_ctx* lane, const uint8_t* d_in, const uint64_t* rel, uint32_t nu, uint8_t* d_out, uint64_t cap, uint64_t* oo) {
            return kc_s2_encode_blocks_lvl_dev(lane, level, d_in, rel, nu, d_out, cap, oo);
        };
        auto mx = [](uint64_t len) { return (uint64_t)kc_s2_max_encoded_len((int64_t)len); };
        // sub-batches: halves (at most 1 GiB), not the engine's quarters — an S2 launch of 512 MiB is a quarter of a residency and four such
        // launches side by side run behind one of 2 GiB: C4 (2 GiB calls, four in flight) 40.9-41.9 GB/s with quarters, 45.2-45.3 with halves,
        // one call alone 25.5 / 25.3 (profiles/r06_ab_kernels.txt, session r8g)
        uint64_t sub = 0;
        if (c->cfg.host_roll_mib < 1) sub = std::min<uint64_t>((uint64_t)1 << 30, std::max<uint64_t>((total + 1) / 2, (uint64_t)64 << 20));
        const kc_status rs = host_rolling(c, src, blk_off, n, dst, dst_cap, out_off, enc, mx, sub);
        if (rs != KC_ERR_UNSUPPORTED || !c->err.empty()) return rs;  // UNSUPPORTED with no message: no engine on this device
    }
    if (total >= ov_min && total <= c->max_batch_bytes && !c->cfg.host_serial && c->cfg.host_pipe_mib < 16 && level < KC_S2_LEVEL_BEST) {  // (the best levels: 4.5 MiB of tables per block, several device batches)
        uint64_t need = 0;
        for (uint32_t i = 0; i < n; i++) need += ((uint64_t)kc_s2_max_encoded_len((int64_t)(blk_off[i + 1] - blk_off[i])) + 15) & ~(uint64_t)15;
        auto enq = [&](ChunkFeed& feed, const uint8_t* d_in, const uint64_t* rel, uint8_t* d_out) {
            return s2_encode_dev(c, d_in, rel, n, d_out, need, out_off, 0, 0, level, &feed);
        };
        auto region = [&](uint32_t u0) { return c->plan.stage_off[u0]; };
        auto fin = [&](bool* redo) {
            *redo = false;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipGetLastError());
            return KC_OK;
        };
        return host_chunk_fed(c, src, blk_off, n, dst, dst_cap, out_off, need, enq, region, fin);
    }
    if (total >= 2 * host_sub_bytes(c, total) && !c->cfg.host_serial) {
        auto enc = [&](const uint8_t* d_in, const uint64_t* rel, uint32_t nu, uint8_t* d_out, uint64_t cap, uint64_t* oo) {
            return kc_s2_encode_blocks_lvl_dev(c, level, d_in, rel, nu, d_out, cap, oo);
        };
        auto mx = [&](uint64_t len) { return (uint64_t)kc_s2_max_encoded_len((int64_t)len); };
        return host_pipeline(c, src, blk_off, n, dst, dst_cap, out_off, host_sub_bytes(c, total), enc, mx);
    }
    uint64_t need = 0;
    for (uint32_t i = 0; i < n; i++) need += ((uint64_t)kc_s2_max_encoded_len((int64_t)(blk_off[i + 1] - blk_off[i])) + 15) & ~(uint64_t)15;
    kc_status s;
    if ((s = ensure(c, c->tmp_src, total + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
    HIPCHK(c, hipMemcpyAsync(c->tmp_src.p, src + blk_off[0], total, hipMemcpyHostToDevice, c->stream));
    std::vector<uint64_t> rel(n + 1);
    for (uint32_t i = 0; i <= n; i++) rel[i] = blk_off[i] - blk_off[0];
    s = kc_s2_encode_blocks_lvl_dev(c, level, (const uint8_t*)c->tmp_src.p, rel.data(), n, (uint8_t*)c->tmp_dst.p, need, out_off);
    if (s != KC_OK) return s;
    const uint64_t outn = out_off[n];
    if (outn > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
    HIPCHK(c, hipMemcpyAsync(dst, c->tmp_dst.p, outn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KC_OK;
}

}  // extern "C"
