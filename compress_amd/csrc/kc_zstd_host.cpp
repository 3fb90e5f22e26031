// kc_zstd_host.cpp — zstd entry points over HOST buffers (what the cgo shim calls): units, streams, Flush cuts, the asynchronous
// submit / wait pair, and the diagnostics (checksums of units, the match finder's parse).
#include "kc_hostpipe.h"

extern "C" {

kc_status kc_zstd_encode_units(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                               uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off || !out_off || (n_units && (!src || !dst))) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_supported(c, o);  // before any byte moves: an unsupported request must not pay the PCIe copy
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    if (n_units == 0) { out_off[0] = 0; return KC_OK; }
    if ((s = validate_units(c, o, unit_off, n_units)) != KC_OK) return s;
    const uint64_t total = unit_off[n_units] - unit_off[0];
    const uint64_t ov_min = c->cfg.host_overlap_min_mib >= 0 ? (uint64_t)c->cfg.host_overlap_min_mib << 20 : (uint64_t)1 << 30;
    if (total >= ov_min && !c->cfg.host_serial && c->cfg.host_pipe_mib < 16 && c->cuts == nullptr && c->cfg.host_roll && c->job_hist == nullptr &&
        o->level != KC_SPEED_BEST) {  // (the best level: 34 MiB of persistent table slots per unit and context - not on four lanes at once)
        const kc_zstd_opts oc = *o;
        const int sm = c->stream_mode;
        RollEncFn enc = [oc, sm](kc_ctx* lane, const uint8_t* d_in, const uint64_t* rel, uint32_t nu, uint8_t* d_out, uint64_t cap, uint64_t* oo) {
            return sm ? kc_zstd_encode_streams_dev(lane, &oc, d_in, rel, nu, d_out, cap, oo) : kc_zstd_encode_units_dev(lane, &oc, d_in, rel, nu, d_out, cap, oo);
        };
        auto mx = [oc](uint64_t len) { return (uint64_t)kc_zstd_max_encoded_size(&oc, (int64_t)len); };
        // SpeedBetterCompression: a unit takes ~45 ms however few are resident (a chain of dependent table trips), so its sub-batches are
        // halves, not quarters: measured on C5 (1 GiB calls, four in flight) 15.2 GB/s with quarters, 18.4 with halves, 19.5 uncut; one
        // call alone 13.7 / 13.0 / 12.0 (profiles/r06_ab_kernels.txt, session r8c)
        uint64_t sub = 0;
        if (o->level == KC_SPEED_BETTER && c->cfg.host_roll_mib < 1)
            sub = std::min<uint64_t>((uint64_t)1 << 30, std::max<uint64_t>((total + 1) / 2, (uint64_t)64 << 20));
        s = host_rolling(c, src, unit_off, n_units, dst, dst_cap, out_off, enc, mx, sub);
        if (s != KC_ERR_UNSUPPORTED || !c->err.empty()) return s;  // UNSUPPORTED with no message: no engine on this device
    }
    if (total >= ov_min && !c->cfg.host_serial && c->cfg.host_pipe_mib < 16 && c->cuts == nullptr) {
        s = host_overlapped_zstd(c, o, src, unit_off, n_units, dst, dst_cap, out_off);
        if (s != KC_ERR_UNSUPPORTED || !c->err.empty()) return s;  // UNSUPPORTED with no message: shape not served by the one-batch path
    }
    const uint64_t sub = host_sub_bytes(c, total);
    if (total >= 2 * sub && !c->cfg.host_serial && c->cuts == nullptr) {  // (Flush points are indexed by unit: one batch loop)
        auto enc = [&](const uint8_t* d_in, const uint64_t* rel, uint32_t nu, uint8_t* d_out, uint64_t cap, uint64_t* oo) {
            return kc_zstd_encode_units_dev(c, o, d_in, rel, nu, d_out, cap, oo);
        };
        auto mx = [&](uint64_t len) { return (uint64_t)kc_zstd_max_encoded_size(o, (int64_t)len); };
        return host_pipeline(c, src, unit_off, n_units, dst, dst_cap, out_off, sub, enc, mx);
    }
    uint64_t need = 0;
    for (uint32_t i = 0; i < n_units; i++)
        need += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)(unit_off[i + 1] - unit_off[i])) + (c->cuts ? 3 * (c->cut_off[i + 1] - c->cut_off[i]) + 3 : 0) + 15) & ~(uint64_t)15;
    if ((s = ensure(c, c->tmp_src, total + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
    HIPCHK(c, hipMemcpyAsync(c->tmp_src.p, src + unit_off[0], total, hipMemcpyHostToDevice, c->stream));
    std::vector<uint64_t> rel(n_units + 1);
    for (uint32_t i = 0; i <= n_units; i++) rel[i] = unit_off[i] - unit_off[0];
    s = kc_zstd_encode_units_dev(c, o, (const uint8_t*)c->tmp_src.p, rel.data(), n_units, (uint8_t*)c->tmp_dst.p, need, out_off);
    if (s != KC_OK) return s;
    const uint64_t outn = out_off[n_units];
    if (outn > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
    HIPCHK(c, hipMemcpyAsync(dst, c->tmp_dst.p, outn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KC_OK;
}

kc_status kc_zstd_encode_streams(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                                 uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o) return KC_ERR_BAD_ARG;
    c->stream_mode = 1;
    const kc_status s = kc_zstd_encode_units(c, o, src, unit_off, n_units, dst, dst_cap, out_off);
    c->stream_mode = 0;
    return s;
}

// Streams with Flush points.  cut_off: n_units+1 indices into cuts; cuts[cut_off[i] .. cut_off[i+1]) = for stream i, ascending, the
// number of bytes that had been written when Flush was called.
static kc_status check_cuts(kc_ctx* c, const uint64_t* unit_off, uint32_t n_units, const uint64_t* cut_off, const uint64_t* cuts) {
    if (!cut_off || (cut_off[n_units] > cut_off[0] && !cuts)) return KC_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n_units; i++) {
        if (cut_off[i + 1] < cut_off[i]) { c->err = "cut_off not ascending"; return KC_ERR_BAD_ARG; }
        for (uint64_t k = cut_off[i]; k + 1 < cut_off[i + 1]; k++)
            if (cuts[k + 1] < cuts[k]) { c->err = "cuts of a stream not ascending"; return KC_ERR_BAD_ARG; }
    }
    (void)unit_off;
    return KC_OK;
}

// The block plan of one stream with Flush points, as the device path lays it out (host logic only; tests without a GPU).
int64_t kc_zstd_plan_stream_blocks(int32_t block_size, uint64_t len, const uint64_t* cuts, uint64_t n_cuts, uint32_t* starts, uint64_t starts_cap,
                                   uint32_t* flags) {
    if (block_size <= 0 || !flags || (n_cuts && !cuts)) return -1;
    std::vector<uint32_t> st;
    const uint32_t n = plan_stream_blocks((uint64_t)block_size, len, cuts, n_cuts, &st, flags);
    if (starts) { if (st.size() > starts_cap) return -2; for (size_t i = 0; i < st.size(); i++) starts[i] = st[i]; }
    return (int64_t)n;
}

kc_status kc_zstd_encode_streams_cuts_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                          const uint64_t* cut_off, const uint64_t* cuts, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_cuts(c, unit_off, n_units, cut_off, cuts);
    if (s != KC_OK) return s;
    static const uint64_t none = 0;
    c->cut_off = cut_off;
    c->cuts = cuts ? cuts : &none;
    c->cut_unit0 = 0;
    s = kc_zstd_encode_streams_dev(c, o, d_src, unit_off, n_units, d_dst, dst_cap, out_off);
    c->cuts = nullptr;
    c->cut_off = nullptr;
    return s;
}

kc_status kc_zstd_encode_streams_cuts(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                                      const uint64_t* cut_off, const uint64_t* cuts, uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_cuts(c, unit_off, n_units, cut_off, cuts);
    if (s != KC_OK) return s;
    static const uint64_t none = 0;
    c->cut_off = cut_off;
    c->cuts = cuts ? cuts : &none;
    c->cut_unit0 = 0;
    s = kc_zstd_encode_streams(c, o, src, unit_off, n_units, dst, dst_cap, out_off);
    c->cuts = nullptr;
    c->cut_off = nullptr;
    return s;
}

}  // extern "C"

namespace kci {
void host_pipe_free(void* h) { delete (HostPipe*)h; }
}  // namespace kci

extern "C" {

// Asynchronous form of the host-buffer entry points: submit returns at once, the call runs on a thread of its own (staging,
// kernels and drain of a batch are already overlapped inside one call; with two contexts a caller also overlaps consecutive
// batches: submit(A, batch k+1) while wait(B) drains batch k).  One job per context; every buffer, and the option struct's
// dictionary, must stay valid until kc_wait returns the job's status.
kc_status kc_zstd_encode_units_submit(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                                      uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o) return KC_ERR_BAD_ARG;
    if (c->job_active) return KC_ERR_BAD_ARG;  // a submitted job is still in flight: kc_wait first (c->err belongs to the job's thread)
    const kc_zstd_opts oc = *o;
    c->job_active = true;
    c->job = std::thread([=] { c->job_status = kc_zstd_encode_units(c, &oc, src, unit_off, n_units, dst, dst_cap, out_off); });
    return KC_OK;
}

kc_status kc_s2_encode_blocks_lvl_submit(kc_ctx* c, int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* dst,
                                         uint64_t dst_cap, uint64_t* out_off) {
    if (!c) return KC_ERR_BAD_ARG;
    if (c->job_active) return KC_ERR_BAD_ARG;  // a submitted job is still in flight: kc_wait first (c->err belongs to the job's thread)
    c->job_active = true;
    c->job = std::thread([=] { c->job_status = kc_s2_encode_blocks_lvl(c, level, src, blk_off, n, dst, dst_cap, out_off); });
    return KC_OK;
}

kc_status kc_wait(kc_ctx* c) {
    if (!c) return KC_ERR_BAD_ARG;
    if (!c->job_active) { c->err = "no submitted job on this context"; return KC_ERR_BAD_ARG; }
    c->job.join();
    c->job_active = false;
    return c->job_status;
}

kc_status kc_xxh64_units_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units, uint64_t* out_hash) {
    if (!c || !unit_off || !out_hash || (n_units && !d_src)) return KC_ERR_BAD_ARG;
    c->err.clear();
    HIPCHK(c, hipSetDevice(c->device));
    if (n_units == 0) return KC_OK;
    kc_status s;
    if ((s = ensure(c, c->unit_off, (n_units + 1) * 8)) || (s = ensure(c, c->xxh, (size_t)n_units * 8))) return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, unit_off, (n_units + 1) * 8, hipMemcpyHostToDevice, c->stream));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    kc_launch_xxh64(d_src, (const uint64_t*)c->unit_off.p, n_units, (uint64_t*)c->xxh.p, c->stream);
    HIPCHK(c, hipMemcpyAsync(out_hash, c->xxh.p, (size_t)n_units * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return KC_OK;
}

kc_status kc_zstd_debug_parse_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                  uint32_t* seqs, uint64_t seq_cap, uint64_t* blk_first_seq, uint32_t* blk_extra_lits, uint32_t* blk_flags,
                                  uint32_t blk_cap, uint32_t* n_blocks_out) {
    if (!c || !o || !unit_off || !seqs || !blk_first_seq || !blk_extra_lits || !n_blocks_out) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    const int bs = o->block_size;
    std::vector<uint32_t> blk0(n_units + 1);
    uint32_t nb = 0;
    for (uint32_t i = 0; i < n_units; i++) { blk0[i] = nb; nb += (uint32_t)((unit_off[i + 1] - unit_off[i] + bs - 1) / bs); }
    blk0[n_units] = nb;
    if (nb > blk_cap) return KC_ERR_DST_TOO_SMALL;
    const uint32_t seq_stride = (uint32_t)(bs / 4 + 8);
    if ((s = ensure(c, c->unit_off, (n_units + 1) * 8)) || (s = ensure(c, c->unit_blk0, (n_units + 1) * 4)) ||
        (s = ensure(c, c->seqs, (size_t)nb * seq_stride * 8)) || (s = ensure(c, c->meta, (size_t)nb * sizeof(KcBlkMeta))))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, unit_off, (n_units + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->unit_blk0.p, blk0.data(), (n_units + 1) * 4, hipMemcpyHostToDevice, c->stream));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    KcMatchParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.src = d_src;
    mp.src_end = d_src + unit_off[n_units];
    mp.unit_off = (const uint64_t*)c->unit_off.p;
    mp.unit_blk0 = (const uint32_t*)c->unit_blk0.p;
    mp.seqs = (uint64_t*)c->seqs.p;
    mp.meta = (KcBlkMeta*)c->meta.p;
    mp.seq_stride = seq_stride;
    mp.block_size = bs;
    mp.max_match_off = o->window_size;
    mp.spec_w0 = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : 1;
    mp.spec_grow = c->cfg.spec_grow >= 0 ? (int)c->cfg.spec_grow : 2;
    if (mp.spec_w0 < 1) mp.spec_w0 = 1;
    if (mp.spec_w0 > 8) mp.spec_w0 = 8;
    mp.hist0 = 0;
    mp.rep1 = 1;
    mp.rep2 = 4;
    mp.rep3 = 8;
    {
        uint64_t maxLen = 16;
        for (uint32_t i = 0; i < n_units; i++) maxLen = std::max<uint64_t>(maxLen, unit_off[i + 1] - unit_off[i]);
        c->plan.max_unit_bytes = maxLen;
        int pb = 1;
        while (((uint64_t)1 << pb) <= maxLen + 2) pb++;
        mp.pos_bits = pb;
    }
    if (o->dict != nullptr && o->dict_len > 0) { c->err = "debug parse does not take dictionaries"; return KC_ERR_UNSUPPORTED; }
    if ((s = launch_match(c, mp, unit_off, n_units, n_units, bs, c->stream, o->level)) != KC_OK) return s;
    std::vector<KcBlkMeta> meta(nb);
    HIPCHK(c, hipMemcpyAsync(meta.data(), c->meta.p, (size_t)nb * sizeof(KcBlkMeta), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    uint64_t total = 0;
    std::vector<uint64_t> packed;
    for (uint32_t b = 0; b < nb; b++) {
        blk_first_seq[b] = total;
        blk_extra_lits[b] = meta[b].extra_lits;
        if (blk_flags) blk_flags[b] = meta[b].flags;
        const uint32_t n = meta[b].nseq;
        if (total + n > seq_cap) return KC_ERR_DST_TOO_SMALL;
        packed.resize(n);
        if (n) HIPCHK(c, hipMemcpy(packed.data(), (const uint64_t*)c->seqs.p + (size_t)b * seq_stride, (size_t)n * 8, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < n; k++) {
            const uint64_t v = packed[k];
            seqs[3 * (total + k) + 0] = (uint32_t)(v >> 44);
            seqs[3 * (total + k) + 1] = (uint32_t)((v >> 24) & 0xFFFFF);
            seqs[3 * (total + k) + 2] = (uint32_t)(v & 0xFFFFFF);
        }
        total += n;
    }
    blk_first_seq[nb] = total;
    *n_blocks_out = nb;
    return KC_OK;
}

}  // extern "C"
