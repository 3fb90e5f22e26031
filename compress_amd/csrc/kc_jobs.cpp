// kc_jobs.cpp — WithConcurrentBlocks: one stream cut into jobs, each a device unit with its overlap prefix as history.
#include "kc_hostpipe.h"

// ---------------------------------------------------------------------------------------
// WithConcurrentBlocks (zstd/enc_jobs.go, encoder.go:214-247, 585-597, 652-700): ONE stream cut into jobs of
// max(4 * window, 512 KiB) input bytes, each encoded on a freshly reset encoder whose history is the last overlapSize bytes of
// the previous job's input (ResetPrefix), the job outputs concatenated behind one frame header.  The jobs are independent
// units for the device: unit k = [overlap prefix || job input] in a work buffer, its table primed from the prefix on the host
// exactly as ResetPrefix does (enc_fast.go:800-811, enc_dfast.go:1040-1050, enc_better.go:1099-1112).
// ---------------------------------------------------------------------------------------

namespace kci {

uint64_t xxh64_host(const uint8_t* p, size_t len) {  // zstd/internal/xxhash/xxhash.go:27-230, seed 0
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    auto rol = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { return rol(acc + in * P2, 31) * P1; };
    auto merge = [&](uint64_t acc, uint64_t v) { return (acc ^ round(0, v)) * P1 + P4; };
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        for (; p + 32 <= end; p += 32) { v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24)); }
        h = rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = P5;
    }
    h += (uint64_t)len;
    for (; p + 8 <= end; p += 8) { h ^= round(0, rd64(p)); h = rol(h, 27) * P1 + P4; }
    if (p + 4 <= end) { uint32_t v; memcpy(&v, p, 4); h ^= (uint64_t)v * P1; h = rol(h, 23) * P2 + P3; p += 4; }
    for (; p < end; p++) { h ^= (uint64_t)(*p) * P5; h = rol(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// One job's tables as ResetPrefix leaves them, in the device entry format ((position + 1) | tag << pos_bits, position counted
// from the start of the prefix).  `out` = the unit's table slot (match_table_bytes(level)), zeroed by the caller.
void build_prefix_tables(int level, const uint8_t* prefix, size_t n, int pos_bits, uint8_t* out) {
    const int TB = (32 - pos_bits) > 16 ? 16 : (32 - pos_bits);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](size_t pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << pos_bits); };
    auto ld = [&](size_t i) -> uint64_t { uint64_t v; memcpy(&v, prefix + i, 8); return v; };
    if (n < 8) return;
    const size_t end = n - 8;
    if (level == KC_SPEED_BETTER) {  // enc_better.go:1099-1112: i = 0, 2, ... : long table with its chain, short table one byte on
        uint32_t* ltab = (uint32_t*)out;  // pairs {offset, prev}
        uint32_t* stab = (uint32_t*)(out + ((size_t)8 << 19));
        for (size_t i = 0; i < end; i += 2) {
            const uint64_t cv = ld(i);
            const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 19));
            const uint32_t old = ltab[2 * h];
            ltab[2 * h] = mk(i, (uint32_t)cv);
            ltab[2 * h + 1] = old;
            const uint64_t v = cv >> 8;
            stab[(uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 13))] = mk(i + 1, (uint32_t)v);
        }
        return;
    }
    // fastEncoder.ResetPrefix (enc_fast.go:800-811): every 4th position from 1, 6-byte hash, 2^15 entries.  doubleFastEncoder embeds
    // it (enc_dfast.go:1040-1041): the same entries land in ITS short table, although its lookups hash 5 bytes — kept as it is.
    uint32_t* ftab = level == KC_SPEED_DEFAULT ? (uint32_t*)(out + ((size_t)4 << 17)) : (uint32_t*)out;
    for (size_t i = 1; i < end; i += 4) {
        const uint64_t cv = ld(i);
        ftab[(uint32_t)(((cv << 16) * 227718039650203ULL) >> (64 - 15))] = mk(i, (uint32_t)cv);
    }
    if (level == KC_SPEED_DEFAULT) {  // enc_dfast.go:1042-1050: every 2nd position from 1 into the long table
        uint32_t* ltab = (uint32_t*)out;
        for (size_t i = 1; i < end; i += 2) {
            const uint64_t cv = ld(i);
            ltab[(uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 17))] = mk(i, (uint32_t)cv);
        }
    }
}

}  // namespace kci

extern "C" {

int64_t kc_zstd_job_size(const kc_zstd_opts* o) {  // encoderOptions.jobSize, encoder_options.go:356-359
    return o ? std::max<int64_t>((int64_t)o->window_size * 4, (int64_t)512 << 10) : -1;
}
int64_t kc_zstd_overlap_size(const kc_zstd_opts* o) {  // encoderOptions.overlapSize, encoder_options.go:362-371
    if (!o) return -1;
    return o->level == KC_SPEED_BEST ? o->window_size / 2 : (o->level == KC_SPEED_BETTER ? o->window_size / 4 : o->window_size / 8);
}

kc_status kc_zstd_encode_jobs(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, uint64_t len, const uint64_t* cuts, uint64_t n_cuts,
                              uint8_t* dst, uint64_t dst_cap, uint64_t* out_len) {
    if (!c || !o || !out_len || (len && (!src || !dst)) || (n_cuts && !cuts)) return KC_ERR_BAD_ARG;
    if (c->job_active) return KC_ERR_BAD_ARG;  // a submitted call is still in flight on this context: kc_wait first (c->err belongs to its thread)
    c->err.clear();
    *out_len = 0;
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    if (o->dict != nullptr && o->dict_len > 0) {
        c->err = "the reference switches WithConcurrentBlocks off when a dictionary is set (zstd/encoder.go:81,174): use kc_zstd_encode_streams";
        return KC_ERR_UNSUPPORTED;
    }
    for (uint64_t i = 1; i < n_cuts; i++) if (cuts[i] < cuts[i - 1]) { c->err = "flush points not ascending"; return KC_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t jobSize = (uint64_t)kc_zstd_job_size(o), overlap = (uint64_t)kc_zstd_overlap_size(o);
    // jobs dispatched before Close: `filling` reached jobSize during Write (encoder.go:239-244), or a Flush found bytes in it (:587-591)
    std::vector<uint64_t> lo, hi;
    uint64_t pos = 0, ci = 0;
    for (;;) {
        while (ci < n_cuts && cuts[ci] <= pos) ci++;
        uint64_t e = pos + jobSize;
        bool dispatched = e <= len;
        if (ci < n_cuts && cuts[ci] < e && cuts[ci] <= len) { e = cuts[ci]; dispatched = true; }
        if (!dispatched) break;
        lo.push_back(pos);
        hi.push_back(e);
        pos = e;
    }
    const uint64_t tail = len - pos;  // what Close finds in `filling`
    if (lo.empty()) {  // dispatchJob(true) before any header was written (enc_jobs.go:263-289)
        if (tail > 0 && tail <= (uint64_t)o->block_size) {  // single block: the EncodeAll frame
            const uint64_t uo[2] = {0, len};
            uint64_t oo[2] = {0, 0};
            s = kc_zstd_encode_units(c, o, src, uo, 1, dst, dst_cap, oo);
            if (s == KC_OK) *out_len = oo[1];
            return s;
        }
        if (tail == 0 && !o->full_zero) return KC_OK;
    }
    lo.push_back(pos);
    hi.push_back(len);  // the final job (possibly empty)
    const uint32_t nj = (uint32_t)lo.size();
    // frame header (enc_jobs.go:291-304): no content size, window = the encoder's, not single segment, no dictionary id
    uint8_t hdr[8];
    int hl = 0;
    hdr[hl++] = 0x28; hdr[hl++] = 0xb5; hdr[hl++] = 0x2f; hdr[hl++] = 0xfd;
    hdr[hl++] = o->crc ? (uint8_t)(1 << 2) : (uint8_t)0;
    hdr[hl++] = (uint8_t)((bitsLen32((uint32_t)o->window_size - 1) - 10) << 3);
    // work buffer: unit k = [prefix_k || job_k]; prefix_k = the last min(overlap, len(job k-1)) bytes of job k-1 (enc_jobs.go:325-331)
    std::vector<uint32_t> jhist(nj), jflags(nj);
    std::vector<uint64_t> jlen(nj);
    for (uint32_t k = 0; k < nj; k++) {
        const uint64_t ov = k == 0 ? 0 : std::min<uint64_t>(overlap, hi[k - 1] - lo[k - 1]);
        jhist[k] = (uint32_t)ov;
        jflags[k] = k + 1 == nj ? 1u : 0u;
        jlen[k] = ov + (hi[k] - lo[k]);
        if (jlen[k] > KC_MAX_UNIT_BYTES) { c->err = "job larger than 1 GiB: not served by the device path"; return KC_ERR_UNSUPPORTED; }
    }
    const size_t tb = match_table_bytes(o->level);
    uint64_t crc = 0;
    std::thread crcT;
    if (o->crc) crcT = std::thread([&] { crc = xxh64_host(src, (size_t)len); });  // under the device work
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } joinCrc{crcT};
    struct Unhook { kc_ctx* c; ~Unhook() { c->job_hist = nullptr; c->job_flags = nullptr; c->job_tables = nullptr; c->job_primed = false; c->job_redo_list.clear(); } } unhook{c};
    if ((uint64_t)hl > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
    memcpy(dst, hdr, (size_t)hl);
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    c->last_batches = 0;
    // The jobs go to the device in batches bounded like kc_zstd_encode_units_dev's: by input bytes and by the scratch their
    // tables and per-block strides ask for (a long stream at a small window is thousands of jobs).  Each batch's tables are
    // primed for the batch only (on the device; with KC_OPT_JOB_PRIME 0 on the host, whose memory the same budget then bounds).
    uint64_t budget = scratch_budget(c);
    uint64_t done = 0;  // frame bytes behind the header so far
    std::vector<uint8_t> tabs;
    std::vector<uint64_t> boff, oo;
    uint32_t k0 = 0;
    for (int attempt = 0; k0 < nj;) {
        uint32_t k1 = k0;
        uint64_t scratch = 0, bytes = 0, need = 0, maxUnit = 16;
        while (k1 < nj) {
            const uint64_t us = zstd_unit_scratch(o, jlen[k1]);
            if (k1 > k0 && (bytes + jlen[k1] > c->max_batch_bytes || (scratch + us) + ((scratch + us) >> 3) > budget)) break;
            scratch += us;
            bytes += jlen[k1];
            need += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)jlen[k1]) + 15) & ~(uint64_t)15;
            maxUnit = std::max(maxUnit, jlen[k1]);
            k1++;
        }
        const uint32_t nb = k1 - k0;
        int pos_bits = 1;
        while (((uint64_t)1 << pos_bits) <= maxUnit + 2) pos_bits++;  // batch_begin's, for this batch
        const bool primeHost = tb != 0 && c->cfg.job_prime == 0;  // (default: kc_zstd_prime_kernel, from the prefix bytes staged below)
        if (primeHost) {
            try { tabs.assign((size_t)nb * tb, 0); } catch (...) { c->err = "host memory for the jobs' tables"; return KC_ERR_UNSUPPORTED; }
            const int T = std::max(1, std::min<int>(host_copy_threads(c), (int)nb));
            std::vector<std::thread> th;
            std::atomic<uint32_t> next{0};
            for (int t = 0; t < T; t++)
                th.emplace_back([&] {
                    for (uint32_t k = next++; k < nb; k = next++)
                        if (jhist[k0 + k]) build_prefix_tables(o->level, src + lo[k0 + k] - jhist[k0 + k], jhist[k0 + k], pos_bits, tabs.data() + (size_t)k * tb);
                });
            for (auto& x : th) x.join();
        }
        if ((s = ensure(c, c->tmp_src, bytes + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
        boff.assign(nb + 1, 0);
        for (uint32_t k = 0; k < nb; k++) {
            boff[k + 1] = boff[k] + jlen[k0 + k];
            if (jlen[k0 + k]) HIPCHK(c, hipMemcpyAsync((uint8_t*)c->tmp_src.p + boff[k], src + lo[k0 + k] - jhist[k0 + k], jlen[k0 + k], hipMemcpyHostToDevice, c->stream));
        }
        oo.assign(nb + 1, 0);
        uint64_t produced = 0;
        c->job_hist = jhist.data() + k0;
        c->job_flags = jflags.data() + k0;
        c->job_tables = primeHost ? tabs.data() : nullptr;
        c->job_primed = tb != 0;  // (SpeedBestCompression: the kernel indexes each job's prefix itself)
        c->oom = false;
        s = run_batch(c, o, (const uint8_t*)c->tmp_src.p, boff.data(), nb, (uint8_t*)c->tmp_dst.p, need, oo.data(), &produced);
        c->job_redo_list.clear();
        if (s == KC_ERR_UNSUPPORTED && c->oom && nb > 1 && attempt < 6) {
            attempt++;
            budget /= 2;  // another process took device memory since hipMemGetInfo: this batch again at half the size
            c->err.clear();
            continue;
        }
        if (s != KC_OK) return s;
        c->last_batches++;
        if ((uint64_t)hl + done + produced + (o->crc ? 4 : 0) > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
        if (produced) HIPCHK(c, hipMemcpyAsync(dst + hl + done, c->tmp_dst.p, produced, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (tmp_src / tmp_dst and the host tables are the next batch's)
        done += produced;
        k0 = k1;
        if (attempt) { attempt = 0; budget = scratch_budget(c); }  // the squeeze was this batch's: later batches start from what is free now
    }
    if (crcT.joinable()) crcT.join();
    const uint64_t total = (uint64_t)hl + done + (o->crc ? 4 : 0);
    if (o->crc) for (int k = 0; k < 4; k++) dst[hl + done + k] = (uint8_t)(crc >> (8 * k));
    *out_len = total;
    return KC_OK;
}

}  // extern "C"
