// kc_batch.cpp — the zstd device pipeline: table preparation, batch_begin / batch_end (match finder, entropy stage, speculation
// re-run, size scan, checksum, compaction), scratch budgeting, and the device-resident entry points.
#include "kc_host.h"

// ---------------------------------------------------------------------------------------
// zstd device pipeline
// ---------------------------------------------------------------------------------------
namespace kci {


// Host-side construction of the dictionary-primed tables of betterFastEncoderDict.Reset
// (zstd/enc_better.go:1114-1183) in the device entry format (position+1 | tag << pos_bits).
void build_better_dict_tables(const uint8_t* dict, size_t len, int pos_bits, uint8_t* out, int reserved_bits = 0) {
    const int TB = (32 - pos_bits - reserved_bits) > 16 ? 16 : (32 - pos_bits - reserved_bits);  // reserved: the epoch stamp's bits (kc_zstd_match_better.hip)
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](uint64_t pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << pos_bits); };
    auto ld64h = [&](size_t i) -> uint64_t { uint64_t v; memcpy(&v, dict + i, 8); return v; };
    uint32_t* ltab = (uint32_t*)out;  // pairs {offset, prev}
    uint32_t* stab = (uint32_t*)(out + ((size_t)8 << 19));
    if (len < 8) return;
    // short table: every position i, i+1, i+2, i+3 for i stepping by 4 while i < len-8 (:1126-1152)
    for (size_t i = 0; i + 8 < len; i += 4) {
        const uint64_t cv = ld64h(i);
        for (int k = 0; k < 4; k++) {
            const uint64_t v = cv >> (8 * k);
            const uint32_t h = (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 13));
            stab[h] = mk(i + k, (uint32_t)v);
        }
    }
    // long table: every position 0 .. len-9, chained (:1161-1183)
    for (size_t i = 0; i + 8 < len || i == 0; i++) {
        const uint64_t cv = ld64h(i);
        const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 19));
        const uint32_t old = ltab[2 * h];
        ltab[2 * h] = mk(i, (uint32_t)cv);
        ltab[2 * h + 1] = old;
        if (i + 8 >= len) break;
    }
}

// fastEncoderDict.Reset (zstd/enc_fast.go:813-845): 2^15 table, 6-byte hash, positions i and i+1 for i stepping by 2.
// Also the SHORT table of doubleFastEncoderDict, which embeds fastEncoderDict and keeps this priming although its
// lookups use the 5-byte hash (enc_dfast.go:1053-1056).
void build_fast_dict_table(const uint8_t* dict, size_t len, int pos_bits, uint32_t* tab) {
    const int TB = (32 - pos_bits) > 16 ? 16 : (32 - pos_bits);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](uint64_t pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << pos_bits); };
    if (len < 8) return;
    for (size_t i = 0; i + 8 < len; i += 2) {
        uint64_t cv;
        memcpy(&cv, dict + i, 8);
        const uint32_t h0 = (uint32_t)(((cv << 16) * 227718039650203ULL) >> (64 - 15));
        const uint32_t h1 = (uint32_t)((((cv >> 8) << 16) * 227718039650203ULL) >> (64 - 15));
        tab[h0] = mk(i, (uint32_t)cv);
        tab[h1] = mk(i + 1, (uint32_t)(cv >> 8));
    }
}
// doubleFastEncoderDict.Reset long table (zstd/enc_dfast.go:1060-1083): every position 0 .. len-9, 8-byte hash, 2^17.
void build_dfast_dict_long(const uint8_t* dict, size_t len, int pos_bits, uint32_t* ltab) {
    const int TB = (32 - pos_bits) > 16 ? 16 : (32 - pos_bits);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    if (len < 8) return;
    for (size_t i = 0; i == 0 || i + 8 < len; i++) {
        uint64_t cv;
        memcpy(&cv, dict + i, 8);
        const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 17));
        ltab[h] = ((uint32_t)i + 1u) | (tagOf((uint32_t)cv) << pos_bits);
        if (i + 8 >= len) break;
    }
}

kc_status check_supported(kc_ctx* c, const kc_zstd_opts* o) {
    if (o->level < KC_SPEED_FASTEST || o->level > KC_SPEED_BEST) { c->err = "unknown encoder level"; return KC_ERR_UNSUPPORTED; }
    // (the reference takes any dictionary below 2 GiB as history, zstd/dict.go:27, enc_base.go:160-198; here the dictionary is staged in
    // front of every unit of a batch, which the scratch budget accounts for — 64 MiB keeps positions inside the LDS kernel's field too)
    if (o->dict_len > ((uint64_t)64 << 20)) { c->err = "dictionary larger than 64 MiB not served by the device path"; return KC_ERR_UNSUPPORTED; }
    if (o->block_size < 1024 || o->block_size > kMaxCompressedBlockSize || o->window_size < kMinWindowSize) { c->err = "bad block/window size"; return KC_ERR_BAD_ARG; }
    return KC_OK;
}


// Match finders: sub-wave groups (8 lanes per unit), per-unit hash tables in an HBM arena that is zeroed (or primed from the
// dictionary tables) before every launch.
size_t match_table_bytes(int level) {
    if (level == KC_SPEED_BEST) return 0;  // persistent slots (ensure_best_slots), not per unit
    return level == KC_SPEED_BETTER ? kc_zbetter_table_bytes() : (level == KC_SPEED_DEFAULT ? kc_zdfast_table_bytes() : kc_zfast_table_bytes());
}

// SpeedFastest: which kernel serves a launch of n_launch units.  The LDS-table kernel (one wave per unit, a few ms per unit
// whatever the batch) wins while the units in flight cannot cover the HBM-table kernel's latency; the crossover is measured
// (profiles/r03_crossover_zfast.csv) and set by KC_OPT_ZFAST_LDS_MAX_UNITS; KC_OPT_MATCH_PATH forces a path.  Units (with their
// dictionary history) of 256 KiB and more only fit the HBM path's position field: the HBM kernel takes those, the LDS kernel the rest.
bool zfast_use_lds(const kc_ctx* c, const KcMatchParams& mp, uint32_t n_launch, int level) {
    (void)mp;
    if (level != KC_SPEED_FASTEST) return false;
    if (c->cfg.match_path == KC_PATH_HBM) return false;
    if (c->cfg.match_path == KC_PATH_LDS) return true;
    return (int64_t)n_launch <= c->cfg.zfast_lds_max_units;
}
// with the LDS path chosen: some unit of the batch does not fit its position field and goes through the HBM-table kernel
bool zfast_lds_needs_hbm(const kc_ctx* c, const KcMatchParams& mp) {
    return (uint64_t)mp.hist0 + c->plan.max_unit_bytes > KC_ZFAST_LDS_MAX_UNIT;
}

// per-unit tables of n_launch units: zeroed, or primed from the dictionary tables
// SpeedBestCompression: min(n_launch, KC_OPT_BEST_SLOTS) persistent table slots, zeroed when allocated; a unit starts from whatever
// the slot's earlier units left, past which its position space has moved (kc_zstd_match_best.hip)
kc_status ensure_best_slots(kc_ctx* c, uint32_t n_launch, hipStream_t st) {
    uint32_t want = (uint32_t)std::min<int64_t>((int64_t)n_launch, c->cfg.best_slots);
    if (want < 1) want = 1;
    if (!c->best_cost.p) {
        kc_status s = ensure(c, c->predef, kc_fse_predef_bytes());
        if (s != KC_OK) return s;
        if (!c->predef_ready) {
            kc_launch_fse_predef_init(c->predef.p, st);
            c->predef_ready = true;
        }
        if ((s = ensure(c, c->best_cost, 96 * 4)) != KC_OK) return s;
        kc_launch_zbest_cost(c->predef.p, (int32_t*)c->best_cost.p, st);
    }
    if (want <= c->best_n) return KC_OK;
    // grow in powers of two so that a sequence of growing batches re-allocates a few times at most
    uint32_t n = 1;
    while (n < want) n <<= 1;
    if ((int64_t)n > c->cfg.best_slots) n = (uint32_t)c->cfg.best_slots;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        const uint64_t room = (uint64_t)((double)(fr + c->best_tables.cap) * 0.8) / kc_zbest_table_bytes();
        if (room < 1) { c->err = "device memory exhausted: no room for one SpeedBestCompression table slot (34 MiB)"; c->oom = true; return KC_ERR_UNSUPPORTED; }
        if ((uint64_t)n > room) n = (uint32_t)room;
    } else (void)hipGetLastError();
    if (n <= c->best_n) return KC_OK;
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->best_tables.p) HIPCHK(c, hipFree(c->best_tables.p));
    c->best_tables.p = nullptr;
    c->best_tables.cap = 0;
    c->best_n = 0;
    {   // exactly n slots (ensure() rounds up by an eighth: 4 GiB at 1024 slots)
        const hipError_t e = hipMalloc(&c->best_tables.p, (size_t)n * kc_zbest_table_bytes());
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->best_tables.p = nullptr;
            c->err = "device memory exhausted (" + std::to_string(((size_t)n * kc_zbest_table_bytes()) >> 20) + " MiB of SpeedBestCompression tables wanted)";
            c->oom = true;
            return KC_ERR_UNSUPPORTED;
        }
        c->best_tables.cap = (size_t)n * kc_zbest_table_bytes();
    }
    kc_status s;
    if ((s = ensure(c, c->best_cur, (size_t)8192 * 4)) != KC_OK) return s;
    HIPCHK(c, hipMemsetAsync(c->best_tables.p, 0, (size_t)n * kc_zbest_table_bytes(), st));
    HIPCHK(c, hipMemsetAsync(c->best_cur.p, 0, (size_t)8192 * 4, st));
    c->best_n = n;
    return KC_OK;
}

// SpeedBetterCompression batches whose tables carry epoch stamps instead of being cleared per launch: not the jobs of a
// WithConcurrentBlocks stream (their tables are primed per unit on the host), only while the stamp fits above position and tag, and
// — measured — not with a dictionary: there every lookup has to read the shared dictionary table beside the unit's own bucket, and
// on C5 that costs more (match finder 52.9 -> 63.7 ms per GiB) than the 7.7 ms of copying the dictionary tables it saves
// (KC_OPT_BETTER_DICT_EPOCH = 1 turns it on for measurements).
bool better_epoch_mode(const kc_ctx* c, int level, int pos_bits, int hist0) {
    return level == KC_SPEED_BETTER && !c->job_primed && c->job_hist == nullptr && pos_bits <= 22 && (hist0 == 0 || c->cfg.better_dict_epoch != 0);
}

kc_status prepare_tables(kc_ctx* c, const KcMatchParams& mp, uint32_t n_launch, hipStream_t st, int level) {
    if (level == KC_SPEED_BEST) return ensure_best_slots(c, n_launch, st);
    { const kc_status sc = tables_claim(c, st); if (sc != KC_OK) return sc; }
    c->better_epoch_now = 0;
    if (better_epoch_mode(c, level, mp.pos_bits, mp.hist0)) {
        const size_t tbb = match_table_bytes(level);
        kc_status se = ensure(c, c->tables, (size_t)n_launch * tbb);
        if (se != KC_OK) return se;
        const bool fresh = c->tab_owner != 1 || c->tab_pb != mp.pos_bits || n_launch > c->tab_units || c->tab_ep >= 15u || c->tab_ptr != c->tables.p;
        if (fresh) {
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tbb, st));
            c->tab_owner = 1;
            c->tab_pb = mp.pos_bits;
            c->tab_units = n_launch;
            c->tab_ep = 1;
            c->tab_ptr = c->tables.p;
        } else {
            c->tab_ep++;
        }
        c->better_epoch_now = c->tab_ep;
        return KC_OK;  // no dictionary copy either: the kernel reads the shared dictionary tables for buckets it has not written
    }
    if (zfast_use_lds(c, mp, n_launch, level) && !zfast_lds_needs_hbm(c, mp) && !c->job_primed) return KC_OK;  // the tables live in LDS (the arena is not touched)
    const bool fastEpoch = level == KC_SPEED_FASTEST && c->cfg.zfast_epoch != 0 && mp.hist0 == 0 && !c->job_primed && mp.pos_bits + KC_ZF_EPOCH_BITS + 4 <= 32;
    if (!fastEpoch) c->tab_owner = 0;  // (whatever follows rewrites the arena)
    c->fast_epoch_now = 0;
    const size_t tb = match_table_bytes(level);
    if (c->job_primed) {  // jobs: slot i holds the table ResetPrefix leaves from the prefix of unit i (or, in a re-run, of unit list[i])
        kc_status sj = ensure(c, c->tables, (size_t)n_launch * tb);
        if (sj != KC_OK) return sj;
        if (c->job_tables == nullptr) {  // primed here, from the prefix bytes already in the source buffer
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tb, st));
            KcPrimeParams pp;
            pp.src = mp.src;
            pp.unit_off = mp.unit_off;
            pp.unit_hist = mp.unit_hist;
            pp.unit_list = mp.unit_list;
            pp.unit_base = mp.unit_base;
            pp.n_launch = n_launch;
            pp.level = level;
            pp.pos_bits = mp.pos_bits;
            pp.tables = (uint8_t*)c->tables.p;
            pp.table_bytes = tb;
            kc_launch_zstd_prime(pp, st);
            return KC_OK;
        }
        if (mp.unit_list == nullptr) {
            HIPCHK(c, hipMemcpyAsync(c->tables.p, c->job_tables, (size_t)n_launch * tb, hipMemcpyHostToDevice, st));
            return KC_OK;
        }
        if (c->job_redo_list.size() < n_launch) { c->err = "job re-run without its unit list"; return KC_ERR_INTERNAL; }
        for (uint32_t i = 0; i < n_launch; i++)
            HIPCHK(c, hipMemcpyAsync((uint8_t*)c->tables.p + (size_t)i * tb, c->job_tables + (size_t)c->job_redo_list[i] * tb, tb, hipMemcpyHostToDevice, st));
        return KC_OK;
    }
    kc_status s = ensure(c, c->tables, (size_t)n_launch * tb);
    if (s != KC_OK) return s;
    if (mp.hist0 > 0) kc_launch_bcast((const uint8_t*)c->proto.p, (uint8_t*)c->tables.p, tb, n_launch, st);
    else if (fastEpoch) {
        // SpeedFastest, no dictionary: the entries carry this launch's stamp (kc_zstd_match.hip), so what earlier launches left in
        // the slots reads as empty and nothing is cleared (4 GiB of stores per 4 GiB batch: 0.65 ms) until the stamp wraps
        const bool fresh = c->tab_owner != 2 || c->tab_pb != mp.pos_bits || n_launch > c->tab_units || c->tab_ep >= (1u << KC_ZF_EPOCH_BITS) - 1u || c->tab_ptr != c->tables.p;
        if (fresh) {
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tb, st));
            c->tab_owner = 2;
            c->tab_pb = mp.pos_bits;
            c->tab_units = n_launch;
            c->tab_ep = 1;
            c->tab_ptr = c->tables.p;
        } else {
            c->tab_ep++;
        }
        c->fast_epoch_now = c->tab_ep;
    }
    else HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tb, st));
    return KC_OK;
}

// the match finder over n_launch units whose tables start at table slot `slot0` (unit = mp.unit_base + i or mp.unit_list[i])
void launch_match_kernel(kc_ctx* c, const KcMatchParams& mp, uint32_t slot0, uint32_t n_launch, hipStream_t st, int level, bool lds = false) {
    if (lds) {
        KcMatchParams ml = mp;
        ml.spec_w0 = (int32_t)c->cfg.lds_spec_w0;
        ml.lds_any_big = c->plan.max_unit_bytes > (uint64_t)131072 ? 1 : 0;
        if (c->job_primed)  // jobs: slot i of the arena holds the table primed from unit i's prefix (prepare_tables)
            kc_launch_zfast_match_lds(ml, (const uint32_t*)((uint8_t*)c->tables.p + (size_t)slot0 * match_table_bytes(level)), 1u << 15, n_launch, st);
        else
            kc_launch_zfast_match_lds(ml, mp.hist0 > 0 ? (const uint32_t*)c->proto.p : nullptr, 0u, n_launch, st);
        c->last_path = KC_PATH_LDS;
        if (zfast_lds_needs_hbm(c, mp)) {  // the units beyond the LDS kernel's position field
            ml = mp;
            ml.lds_split = 1;
            ml.epoch = c->fast_epoch_now;
            ml.xseg_k = (int32_t)c->cfg.zfast_xseg_k;
            ml.empty_filter = c->cfg.zfast_filter != 0;
            ml.tuned = c->cfg.zfast_variant < 0 ? (c->last_incompressible ? 1 : 0) : (int32_t)c->cfg.zfast_variant;
            kc_launch_zfast_match_grp(ml, (uint32_t*)((uint8_t*)c->tables.p + (size_t)slot0 * match_table_bytes(level)), n_launch, st);
        }
        return;
    }
    c->last_path = KC_PATH_HBM;
    if (level == KC_SPEED_BEST) {
        kc_launch_zbest_match(mp, (uint64_t*)c->best_tables.p, (uint32_t*)c->best_cur.p, (const int32_t*)c->best_cost.p, n_launch, c->best_n, st);
        return;
    }
    uint8_t* tab = (uint8_t*)c->tables.p + (size_t)slot0 * match_table_bytes(level);
    if (level == KC_SPEED_BETTER) {
        KcMatchParams mb = mp;
        mb.epoch = c->better_epoch_now;
        mb.proto = (mb.epoch != 0u && mp.hist0 > 0) ? (const uint8_t*)c->proto.p : nullptr;
        kc_launch_zbetter_match_grp(mb, tab, n_launch, mp.hist0 > 0, st);
    }
    else if (level == KC_SPEED_DEFAULT) kc_launch_zdfast_match_grp(mp, (uint32_t*)tab, n_launch, st);
    else {
        KcMatchParams mf = mp;
        mf.epoch = c->fast_epoch_now;
        mf.xseg_k = (int32_t)c->cfg.zfast_xseg_k;
        mf.empty_filter = c->cfg.zfast_filter != 0;
        mf.tuned = c->cfg.zfast_variant < 0 ? (c->last_incompressible ? 1 : 0) : (int32_t)c->cfg.zfast_variant;
        kc_launch_zfast_match_grp(mf, (uint32_t*)tab, n_launch, st);
    }
}

kc_status launch_match(kc_ctx* c, const KcMatchParams& mp, const uint64_t* unit_off, uint32_t n_units, uint32_t n_launch, int bs, hipStream_t st, int level) {
    (void)unit_off; (void)n_units; (void)bs;
    kc_status s = prepare_tables(c, mp, n_launch, st, level);
    if (s != KC_OK) return s;
    HIPCHK(c, hipEventRecord(c->ev[7], st));
    c->ev7_valid = true;
    launch_match_kernel(c, mp, 0, n_launch, st, level, zfast_use_lds(c, mp, n_launch, level));
    return KC_OK;
}

// Blocks of one stream with Flush points (zstd/encoder.go): writeBlocks cuts a block every blockSize bytes after the last Flush
// (:226-253); a Flush that finds nothing buffered does nothing (:552).  Close: a single block still buffered with no header
// written yet is the EncodeAll frame (:272-288), otherwise the stream frame, with an empty last block when nothing is buffered
// (:315-329).  Appends the block starts to *starts; *flags: bit 0 stream frame, bit 1 empty last block.  Returns the block count.
uint32_t plan_stream_blocks(uint64_t bs, uint64_t len, const uint64_t* cuts, uint64_t n_cuts, std::vector<uint32_t>* starts, uint32_t* flags) {
    uint64_t pos = 0, ci = 0, lastStart = 0;
    uint32_t ub = 0;
    while (pos < len) {
        uint64_t e = pos + bs;
        while (ci < n_cuts && cuts[ci] <= pos) ci++;
        if (ci < n_cuts && cuts[ci] < e) e = cuts[ci];
        if (e > len) e = len;
        if (starts) starts->push_back((uint32_t)pos);
        lastStart = pos;
        pos = e;
        ub++;
    }
    const bool flushedAtEnd = n_cuts > 0 && cuts[n_cuts - 1] >= len;
    const bool tailBuffered = ub > 0 && !flushedAtEnd && (len - lastStart) < bs;
    const bool streamU = len > 0 && !(ub == 1 && tailBuffered);
    *flags = (streamU ? 1u : 0u) | ((streamU && !tailBuffered) ? 2u : 0u);
    return ub;
}

kc_status batch_begin(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src_base, const uint64_t* unit_off, uint32_t n_units,
                      uint8_t* d_dst, uint64_t dst_cap, ChunkFeed* feed) {
    hipStream_t st = c->stream;
    if (c->pend) { c->err = "a batch is already in flight on this context"; return KC_ERR_BAD_ARG; }
    const int bs = o->block_size;
    Plan& pl = c->plan;
    pl.n_units = n_units;
    pl.blk0.resize(n_units + 1);
    pl.stage_off.resize(n_units + 1);
    pl.rel_off.resize(n_units + 1);
    uint64_t so = 0;
    uint32_t nb = 0;
    const bool irregular = c->cuts != nullptr;
    pl.blk_start.clear();
    pl.unit_flags.clear();
    for (uint32_t i = 0; i < n_units; i++) {
        const uint64_t len = unit_off[i + 1] - unit_off[i];
        pl.blk0[i] = nb;
        pl.stage_off[i] = so;
        pl.rel_off[i] = unit_off[i] - unit_off[0];
        uint32_t ub = (uint32_t)((len - (c->job_hist ? c->job_hist[i] : 0) + bs - 1) / bs);  // (a job's overlap prefix is history, not blocks)
        if (irregular) {
            const uint64_t* cp = c->cuts + c->cut_off[c->cut_unit0 + i];
            const uint64_t nc = c->cut_off[c->cut_unit0 + i + 1] - c->cut_off[c->cut_unit0 + i];
            uint32_t fl = 0;
            ub = plan_stream_blocks((uint64_t)bs, len, cp, nc, &pl.blk_start, &fl);
            pl.unit_flags.push_back(fl);
        }
        nb += ub;
        // every block costs a 3-byte header: Flush points add blocks that MaxEncodedSize(len) does not count
        so += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)len) + (irregular ? 3ull * (c->cut_off[c->cut_unit0 + i + 1] - c->cut_off[c->cut_unit0 + i]) + 3ull : 0ull) + 15) & ~(uint64_t)15;
    }
    pl.blk0[n_units] = nb;
    pl.stage_off[n_units] = so;
    pl.rel_off[n_units] = unit_off[n_units] - unit_off[0];
    pl.n_blocks = nb;
    pl.seq_stride = (uint32_t)(bs / 4 + 8);
    pl.lit_stride = (uint32_t)(bs + 64);
    if (so > dst_cap) { c->err = "dst_cap smaller than the sum of MaxEncodedSize(unit)"; return KC_ERR_DST_TOO_SMALL; }

    kc_status s;
    if ((s = ensure(c, c->unit_off, (n_units + 1) * 8)) || (s = ensure(c, c->unit_blk0, (n_units + 1) * 4)) ||
        (s = ensure(c, c->stage_off, (n_units + 1) * 8)) || (s = ensure(c, c->out_off, (n_units + 1 + (feed ? feed->cut.size() : 0)) * 8)) ||
        (s = ensure(c, c->seqs, (size_t)nb * pl.seq_stride * 8)) || (s = ensure(c, c->aux, (size_t)nb * pl.seq_stride * 8)) ||
        (s = ensure(c, c->lits, (size_t)nb * pl.lit_stride)) || (s = ensure(c, c->meta, (size_t)nb * sizeof(KcBlkMeta))) ||
        (s = ensure(c, c->stage, so + 64)) || (s = ensure(c, c->out_size, (size_t)n_units * 4)) ||
        (s = ensure(c, c->xxh, (size_t)n_units * 8)) || (s = ensure(c, c->redo, (size_t)n_units * 4)) ||
        (s = ensure(c, c->redo_blk, (size_t)nb + 1)) || (s = ensure(c, c->pop_blk, (size_t)nb + 1)) || (s = ensure(c, c->unit_list, (size_t)n_units * 4)) ||
        (s = ensure(c, c->predef, kc_fse_predef_bytes())) || (s = ensure(c, c->errflag, 64)))
        return s;
    if (!c->predef_ready) {
        kc_launch_fse_predef_init(c->predef.p, st);
        c->predef_ready = true;
    }
    if (irregular) {
        if ((s = ensure(c, c->blk_start, ((size_t)nb + 1) * 4)) || (s = ensure(c, c->unit_flags, (size_t)n_units * 4))) return s;
        if (nb) HIPCHK(c, hipMemcpyAsync(c->blk_start.p, pl.blk_start.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->unit_flags.p, pl.unit_flags.data(), (size_t)n_units * 4, hipMemcpyHostToDevice, st));
    }
    {   // the layout arrays: batches of equal-sized units repeat them exactly (every step of a fixed-size workload), and a copy from
        // pageable memory stalls the host for tens of microseconds each — upload only what differs from what the device holds
        auto same = [](const std::vector<uint8_t>& held, const void* p, size_t n, const void* dev, const void* held_dev) {
            return dev == held_dev && held.size() == n && memcmp(held.data(), p, n) == 0;
        };
        auto keep = [](std::vector<uint8_t>& held, const void* p, size_t n) { held.assign((const uint8_t*)p, (const uint8_t*)p + n); };
        const size_t n8 = (size_t)(n_units + 1) * 8, n4 = (size_t)(n_units + 1) * 4;
        if (!same(c->up_unit_off, pl.rel_off.data(), n8, c->unit_off.p, c->up_ptr[0])) {
            HIPCHK(c, hipMemcpyAsync(c->unit_off.p, pl.rel_off.data(), n8, hipMemcpyHostToDevice, st));
            keep(c->up_unit_off, pl.rel_off.data(), n8);
            c->up_ptr[0] = c->unit_off.p;
        }
        if (!same(c->up_blk0, pl.blk0.data(), n4, c->unit_blk0.p, c->up_ptr[1])) {
            HIPCHK(c, hipMemcpyAsync(c->unit_blk0.p, pl.blk0.data(), n4, hipMemcpyHostToDevice, st));
            keep(c->up_blk0, pl.blk0.data(), n4);
            c->up_ptr[1] = c->unit_blk0.p;
        }
        if (!same(c->up_stage_off, pl.stage_off.data(), n8, c->stage_off.p, c->up_ptr[2])) {
            HIPCHK(c, hipMemcpyAsync(c->stage_off.p, pl.stage_off.data(), n8, hipMemcpyHostToDevice, st));
            keep(c->up_stage_off, pl.stage_off.data(), n8);
            c->up_ptr[2] = c->stage_off.p;
        }
    }
    // (the flag arrays of the batch are zeroed by one launch further down: kc_launch_clear)

    const uint8_t* d_src = d_src_base + unit_off[0];
    // ---- dictionary (raw content, WithEncoderDictRaw): history = dict || unit (enc_base.go:189-198) ----
    const bool useDict = o->dict != nullptr && o->dict_len > 0;
    const int hist0 = useDict ? (int)o->dict_len : 0;
    uint64_t maxLen = 16;
    for (uint32_t i = 0; i < n_units; i++) maxLen = std::max<uint64_t>(maxLen, unit_off[i + 1] - unit_off[i]);
    pl.max_unit_bytes = maxLen;
    int pos_bits = 1;
    while (((uint64_t)1 << pos_bits) <= (uint64_t)hist0 + maxLen + 2) pos_bits++;
    // the packed sequences keep offset + 3 in 24 bits: fine for the default windows (4 / 8 MiB) and for any unit below 16 MiB
    if (std::min<uint64_t>((uint64_t)o->window_size, (uint64_t)hist0 + maxLen) + 3 > 0xFFFFFFull) {
        c->err = "window above 8 MiB with units above 16 MiB: offsets beyond the device path's 24-bit sequence field";
        return KC_ERR_UNSUPPORTED;
    }
    const uint8_t* k_src = d_src;               // what the match finder / entropy kernels read
    const uint64_t* k_off = (const uint64_t*)c->unit_off.p;
    if (useDict) {
        std::vector<uint64_t> woff(n_units + 1);
        for (uint32_t i = 0; i <= n_units; i++) woff[i] = pl.rel_off[i] + (uint64_t)i * (uint64_t)hist0;
        if ((s = ensure(c, c->work, woff[n_units] + 64)) || (s = ensure(c, c->work_off, (n_units + 1) * 8)) ||
            (s = ensure(c, c->dictbuf, (size_t)hist0 + 64)) || (s = ensure(c, c->proto, kc_zbetter_table_bytes())))
            return s;
        HIPCHK(c, hipMemcpyAsync(c->work_off.p, woff.data(), (n_units + 1) * 8, hipMemcpyHostToDevice, st));
        // The dictionary's content and its pristine tables (betterFastEncoderDict.Reset, enc_better.go:1114-1183, in the device entry
        // format) are the same from batch to batch: rebuilt and uploaded only when the dictionary, the level, the position width or the
        // stamp mode changed (round 6: the rolling host pipeline runs a batch per 256 MiB — 4 MiB of host table building, two uploads
        // and a stream synchronisation per batch before).
        const bool epochMode = better_epoch_mode(c, o->level, pos_bits, hist0);
        uint64_t key = 0xcbf29ce484222325ULL;
        for (int i = 0; i < hist0; i++) key = (key ^ o->dict[i]) * 0x100000001b3ULL;
        key ^= ((uint64_t)(uint32_t)hist0 << 32) ^ ((uint64_t)o->level << 8) ^ ((uint64_t)pos_bits << 16) ^ (epochMode ? (uint64_t)1 << 24 : 0);
        if (key == 0) key = 1;
        const size_t tbz = kc_zbetter_table_bytes();
        if (c->proto_key != key || c->proto_ptr != c->proto.p || c->dictbuf_ptr != c->dictbuf.p) {
            c->proto_key = 0;
            HIPCHK(c, hipMemcpyAsync(c->dictbuf.p, o->dict, (size_t)hist0, hipMemcpyHostToDevice, st));
            std::vector<uint8_t> proto(tbz, 0);
            if (o->level == KC_SPEED_BEST) { /* the kernel indexes the dictionary itself, per unit (bestFastEncoder.Reset) */ }
            else if (o->level == KC_SPEED_BETTER) {
                build_better_dict_tables(o->dict, (size_t)hist0, pos_bits, proto.data(), epochMode ? 4 : 0);
            } else if (o->level == KC_SPEED_DEFAULT) {
                build_dfast_dict_long(o->dict, (size_t)hist0, pos_bits, (uint32_t*)proto.data());
                build_fast_dict_table(o->dict, (size_t)hist0, pos_bits, (uint32_t*)(proto.data() + ((size_t)4 << 17)));
            } else build_fast_dict_table(o->dict, (size_t)hist0, pos_bits, (uint32_t*)proto.data());
            HIPCHK(c, hipMemcpyAsync(c->proto.p, proto.data(), proto.size(), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));  // proto and woff are locals
            c->proto_key = key;
            c->proto_ptr = c->proto.p;
            c->dictbuf_ptr = c->dictbuf.p;
        } else {
            HIPCHK(c, hipStreamSynchronize(st));  // woff is a local
        }
        kc_launch_prefix_units(d_src, (const uint64_t*)c->unit_off.p, (const uint64_t*)c->work_off.p, (const uint8_t*)c->dictbuf.p,
                               (uint32_t)hist0, (uint8_t*)c->work.p, n_units, st);
        k_src = (const uint8_t*)c->work.p;
        k_off = (const uint64_t*)c->work_off.p;
    }
    if (c->job_hist) {
        if (useDict) { c->err = "jobs of a WithConcurrentBlocks stream take no dictionary"; return KC_ERR_INTERNAL; }
        if ((s = ensure(c, c->d_job_hist, (size_t)n_units * 4)) || (s = ensure(c, c->d_job_flags, (size_t)n_units * 4))) return s;
        HIPCHK(c, hipMemcpyAsync(c->d_job_hist.p, c->job_hist, (size_t)n_units * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_job_flags.p, c->job_flags, (size_t)n_units * 4, hipMemcpyHostToDevice, st));
    }
    KcMatchParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.unit_hist = c->job_hist ? (const uint32_t*)c->d_job_hist.p : nullptr;
    mp.job_flags = c->job_hist ? (const uint32_t*)c->d_job_flags.p : nullptr;
    mp.src = k_src;
    mp.src_end = k_src + (useDict ? pl.rel_off[n_units] + (uint64_t)n_units * (uint64_t)hist0 : pl.rel_off[n_units]);
    mp.unit_off = k_off;
    mp.hist0 = hist0;
    mp.pos_bits = pos_bits;
    mp.stream_mode = c->stream_mode;
    mp.rep1 = (int32_t)o->dict_offsets[0];
    mp.rep2 = (int32_t)o->dict_offsets[1];
    mp.rep3 = (int32_t)o->dict_offsets[2];
    if (mp.rep1 <= 0 || mp.rep2 <= 0 || mp.rep3 <= 0) { mp.rep1 = 1; mp.rep2 = 4; mp.rep3 = 8; }  // opts not initialised through kc_zstd_opts_default
    mp.unit_blk0 = (const uint32_t*)c->unit_blk0.p;
    mp.seqs = (uint64_t*)c->seqs.p;
    mp.meta = (KcBlkMeta*)c->meta.p;
    mp.pop_blk = nullptr;
    mp.unit_list = nullptr;
    mp.blk_start = irregular ? (const uint32_t*)c->blk_start.p : nullptr;
    mp.unit_flags = irregular ? (const uint32_t*)c->unit_flags.p : nullptr;
    mp.seq_stride = pl.seq_stride;
    mp.block_size = bs;
    mp.max_match_off = o->window_size;
    // SpeedDefault, round 2 (ms per launch): at 4 GiB (32768 units: DRAM-transaction bound, wasted probes cost) width 2 / 3 / 4 then doubling
    // 265.9 / 266.7 / 274.0, 2 then +1 264.1; at 2 GiB (latency bound) 163.6 / - / 155.4, fixed 1: 285.5.  The BASELINE size is 4 GiB.
    // SpeedBetterCompression (1 GiB = 8192 units: latency-bound, wide speculation pays): width 1 / 2 / 4 then doubling 98.6 / 91.5 / 86.0,
    // fixed 4: 101.4, fixed 8: 80.4 ms with 8 lanes per unit; 16 lanes per unit, fixed 16: 61.1 ms (one probe per round, round 1: 181 ms)
    mp.spec_w0 = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : (o->level == KC_SPEED_DEFAULT ? 2 : (o->level == KC_SPEED_BETTER ? 16 : 1));
    // measured on C2 (ms per 4 GiB): width 1 then +1 per miss 137, fixed 2 136.5, 1 then doubling 140, fixed 1 167, fixed 4 157
    mp.spec_grow = c->cfg.spec_grow >= 0 ? (int)c->cfg.spec_grow : (o->level == KC_SPEED_FASTEST ? 1 : (o->level == KC_SPEED_BETTER ? 0 : 2));
    if (mp.spec_w0 < 1) mp.spec_w0 = 1;
    if (mp.spec_w0 > 64) mp.spec_w0 = 64;  // the kernels clamp to their group size

    KcEntropyParams ep;
    memset(&ep, 0, sizeof(ep));
    ep.unit_hist = mp.unit_hist;
    ep.job_flags = mp.job_flags;
    ep.src = k_src;
    ep.unit_off = k_off;
    ep.hist0 = hist0;
    ep.stream_mode = c->stream_mode;
    ep.stream_sync = o->concurrent == 1;
    ep.dict_huf = nullptr;
    ep.dict_huf_len = 0;
    ep.dict_huf_log = 0;
    if (o->dict_huf_len > 0) {  // dictionary literal table -> prevTable of every unit's first block
        uint8_t blobh[768];
        memcpy(blobh, o->dict_huf_val, 512);
        memcpy(blobh + 512, o->dict_huf_nbits, 256);
        if ((s = ensure(c, c->dicthuf, 768))) return s;
        HIPCHK(c, hipMemcpyAsync(c->dicthuf.p, blobh, 768, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));  // blobh is a stack buffer
        ep.dict_huf = (const uint8_t*)c->dicthuf.p;
        ep.dict_huf_len = o->dict_huf_len;
        ep.dict_huf_log = o->dict_huf_log;
    }
    ep.unit_blk0 = mp.unit_blk0;
    ep.seqs = mp.seqs;
    ep.meta = mp.meta;
    ep.lits = (uint8_t*)c->lits.p;
    ep.aux = (uint64_t*)c->aux.p;
    ep.stage = (uint8_t*)c->stage.p;
    ep.stage_off = (const uint64_t*)c->stage_off.p;
    ep.out_size = (uint32_t*)c->out_size.p;
    ep.xxh = (const uint64_t*)c->xxh.p;
    ep.redo_mask = (uint32_t*)c->redo.p;
    ep.redo_blk = (uint8_t*)c->redo_blk.p;
    ep.blk_start = mp.blk_start;
    ep.unit_flags = mp.unit_flags;
    ep.unit_list = nullptr;
    ep.predef = c->predef.p;
    ep.seq_stride = pl.seq_stride;
    ep.lit_stride = pl.lit_stride;
    ep.block_size = bs;
    ep.window_size = o->window_size;
    ep.crc = o->crc;
    ep.single = o->single;
    ep.no_entropy = o->no_entropy;
    ep.all_lit_entropy = o->all_lit_entropy;
    ep.full_zero = o->full_zero;
    ep.dict_id = o->dict_id;
    ep.err_flag = (uint32_t*)c->errflag.p;
    // raw blocks are copied once, by the compaction, from the source (KcRawDef): the entries of blocks that are not raw stay zero
    if ((s = ensure(c, c->rawdef, ((size_t)nb + 1) * sizeof(KcRawDef))) != KC_OK) return s;
    ep.rawdef = (KcRawDef*)c->rawdef.p;
    // The checksum moves behind the entropy stage (kc_xxh64_fin_kernel) where the batch has a regular block grid and no history in
    // front of its units: frames that turn out to be raw blocks only then get their payload copied by the pass that hashes it.
    const bool fuse_xxh = c->cfg.fuse_raw_xxh != 0 && o->crc && !c->job_hist && !useDict && hist0 == 0 && !irregular && (bs % 256) == 0 && feed == nullptr;
    ep.unit_raw = nullptr;
    if (fuse_xxh) {
        if ((s = ensure(c, c->unit_raw, (size_t)n_units * 4 + 4)) != KC_OK) return s;
        ep.unit_raw = (uint32_t*)c->unit_raw.p;
        ep.xxh = nullptr;
    }
    {   // the batch's flag arrays, zeroed by one launch (every DevBuf has at least 256 bytes of slack behind the size asked for)
        KcClearList cl;
        memset(&cl, 0, sizeof(cl));
        auto add = [&](void* q, size_t bytes) { cl.p[cl.count] = q; cl.n16[cl.count] = (bytes + 15) / 16; cl.count++; };
        add(c->redo.p, (size_t)n_units * 4);
        add(c->redo_blk.p, (size_t)nb + 1);
        add(c->errflag.p, 64);
        add(c->rawdef.p, ((size_t)nb + 1) * sizeof(KcRawDef));
        if (fuse_xxh) add(c->unit_raw.p, (size_t)n_units * 4);
        kc_launch_clear(cl, st);
    }
    ep.prof = nullptr;
    const bool k2prof = c->cfg.k2_prof != 0;
    if (k2prof) {
        if ((s = ensure(c, c->prof, 48 * 8)) != KC_OK) return s;
        HIPCHK(c, hipMemsetAsync(c->prof.p, 0, 48 * 8, st));
        ep.prof = (unsigned long long*)c->prof.p;
        mp.prof = (unsigned long long*)c->prof.p + 32;
    }

    if (c->chain_after) HIPCHK(c, hipStreamWaitEvent(st, c->chain_after->ev[2], 0));  // pipelined contexts: one match finder at a time
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    // The no-match pre-scan (kc_zstd_prescan.hip): SpeedFastest EncodeAll batches whose frames take the deferred-payload path
    // (fuse_xxh: checksum on, regular block grid, no dictionary / job prefix / chunk feed), literal-only blocks going out raw
    // (rawAllLits, the default below SpeedBetterCompression).  On by option, or per batch when the context's previous batch did not
    // compress (the same signal that picks the match finder's form for such input).
    c->prescan_ran = false;
    {
        const bool tuned_now = c->cfg.zfast_variant < 0 ? c->last_incompressible : c->cfg.zfast_variant == 1;
        const bool want = c->cfg.zfast_prescan > 0 || (c->cfg.zfast_prescan < 0 && tuned_now);
        if (want && o->level == KC_SPEED_FASTEST && fuse_xxh && !c->stream_mode && !o->all_lit_entropy && bs >= 16 && n_units > 0) {
            if (c->probe_bs != bs || c->probe_rel.p == nullptr) {
                std::vector<uint32_t> rel(4096);
                uint32_t n = kc_zfast_probe_positions(bs, rel.data(), (uint32_t)rel.size());
                if (n > rel.size()) { rel.resize(n); n = kc_zfast_probe_positions(bs, rel.data(), (uint32_t)rel.size()); }
                if ((s = ensure(c, c->probe_rel, (size_t)n * 4 + 16)) != KC_OK) return s;
                HIPCHK(c, hipMemcpyAsync(c->probe_rel.p, rel.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
                HIPCHK(c, hipStreamSynchronize(st));  // rel is a local
                c->probe_bs = bs;
                c->probe_n = n;
            }
            if ((s = ensure(c, c->unit_done, (size_t)n_units * 4 + 16)) != KC_OK) return s;
            KcPrescanParams pp;
            memset(&pp, 0, sizeof(pp));
            pp.src = k_src;
            pp.unit_off = k_off;
            pp.unit_blk0 = mp.unit_blk0;
            pp.n_units = n_units;
            pp.block_size = bs;
            pp.probe_rel = (const uint32_t*)c->probe_rel.p;
            pp.n_probe = c->probe_n;
            pp.rep1 = mp.rep1;
            pp.rep2 = mp.rep2;
            pp.meta = mp.meta;
            pp.unit_done = (uint32_t*)c->unit_done.p;
            pp.stage = ep.stage;
            pp.stage_off = ep.stage_off;
            pp.out_size = ep.out_size;
            pp.rawdef = ep.rawdef;
            pp.unit_raw = ep.unit_raw;
            pp.window_size = o->window_size;
            pp.crc = o->crc;
            pp.single = o->single;
            pp.dict_id = o->dict_id;
            kc_launch_zfast_prescan(pp, st);
            mp.unit_done = pp.unit_done;
            ep.unit_done = pp.unit_done;
            c->prescan_ran = true;
        }
    }
    mp.unit_base = 0;
    ep.unit_base = 0;
    if (feed == nullptr) {
        if (o->crc && !c->job_hist && !fuse_xxh) kc_launch_xxh64(d_src, (const uint64_t*)c->unit_off.p, n_units, (uint64_t*)c->xxh.p, st);  // (a job stream's checksum is the host's)
        HIPCHK(c, hipEventRecord(c->ev[1], st));
        if ((s = launch_match(c, mp, unit_off, n_units, n_units, bs, st, o->level)) != KC_OK) return s;
    } else {
        // the source is still arriving: per chunk, checksum + match finder on the chunk's stream behind its H2D copy
        if (useDict) { c->err = "chunk feed does not take dictionaries"; return KC_ERR_INTERNAL; }
        HIPCHK(c, hipEventRecord(c->ev[1], st));
        if ((s = prepare_tables(c, mp, n_units, st, o->level)) != KC_OK) return s;
        const bool feed_lds = zfast_use_lds(c, mp, n_units, o->level);
        HIPCHK(c, hipEventRecord(c->ev[6], st));  // everything the chunk kernels need from this stream (offset arrays, tables)
        const size_t nchunk = feed->cut.size() - 1;
        for (size_t k = 0; k < nchunk; k++) {
            if (!feed->wait_recorded(k)) { c->err = "host pipeline: staging failed"; return KC_ERR_HIP; }
            hipStream_t sk = feed->streams[k % feed->streams.size()];
            const uint32_t u0 = feed->cut[k], nk = feed->cut[k + 1] - feed->cut[k];
            HIPCHK(c, hipStreamWaitEvent(sk, c->ev[6], 0));
            HIPCHK(c, hipStreamWaitEvent(sk, feed->landed[k], 0));
            if (o->crc) kc_launch_xxh64(d_src, (const uint64_t*)c->unit_off.p + u0, nk, (uint64_t*)c->xxh.p + u0, sk);
            KcMatchParams mk = mp;
            mk.unit_base = u0;
            launch_match_kernel(c, mk, u0, nk, sk, o->level, feed_lds);
            KcEntropyParams ek = ep;
            ek.unit_base = u0;
            kc_launch_zstd_entropy(ek, nk, sk);
            feed->loc_off = (uint64_t*)c->out_off.p;
            kc_launch_scan_sizes((const uint32_t*)c->out_size.p + u0, nk, feed->loc_off + u0 + k, sk);
            kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p + u0, (const uint32_t*)c->out_size.p + u0,
                              feed->loc_off + u0 + k, d_dst + pl.stage_off[u0], nk, sk, ep.src, ep.unit_off + u0, ep.unit_blk0 + u0, ep.rawdef);
            HIPCHK(c, hipEventRecord(feed->done[k], sk));
        }
        for (size_t k = 0; k < nchunk; k++) HIPCHK(c, hipStreamWaitEvent(st, feed->done[k], 0));
    }
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    HIPCHK(c, hipGetLastError());
    Pending* P = new Pending();
    P->o = *o;
    P->mp = mp;
    P->ep = ep;
    P->unit_off.assign(unit_off, unit_off + n_units + 1);
    P->n_units = n_units;
    P->d_dst = d_dst;
    P->need = pl.stage_off[n_units];
    P->fed = feed != nullptr;
    P->bs = bs;
    P->k2prof = k2prof;
    c->pend = P;
    return KC_OK;
}

kc_status batch_end(kc_ctx* c, uint64_t* out_off_host, uint64_t* produced) {
    if (!c->pend) { c->err = "no batch in flight on this context"; return KC_ERR_BAD_ARG; }
    std::unique_ptr<Pending> P((Pending*)c->pend);
    c->pend = nullptr;
    hipStream_t st = c->stream;
    if (c->stream2 != nullptr) {  // the second stage on a stream of its own (e.g. one restricted to other CUs than the match finder's)
        st = c->stream2;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev[2], 0));
    }
    const kc_zstd_opts* o = &P->o;
    KcMatchParams& mp = P->mp;
    KcEntropyParams& ep = P->ep;
    const uint64_t* unit_off = P->unit_off.data();
    const uint32_t n_units = P->n_units;
    uint8_t* d_dst = P->d_dst;
    const int bs = P->bs;
    const bool k2prof = P->k2prof;
    kc_status s;
    kc_launch_zstd_entropy(ep, n_units, st);
    HIPCHK(c, hipEventRecord(c->ev[3], st));
    HIPCHK(c, hipGetLastError());

    // sizes -> offsets, checksum (+ payload of the raw-only frames), compaction.  Enqueued right behind the entropy stage, before the
    // host has looked at the re-run flags: a re-run is rare, and when one happens the pass simply runs again behind it (it rewrites
    // every byte of dst) — so the device does not idle through the flag read-back of every batch.
    auto finish_pass = [&]() -> kc_status {
        kc_launch_scan_sizes((const uint32_t*)c->out_size.p, n_units, (uint64_t*)c->out_off.p, st);
        if (ep.unit_raw != nullptr) {  // the checksum, and with it the payload of the frames that are raw blocks only
            KcXxhFinParams xf;
            xf.src = ep.src;
            xf.unit_off = ep.unit_off;
            xf.n_units = n_units;
            xf.stage = (uint8_t*)c->stage.p;
            xf.stage_off = (const uint64_t*)c->stage_off.p;
            xf.out_size = (const uint32_t*)c->out_size.p;
            xf.out_off = (const uint64_t*)c->out_off.p;
            xf.dst = d_dst;
            xf.unit_raw = ep.unit_raw;
            xf.rawdef = ep.rawdef;
            xf.unit_blk0 = ep.unit_blk0;
            xf.xxh_out = (uint64_t*)c->xxh.p;
            xf.mode = (int32_t)c->cfg.xxh_fin_mode;
            kc_launch_xxh64_fin(xf, st);
        }
        kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p, (const uint32_t*)c->out_size.p,
                          (const uint64_t*)c->out_off.p, d_dst, n_units, st, ep.src, ep.unit_off, ep.unit_blk0, ep.rawdef, ep.unit_raw);
        HIPCHK(c, hipGetLastError());
        return KC_OK;
    };
    HIPCHK(c, hipEventRecord(c->ev[4], st));
    if ((s = finish_pass()) != KC_OK) return s;
    HIPCHK(c, hipEventRecord(c->ev[5], st));
    // Speculation check: a block that fell back to raw only after entropy coding (blockenc.go:811-817)
    // pops the repeat offsets; if the following block was parsed with the un-popped offsets the unit is
    // re-run with that verdict forced.  Rare (needs a compressible-looking block that ends larger than raw).
    uint32_t redo_units = 0;
    {
        const Plan& pl = c->plan;  // this batch's layout (one batch in flight per context)
        const uint32_t nb = pl.n_blocks;
        uint32_t maxBlocks = 1;
        for (uint32_t i = 0; i < n_units; i++) maxBlocks = std::max(maxBlocks, pl.blk0[i + 1] - pl.blk0[i]);
        std::vector<uint32_t> redo(n_units), list;
        std::vector<uint8_t> redo_blk, pop_blk;
        uint32_t errv[16];
        for (uint32_t iter = 0;; iter++) {
            HIPCHK(c, hipMemcpyAsync(redo.data(), c->redo.p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(errv, c->errflag.p, 64, hipMemcpyDeviceToHost, st));
            std::vector<uint32_t> doneh;
            if (iter == 0 && c->prescan_ran) {
                doneh.resize(n_units);
                HIPCHK(c, hipMemcpyAsync(doneh.data(), c->unit_done.p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
            }
            HIPCHK(c, hipStreamSynchronize(st));
            if (iter == 0) {
                c->last_prescan_units = 0;
                for (uint32_t v : doneh) c->last_prescan_units += v != 0u;
            }
            if (errv[0] != 0) {
                char b[96];
                snprintf(b, sizeof(b), "device invariant violated (code %u)", errv[0]);
                c->err = b;
                return errv[0] == 100u ? KC_ERR_UNSUPPORTED : KC_ERR_INTERNAL;
            }
            list.clear();
            for (uint32_t i = 0; i < n_units; i++)
                if (redo[i]) list.push_back(i);
            if (list.empty()) break;
            if (iter > maxBlocks + 1) { c->err = "speculation re-run did not converge"; return KC_ERR_INTERNAL; }  // every pass settles one more block per unit
            redo_blk.resize(nb);
            if (pop_blk.empty()) pop_blk.assign(nb, 0);
            HIPCHK(c, hipMemcpyAsync(redo_blk.data(), c->redo_blk.p, nb, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            for (uint32_t i : list)
                for (uint32_t b = pl.blk0[i]; b < pl.blk0[i + 1]; b++)
                    if (redo_blk[b]) { pop_blk[b] = 1; break; }  // only the lowest flagged block is trustworthy
            redo_units += (uint32_t)list.size();
            HIPCHK(c, hipMemcpyAsync(c->pop_blk.p, pop_blk.data(), nb, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->unit_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemsetAsync(c->redo.p, 0, (size_t)n_units * 4, st));
            HIPCHK(c, hipMemsetAsync(c->redo_blk.p, 0, (size_t)nb + 1, st));
            c->job_redo_list = list;
            mp.pop_blk = (const uint8_t*)c->pop_blk.p;
            mp.unit_list = (const uint32_t*)c->unit_list.p;
            ep.unit_list = mp.unit_list;
            if ((s = launch_match(c, mp, unit_off, n_units, (uint32_t)list.size(), bs, st, o->level)) != KC_OK) return s;
            kc_launch_zstd_entropy(ep, (uint32_t)list.size(), st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));  // pop_blk / list are host vectors: the copies above must have been taken before the next pass rewrites them
        }
    }
    if (redo_units != 0) {  // the frames of the re-run units changed: sizes, offsets and everything behind them
        HIPCHK(c, hipEventRecord(c->ev[4], st));
        if ((s = finish_pass()) != KC_OK) return s;
        HIPCHK(c, hipEventRecord(c->ev[5], st));
    }
    HIPCHK(c, hipMemcpyAsync(out_off_host, c->out_off.p, (n_units + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t45 = 0, t05 = 0, tk2 = 0;
    (void)hipEventElapsedTime(&t01, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&t23, c->ev[2], c->ev[3]);
    (void)hipEventElapsedTime(&t34, c->ev[3], c->ev[4]);
    (void)hipEventElapsedTime(&t45, c->ev[4], c->ev[5]);
    (void)hipEventElapsedTime(&t05, c->ev[0], c->ev[5]);
    if (c->ev7_valid) {  // (the speculation re-run records it again: then it brackets the re-run's preparation, a few units)
        float t17 = 0;
        if (hipEventElapsedTime(&t17, c->ev[1], c->ev[7]) == hipSuccess && t17 >= 0 && t17 <= t12) c->last.prep_ms += t17;
        c->ev7_valid = false;
    }
    tk2 = t23;
    c->last.match_ms += t12;       // all match-finder launches (with overlap: includes time shared with entropy kernels)
    c->last.entropy_ms += tk2;     // first to last entropy launch on its stream
    c->last.other_ms += t01 + t34 + t45;
    c->last.total_ms += t05;
    c->last.redo_units += redo_units;
    if (k2prof) {
        unsigned long long pv[48];
        HIPCHK(c, hipMemcpy(pv, c->prof.p, sizeof(pv), hipMemcpyDeviceToHost));
        {
            unsigned long long lt = 0;
            for (int i = 32; i < 40; i++) lt += pv[i];
            if (lt && o->level == KC_SPEED_DEFAULT) {  // -DKC_ZD_STATS build of kc_zstd_match_dfast.hip
                fprintf(stderr, "[dfast stats] per unit: probes looked up %.0f, committed %.0f, candidate / repeat 16-byte loads %.0f, long lookups at s+1 %.0f, matches %.0f, offset-2 matches %.0f, ring refills (128 B) %.0f\n",
                        (double)pv[32] / n_units, (double)pv[33] / n_units, (double)pv[34] / n_units, (double)pv[35] / n_units, (double)pv[36] / n_units, (double)pv[37] / n_units, (double)pv[38] / n_units);
            } else if (lt) {
                fprintf(stderr, "[LDS match prof] shader clocks per phase (window, probe bytes, table, candidates issued, verdicts, commit, -, round tail):");
                for (int i = 32; i < 40; i++) fprintf(stderr, " %.1f%%", 100.0 * (double)pv[i] / (double)lt);
                fprintf(stderr, "  (total %.4g cycles over %u units)\n", (double)lt, n_units);
            }
        }
        {
            unsigned long long ft = 0;
            for (int i = 40; i < 48; i++) ft += pv[i];
            if (ft) {
                fprintf(stderr, "[K2 fine] wave 0, shader clocks per unit; -DKC_K2_FINE=1: gather (pass A, barrier, step: sequences + scan, step: literal loads + ORs, step: flush + histogram, step: carry + zero, tail, closing barrier); =2: (Huffman size pass, payload zero fill + barrier, stream emit, barrier, payload copy + headers, code staging, chains, pack):");
                for (int i = 40; i < 48; i++) fprintf(stderr, " %.0f", (double)pv[i] / (double)(n_units ? n_units : 1));
                fprintf(stderr, "\n");
            }
        }
        unsigned long long tot = 0;
        for (int i = 0; i < 16; i++) tot += pv[i];
        fprintf(stderr, "[K2 prof] shader-clock share per phase:");
        for (int i = 0; i < 14; i++) fprintf(stderr, " p%d=%.1f%%", i, tot ? 100.0 * (double)pv[i] / (double)tot : 0.0);
        fprintf(stderr, "  (total %.3g cycles over %u units)\n", (double)tot, n_units);
        fprintf(stderr, "[K2 prof] tANS chains (counted with -DKC_CHAIN_STATS): %llu chunk-streams (%llu RLE, %llu predefined), %llu repair passes (LL %llu, OF %llu, ML %llu; %llu on RLE tables), %llu segments re-encoded\n",
                pv[16], pv[26], pv[27], pv[17], pv[18], pv[19], pv[20], pv[25], pv[24]);
    }
    *produced = out_off_host[n_units];
    {
        const uint64_t in_total = unit_off[n_units] - unit_off[0];
        c->last_incompressible = in_total >= (1u << 20) && (double)*produced >= 0.98 * (double)in_total;
    }
    return KC_OK;
}

// A batch is bounded by its input bytes AND by the device scratch it needs: tables are per unit, sequences / literals /
// staging are per block at a fixed stride whatever the block's actual length, so many small units need far more than the
// "~6x input" of full-size units (1M x 4 KiB units at SpeedDefault would ask for hundreds of GiB in one batch).
uint64_t zstd_unit_scratch(const kc_zstd_opts* o, uint64_t len, uint64_t n_cuts) {
    const uint64_t bsz = (uint64_t)o->block_size;
    const uint64_t table_b = o->level == KC_SPEED_BEST ? 0 : o->level == KC_SPEED_BETTER ? kc_zbetter_table_bytes() : (o->level == KC_SPEED_DEFAULT ? kc_zdfast_table_bytes() : kc_zfast_table_bytes());
    const uint64_t per_block = 2 * (bsz / 4 + 8) * 8 + (bsz + 64) + sizeof(KcBlkMeta);
    const uint64_t hist0 = (o->dict != nullptr) ? o->dict_len : 0;
    const uint64_t blocks = (len + bsz - 1) / bsz + n_cuts;  // every Flush point can add a block, at the full per-block strides
    const uint64_t enc = ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)len) + 3 * n_cuts + 3 + 15) & ~(uint64_t)15;
    return table_b + blocks * per_block + enc + (hist0 ? hist0 + len : 0) + 64;
}

// Scratch a batch may ask for: the configured ceiling, or 85 % of what is free plus what this context already owns (re-used).
uint64_t scratch_budget(kc_ctx* c) {
    uint64_t budget = c->max_scratch_bytes;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        uint64_t held = 0;
        const DevBuf* bufs[] = {&c->seqs, &c->aux, &c->lits, &c->meta, &c->stage, &c->tables, &c->work};
        for (const DevBuf* b : bufs) held += b->cap;
        const uint64_t avail = (uint64_t)((double)(fr + held) * 0.85);
        if (avail < budget) budget = avail;
    } else {
        (void)hipGetLastError();
    }
    return budget;
}

// End of a chunk-fed batch (batch_begin with a ChunkFeed): every chunk has already been entropy coded and compacted on its own
// stream.  *redo is set when a unit needs the speculation re-run (see batch_end): the caller encodes the batch again the plain way.
kc_status feed_finish(kc_ctx* c, bool* redo_needed) {
    if (!c->pend) { c->err = "no batch in flight on this context"; return KC_ERR_BAD_ARG; }
    std::unique_ptr<Pending> P((Pending*)c->pend);
    c->pend = nullptr;
    hipStream_t st = c->stream;
    const uint32_t n_units = P->n_units;
    std::vector<uint32_t> redo(n_units);
    uint32_t errv[16];
    HIPCHK(c, hipMemcpyAsync(redo.data(), c->redo.p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(errv, c->errflag.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    if (errv[0] != 0) {
        char b[96];
        snprintf(b, sizeof(b), "device invariant violated (code %u)", errv[0]);
        c->err = b;
        return errv[0] == 100u ? KC_ERR_UNSUPPORTED : KC_ERR_INTERNAL;
    }
    *redo_needed = false;
    for (uint32_t i = 0; i < n_units; i++) if (redo[i]) *redo_needed = true;
    return KC_OK;
}

kc_status run_batch(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src_base, const uint64_t* unit_off, uint32_t n_units,
                    uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off_host, uint64_t* produced) {
    kc_status s = batch_begin(c, o, d_src_base, unit_off, n_units, d_dst, dst_cap);
    if (s != KC_OK) return s;
    return batch_end(c, out_off_host, produced);
}

}  // namespace kci

extern "C" {

kc_status kc_zstd_encode_units_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                   uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off || !out_off || (n_units && (!d_src || !d_dst))) return KC_ERR_BAD_ARG;
    c->err.clear();
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    for (uint32_t i = 0; i < n_units; i++) {
        if (unit_off[i + 1] < unit_off[i]) { c->err = "unit_off not ascending"; return KC_ERR_BAD_ARG; }
        if (unit_off[i + 1] - unit_off[i] > KC_MAX_UNIT_BYTES) {
            c->err = "unit larger than 1 GiB: not served by the device path";
            return KC_ERR_UNSUPPORTED;
        }
    }
    out_off[0] = 0;
    c->last_batches = 0;
    uint64_t pos = 0;
    uint32_t i0 = 0;
    std::vector<uint64_t> tmp;
    auto unit_scratch = [&](uint32_t i) {
        const uint64_t nc = c->cuts ? c->cut_off[i + 1] - c->cut_off[i] : 0;
        return zstd_unit_scratch(o, unit_off[i + 1] - unit_off[i], nc);
    };
    uint64_t budget = scratch_budget(c);
    for (int attempt = 0;; attempt++) {
        bool oom = false;
        while (i0 < n_units) {
            uint32_t i1 = i0;
            const uint64_t cap_bytes = o->level == KC_SPEED_BETTER ? ((uint64_t)1 << 30) : c->max_batch_bytes;  // better: 4 MiB of tables per unit
            const uint32_t cap_units = o->level == KC_SPEED_BETTER ? 16384u : 0xFFFFFFFFu;
            uint64_t scratch = 0;
            while (i1 < n_units) {
                const uint64_t us = unit_scratch(i1);
                // ensure() over-allocates by 1/8
                if (i1 > i0 && (unit_off[i1 + 1] - unit_off[i0] > cap_bytes || i1 - i0 >= cap_units || (scratch + us) + ((scratch + us) >> 3) > budget)) break;
                scratch += us;
                i1++;
            }
            const uint32_t nb = i1 - i0;
            tmp.resize(nb + 1);
            uint64_t produced = 0;
            c->cut_unit0 = i0;
            c->oom = false;
            s = run_batch(c, o, d_src, unit_off + i0, nb, d_dst + pos, dst_cap - pos, tmp.data(), &produced);
            if (s == KC_ERR_UNSUPPORTED && c->oom && nb > 1 && attempt < 6) { oom = true; break; }
            if (s != KC_OK) return s;
            c->last_batches++;
            for (uint32_t k = 0; k <= nb; k++) out_off[i0 + k] = pos + tmp[k];
            pos += produced;
            i0 = i1;
        }
        if (!oom) break;
        budget /= 2;  // another process took device memory since hipMemGetInfo: retry this batch at half the size
        c->err.clear();
    }
    if (n_units == 0) out_off[0] = 0;
    return KC_OK;
}

}  // extern "C"
namespace kci {
kc_status validate_units(kc_ctx* c, const kc_zstd_opts* o, const uint64_t* unit_off, uint32_t n_units) {
    for (uint32_t i = 0; i < n_units; i++) {
        if (unit_off[i + 1] < unit_off[i]) { c->err = "unit_off not ascending"; return KC_ERR_BAD_ARG; }
        if (unit_off[i + 1] - unit_off[i] > KC_MAX_UNIT_BYTES) {
            c->err = "unit larger than 1 GiB: not served by the device path";
            return KC_ERR_UNSUPPORTED;
        }
    }
    return KC_OK;
}
}  // namespace kci
extern "C" {

kc_status kc_zstd_encode_units_dev_begin(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                         uint8_t* d_dst, uint64_t dst_cap) {
    if (!c || !o || !unit_off || n_units == 0 || !d_src || !d_dst) return KC_ERR_BAD_ARG;
    c->err.clear();
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    if ((s = validate_units(c, o, unit_off, n_units)) != KC_OK) return s;
    const uint64_t cap_bytes = o->level == KC_SPEED_BETTER ? ((uint64_t)1 << 30) : c->max_batch_bytes;
    if (unit_off[n_units] - unit_off[0] > cap_bytes || (o->level == KC_SPEED_BETTER && n_units > 16384u)) {
        c->err = "begin/end serves one device batch; use kc_zstd_encode_units_dev for larger inputs";
        return KC_ERR_UNSUPPORTED;
    }
    return batch_begin(c, o, d_src, unit_off, n_units, d_dst, dst_cap);
}

kc_status kc_zstd_encode_units_dev_end(kc_ctx* c, uint64_t* out_off) {
    if (!c || !out_off) return KC_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t produced = 0;
    return batch_end(c, out_off, &produced);
}

kc_status kc_zstd_encode_units_dev_end_at(kc_ctx* c, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !out_off || !d_dst) return KC_ERR_BAD_ARG;
    if (!c->pend) { c->err = "no batch in flight on this context"; return KC_ERR_BAD_ARG; }
    Pending* P = (Pending*)c->pend;
    // the frames leave the staging slots only in _end (sizes -> offsets -> compaction): until then their place can still be named
    if (P->fed) { c->err = "a chunk-fed batch has written its frames already"; return KC_ERR_UNSUPPORTED; }
    if (P->need > dst_cap) { c->err = "dst_cap smaller than the sum of MaxEncodedSize(unit)"; return KC_ERR_DST_TOO_SMALL; }
    P->d_dst = d_dst;
    return kc_zstd_encode_units_dev_end(c, out_off);
}

void kc_ctx_chain_after(kc_ctx* c, kc_ctx* prev) {
    if (c) c->chain_after = prev;
}

kc_status kc_zstd_encode_streams_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                     uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o) return KC_ERR_BAD_ARG;
    c->stream_mode = 1;
    const kc_status s = kc_zstd_encode_units_dev(c, o, d_src, unit_off, n_units, d_dst, dst_cap, out_off);
    c->stream_mode = 0;
    return s;
}

}  // extern "C"

