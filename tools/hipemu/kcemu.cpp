// kcemu — the LDS-table kernels of compress_amd/csrc compiled for the hipemu CPU emulator, behind a small C API for the
// tests (tests/test_emu_lds.py).  TEST INFRASTRUCTURE: the product (libkcgpu.so) never contains or loads this.
#include <hip/hip_runtime.h>
#include "../../compress_amd/csrc/kc_s2_lds.hip"
#include "../../compress_amd/csrc/kc_s2.hip"
#include "../../compress_amd/csrc/kc_zstd_match_lds.hip"
#include "../../compress_amd/csrc/kc_zstd_match.hip"
#include "../../compress_amd/csrc/kc_misc.hip"
#include "../../compress_amd/csrc/kc_zstd_entropy.hip"
#include "../../compress_amd/csrc/kc_zstd_prescan.hip"
#include "../../compress_amd/csrc/kc_zstd_prime.hip"
#include "../../compress_amd/csrc/kc_zstd_match_dfast.hip"
#include "../../compress_amd/csrc/kc_zstd_match_better.hip"
#include <vector>
#include "../../compress_amd/csrc/kc_s2_best.hip"
#include "../../compress_amd/csrc/kc_zstd_match_best.hip"

extern "C" {

// kc_zstd_prime_kernel: n table slots (zeroed by the caller) primed from the first unit_hist[u] bytes of their units; reverse = the
// emulator keeps the LOWEST lane's value where lanes of one store instruction share an address (the hardware may keep any)
int kcemu_zstd_prime(int level, int pos_bits, int reverse, const uint8_t* src, const uint64_t* unit_off, const uint32_t* unit_hist,
                     const uint32_t* unit_list, uint32_t n_launch, uint8_t* tables, uint64_t table_bytes) {
    KcPrimeParams P;
    memset(&P, 0, sizeof(P));
    P.src = src; P.unit_off = unit_off; P.unit_hist = unit_hist; P.unit_list = unit_list; P.unit_base = 0; P.n_launch = n_launch;
    P.level = level; P.pos_bits = pos_bits; P.tables = tables; P.table_bytes = (size_t)table_bytes;
    hipemu::set_reverse(reverse != 0);
    kc_launch_zstd_prime(P, nullptr);
    hipemu::set_reverse(false);
    return 0;
}

// N blocks through kc_s2_encode_lds_kernel; stage slots as the host library lays them out (stage_off, 64-byte aligned)
int kcemu_s2_encode(int level, int framed, int spec_w0, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* stage,
                    const uint64_t* stage_off, uint32_t* out_size) {
    KcS2Params P;
    memset(&P, 0, sizeof(P));
    P.src = src;
    P.blk_off = blk_off;
    P.stage_off = stage_off;
    P.stage = stage;
    P.out_size = out_size;
    P.n_blocks = n;
    P.level = level & 0xFF;
    P.variant = (level >> 8) & 0xFF;  // (bits 8..15: KC_S2_VARIANT_*)
    P.stored_only = (level >> 16) & 1; // (bit 16: s2.WriterUncompressed)
    P.spec_w0 = spec_w0;
    P.framed = framed;
    bool small = false, big = false;
    for (uint32_t i = 0; i < n; i++) ((blk_off[i + 1] - blk_off[i]) <= 65536 ? small : big) = true;
    kc_launch_s2_encode_lds(P, small, big, nullptr);
    return 0;
}

// N blocks through kc_s2_encode_kernel<LEVEL> (the HBM-table throughput kernel, 8 lanes per block: s2.Encode / EncodeBetter / EncodeSnappy /
// EncodeSnappyBetter; level bits 8+: KC_S2_VARIANT_*), tables as the host sizes and zeroes them
int kcemu_s2_hbm(int level, int framed, int w0, int w0b, int grow, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* stage,
                 const uint64_t* stage_off, uint32_t* out_size) {
    KcS2Params P;
    memset(&P, 0, sizeof(P));
    P.src = src; P.blk_off = blk_off; P.stage_off = stage_off; P.stage = stage; P.out_size = out_size; P.n_blocks = n;
    P.level = level & 0xFF; P.variant = level >> 8; P.spec_w0 = w0; P.spec_w0b = w0b; P.spec_grow = grow; P.framed = framed;
    uint64_t mx = 0;
    for (uint32_t i = 0; i < n; i++) mx = std::max<uint64_t>(mx, blk_off[i + 1] - blk_off[i]);
    const size_t tb = kc_s2_table_bytes(P.level, mx, P.variant);
    std::vector<uint32_t> tab((size_t)((n + 7) / 8 * 8) * (tb / 4), 0);
    P.tables = tab.data(); P.table_stride = (uint32_t)(tb / 4);
    hipemu::set_group(8);  // 8 blocks per wave, the groups diverge freely
    kc_launch_s2_encode(P, nullptr);
    hipemu::set_group(64);
    return 0;
}

// the SpeedFastest match finder (kc_zfast_match_lds_kernel) over n units: packed sequences + KcBlkMeta per block
int kcemu_zfast_parse(const uint8_t* src, const uint64_t* unit_off, uint32_t n, int block_size, int window, int hist0, int rep1, int rep2,
                      int stream_mode, const uint32_t* proto, uint64_t* seqs, KcBlkMeta* meta, uint32_t seq_stride, const uint32_t* unit_blk0,
                      int spec_w0, int pos_bits) {
    KcMatchParams P;
    memset(&P, 0, sizeof(P));
    P.src = src;
    P.src_end = src + unit_off[n];
    P.unit_off = unit_off;
    P.unit_blk0 = unit_blk0;
    P.seqs = seqs;
    P.meta = meta;
    P.seq_stride = seq_stride;
    P.block_size = block_size;
    P.max_match_off = window;
    P.spec_w0 = spec_w0;
    P.hist0 = hist0;
    P.pos_bits = pos_bits;
    P.rep1 = rep1;
    P.rep2 = rep2;
    P.stream_mode = stream_mode;
    for (uint32_t i = 0; i < n; i++) if (unit_off[i + 1] - unit_off[i] > 131072) P.lds_any_big = 1;
    kc_launch_zfast_match_lds(P, proto, 0u, n, nullptr);
    return 0;
}

// the SpeedFastest HBM-table group kernel (kc_zfast_match_grp_kernel<8>) over n units; tables: n x 2^15 u32 kept by the caller between
// calls (epoch 0: zeroed by the caller; else the launch's stamp)
int kcemu_zfast_parse_grp(const uint8_t* src, const uint64_t* unit_off, uint32_t n, int block_size, int window, int stream_mode, uint64_t* seqs,
                          KcBlkMeta* meta, uint32_t seq_stride, const uint32_t* unit_blk0, int spec_w0, int spec_grow, int pos_bits, uint32_t* tables,
                          uint32_t epoch, int xseg_k) {
    KcMatchParams P;
    memset(&P, 0, sizeof(P));
    P.src = src;
    P.src_end = src + unit_off[n];
    P.unit_off = unit_off;
    P.unit_blk0 = unit_blk0;
    P.seqs = seqs;
    P.meta = meta;
    P.seq_stride = seq_stride;
    P.block_size = block_size;
    P.max_match_off = window;
    P.spec_w0 = spec_w0;
    P.spec_grow = spec_grow;
    P.pos_bits = pos_bits;
    P.rep1 = 1;
    P.rep2 = 4;
    P.stream_mode = stream_mode;
    P.epoch = epoch;
    P.xseg_k = xseg_k & 0xFFFFFF;
    P.tuned = P.xseg_k < (1 << 20) ? 1 : 0;  // (2^20: the plain form, rounds inside one skip segment, no filter)
    P.empty_filter = (xseg_k >> 24) & 1 ? 0 : 1;  // (bit 24 of the argument switches the filter off)
    hipemu::set_group(8);  // 8 units per wave, the groups diverge freely
    kc_launch_zfast_match_grp(P, tables, n, nullptr);
    hipemu::set_group(64);
    return 0;
}

// kc_xxh64_fin_kernel: the checksum field of every frame, and the payload of the frames flagged in unit_raw copied from the source
int kcemu_xxh_fin(const uint8_t* src, const uint64_t* unit_off, uint32_t n, uint8_t* stage, const uint64_t* stage_off, const uint32_t* out_size,
                  const uint64_t* out_off, uint8_t* dst, const uint32_t* unit_raw, const KcRawDef* rawdef, const uint32_t* unit_blk0, uint64_t* xxh_out,
                  int mode) {
    KcXxhFinParams P;
    memset(&P, 0, sizeof(P));
    P.src = src; P.unit_off = unit_off; P.n_units = n; P.stage = stage; P.stage_off = stage_off; P.out_size = out_size; P.out_off = out_off;
    P.dst = dst; P.unit_raw = unit_raw; P.rawdef = rawdef; P.unit_blk0 = unit_blk0; P.xxh_out = xxh_out; P.mode = mode;
    hipemu::set_group(4);  // one unit per quad; the quads' trip counts differ
    kc_launch_xxh64_fin(P, nullptr);
    hipemu::set_group(64);
    return 0;
}

// The whole SpeedFastest EncodeAll pipeline of the device on the emulator: checksum kernel, match finder (LDS-table kernel, or with
// use_grp = 1 the HBM-table group kernel in the form `tuned`; use_grp = 2 / 3 / 4: the SpeedDefault / SpeedBetterCompression /
// SpeedBestCompression match finders), entropy stage — the frames as they sit in the staging slots (raw blocks'
// payloads included: no rawdef), with the host's layout rules (seq_stride, lit_stride: kc_batch.cpp batch_begin).
int kcemu_zstd_frames(const uint8_t* src, const uint64_t* unit_off, uint32_t n, int block_size, int window, int crc, int single, int full_zero,
                      int stream_mode, int use_grp, int tuned, int entropy_opts /* bit 0: no_entropy, bit 1: all_lit_entropy */, uint8_t* stage, const uint64_t* stage_off, uint32_t* out_size, uint32_t* err_out,
                      uint8_t* fused_dst /* or null */, uint64_t* fused_off, int fused_mode) {
    std::vector<uint32_t> blk0(n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t len = unit_off[i + 1] - unit_off[i];
        blk0[i + 1] = blk0[i] + (uint32_t)((len + (uint64_t)block_size - 1) / (uint64_t)block_size);
    }
    const uint32_t nb = blk0[n];
    const uint32_t seq_stride = (uint32_t)(block_size / 4 + 8), lit_stride = (uint32_t)(block_size + 64);
    std::vector<uint64_t> seqs((size_t)(nb + 1) * seq_stride, 0), aux((size_t)(nb + 1) * seq_stride, 0), xxh(n + 1, 0);
    std::vector<uint8_t> lits((size_t)(nb + 1) * lit_stride, 0), redo_blk(nb + 1, 0), predef(kc_fse_predef_bytes() + 64, 0);
    std::vector<KcBlkMeta> meta(nb + 1);
    std::vector<uint32_t> redo(n + 1, 0), tables;
    uint32_t err[16] = {0};
    uint64_t maxlen = 16;
    for (uint32_t i = 0; i < n; i++) maxlen = std::max<uint64_t>(maxlen, unit_off[i + 1] - unit_off[i]);
    int pb = 1;
    while (((uint64_t)1 << pb) <= maxlen + 2) pb++;
    KcMatchParams M;
    memset(&M, 0, sizeof(M));
    M.src = src; M.src_end = src + unit_off[n]; M.unit_off = unit_off; M.unit_blk0 = blk0.data(); M.seqs = seqs.data(); M.meta = meta.data();
    M.seq_stride = seq_stride; M.block_size = block_size; M.max_match_off = window; M.pos_bits = pb; M.rep1 = 1; M.rep2 = 4; M.rep3 = 8;
    M.stream_mode = stream_mode;
    kc_launch_fse_predef_init(predef.data(), nullptr);
    if (crc) {
        hipemu::set_group(4);
        kc_launch_xxh64(src, unit_off, n, xxh.data(), nullptr);
        hipemu::set_group(64);
    }
    // the no-match pre-scan in front of the match finder (tuned bit 8; kc_batch.cpp batch_begin runs it on fused batches only)
    const bool prescan = (tuned & 0x100) != 0 && fused_dst != nullptr && crc && use_grp <= 1;
    tuned &= 0xFF;
    std::vector<KcRawDef> rawdef;
    std::vector<uint32_t> unit_raw, unit_done, probe_rel;
    if (fused_dst != nullptr) {
        rawdef.assign(nb + 1, KcRawDef{0, 0, 0, 0});
        unit_raw.assign(n + 1, 0);
    }
    if (prescan) {
        probe_rel.resize(8192);
        const uint32_t np = kc_zfast_probe_positions(block_size, probe_rel.data(), (uint32_t)probe_rel.size());
        unit_done.assign(n + 1, 0xFFFFFFFFu);
        KcPrescanParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.src = src; Q.unit_off = unit_off; Q.unit_blk0 = blk0.data(); Q.n_units = n; Q.block_size = block_size; Q.probe_rel = probe_rel.data();
        Q.n_probe = np; Q.rep1 = 1; Q.rep2 = 4; Q.meta = meta.data(); Q.unit_done = unit_done.data(); Q.stage = stage; Q.stage_off = stage_off;
        Q.out_size = out_size; Q.rawdef = rawdef.data(); Q.unit_raw = unit_raw.data(); Q.window_size = window; Q.crc = crc; Q.single = single;
        kc_launch_zfast_prescan(Q, nullptr);
        M.unit_done = unit_done.data();
        uint32_t nd = 0;
        for (uint32_t i = 0; i < n; i++) nd += unit_done[i] == 1u;
        err_out[3] = nd;
    }
    std::vector<uint64_t> btab;
    if (use_grp == 4) {  // SpeedBestCompression: kc_zbest_match_kernel on two persistent table slots, the bit costs from the device's own kernel
        btab.assign((size_t)2 * (kc_zbest_table_bytes() / 8), 0);
        uint32_t cur[2] = {0, 0};
        int32_t cost[96];
        memset(cost, 0, sizeof(cost));
        kc_launch_zbest_cost(predef.data(), cost, nullptr);
        kc_launch_zbest_match(M, btab.data(), cur, cost, n, 2, nullptr);
    } else if (use_grp == 2) {  // SpeedDefault: kc_zdfast_match_grp_kernel, 8 lanes per unit
        tables.assign((size_t)((n + 7) / 8 * 8) * (kc_zdfast_table_bytes() / 4), 0);
        M.spec_w0 = 2; M.spec_grow = 2;
        if (const char* e = getenv("KC_EMU_SPEC_W0")) M.spec_w0 = atoi(e);      // (tests: other speculation policies = other round shapes)
        if (const char* e = getenv("KC_EMU_SPEC_GROW")) M.spec_grow = atoi(e);
        hipemu::set_group(8);
        kc_launch_zdfast_match_grp(M, tables.data(), n, nullptr);
        hipemu::set_group(64);
    } else if (use_grp == 3) {  // SpeedBetterCompression: kc_zbetter_match_grp_kernel, 16 lanes per unit, tables cleared (no stamps)
        tables.assign((size_t)((n + 3) / 4 * 4) * (kc_zbetter_table_bytes() / 4), 0);
        M.spec_w0 = 16; M.spec_grow = 0;
        hipemu::set_group(16);
        kc_launch_zbetter_match_grp(M, (uint8_t*)tables.data(), n, false, nullptr);
        hipemu::set_group(64);
    } else if (use_grp) {
        tables.assign((size_t)((n + 7) / 8 * 8) << 15, 0);
        M.spec_w0 = 1; M.spec_grow = 1; M.tuned = tuned; M.empty_filter = 1;
        hipemu::set_group(8);
        kc_launch_zfast_match_grp(M, tables.data(), n, nullptr);
        hipemu::set_group(64);
    } else {
        M.spec_w0 = 16;
        kc_launch_zfast_match_lds(M, nullptr, 0u, n, nullptr);
    }
    KcEntropyParams E;
    memset(&E, 0, sizeof(E));
    E.src = src; E.unit_off = unit_off; E.unit_blk0 = blk0.data(); E.seqs = seqs.data(); E.meta = meta.data(); E.lits = lits.data(); E.aux = aux.data();
    E.stage = stage; E.stage_off = stage_off; E.out_size = out_size; E.xxh = xxh.data(); E.redo_mask = redo.data(); E.redo_blk = redo_blk.data();
    E.predef = predef.data(); E.seq_stride = seq_stride; E.lit_stride = lit_stride; E.block_size = block_size; E.window_size = window;
    E.crc = crc; E.single = single; E.no_entropy = entropy_opts & 1; E.all_lit_entropy = ((entropy_opts >> 1) & 1) | (use_grp >= 3 && use_grp <= 4 ? 1 : 0) /* allLitEntropy: levels above SpeedDefault */; E.full_zero = full_zero; E.stream_mode = stream_mode;
    E.err_flag = err;
    if (prescan) E.unit_done = unit_done.data();
    if (fused_dst != nullptr) {  // the layout of kc_batch.cpp batch_end: raw payloads deferred, checksum behind the entropy stage
        E.rawdef = rawdef.data();
        E.unit_raw = unit_raw.data();
        if (crc) E.xxh = nullptr;
    }
    kc_launch_zstd_entropy(E, n, nullptr);
    if (fused_dst != nullptr) {
        kc_launch_scan_sizes(out_size, n, fused_off, nullptr);
        if (crc) {
            KcXxhFinParams X;
            memset(&X, 0, sizeof(X));
            X.src = src; X.unit_off = unit_off; X.n_units = n; X.stage = stage; X.stage_off = stage_off; X.out_size = out_size; X.out_off = fused_off;
            X.dst = fused_dst; X.unit_raw = unit_raw.data(); X.rawdef = rawdef.data(); X.unit_blk0 = blk0.data(); X.xxh_out = xxh.data(); X.mode = fused_mode;
            hipemu::set_group(4);
            kc_launch_xxh64_fin(X, nullptr);
            hipemu::set_group(64);
        }
        kc_launch_compact(stage, stage_off, out_size, fused_off, fused_dst, n, nullptr, src, unit_off, blk0.data(), rawdef.data(), crc ? unit_raw.data() : nullptr);
        uint32_t nraw = 0;
        for (uint32_t i = 0; i < n; i++) nraw += unit_raw[i];
        err_out[2] = nraw;
    }
    uint32_t anyredo = 0;
    for (uint32_t i = 0; i < n; i++) anyredo |= redo[i];
    err_out[0] = err[0];
    err_out[1] = anyredo;  // (a unit that needs the speculation re-run: the host would run it again; the test picks inputs that do not)
    return 0;
}

// N blocks through kc_s2_best_kernel (level 4: s2.EncodeBest, 5: s2.EncodeSnappyBest); tables: n x (4.5 MiB / 4) u32, zeroed by the caller
int kcemu_s2_best(int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* stage, const uint64_t* stage_off, uint32_t* out_size,
                  uint32_t* tables) {
    KcS2Params P;
    memset(&P, 0, sizeof(P));
    P.src = src;
    P.blk_off = blk_off;
    P.stage_off = stage_off;
    P.stage = stage;
    P.out_size = out_size;
    P.tables = tables;
    P.table_stride = (uint32_t)(kc_s2_table_bytes(level, 0) / 4);
    P.n_blocks = n;
    P.level = level;
    hipemu::set_group(SBG);  // 16 lanes per block, the groups of a wave diverge freely
    kc_launch_s2_best(P, nullptr);
    hipemu::set_group(64);
    return 0;
}

// n units through kc_zbest_match_kernel (SpeedBestCompression) on n_slots persistent table slots; tables: n_slots x 34 MiB / 8 u64,
// slot_cur: n_slots u32 (0 = fresh); cost: 96 int32 (the predefined-table bit costs)
int kcemu_zbest_parse(const uint8_t* src, const uint64_t* unit_off, uint32_t n, int block_size, int window, int hist0, int rep1, int rep2, int rep3,
                      int stream_mode, uint64_t* seqs, KcBlkMeta* meta, uint32_t seq_stride, const uint32_t* unit_blk0, const uint32_t* unit_hist,
                      const uint32_t* job_flags, uint64_t* tables, uint32_t* slot_cur, const int32_t* cost, uint32_t n_slots) {
    KcMatchParams P;
    memset(&P, 0, sizeof(P));
    P.src = src;
    P.src_end = src + unit_off[n];
    P.unit_off = unit_off;
    P.unit_blk0 = unit_blk0;
    P.seqs = seqs;
    P.meta = meta;
    P.seq_stride = seq_stride;
    P.block_size = block_size;
    P.max_match_off = window;
    P.hist0 = hist0;
    P.rep1 = rep1;
    P.rep2 = rep2;
    P.rep3 = rep3;
    P.stream_mode = stream_mode;
    P.unit_hist = unit_hist;
    P.job_flags = job_flags;
    kc_launch_zbest_match(P, tables, slot_cur, cost, n, n_slots, nullptr);
    return 0;
}

uint64_t kcemu_collectives() { return hipemu::collectives(); }

}  // extern "C"
