// driver.cpp — the C entry points of oracle/_ref/libzstdref.so around the TRANSLATED reference encoder (go2cpp.py's output is
// #included below; nothing of it is in the repository).  TEST INFRASTRUCTURE.  What is hand-written here is what a Go caller of the
// reference writes: zstd.NewWriter(nil, opts...) — i.e. encoderOptions.setDefault() and the With* option functions, in the
// caller's order — then EncodeAll(src, nil), which is encodeAll on a fresh encoder of the chosen level (encoder.go:71-99, 722-729).
#include GOREF_GENERATED
#include <pthread.h>

namespace {
thread_local long long g_readfrom_at = -1;
struct Call {
    const uint8_t* src; int64_t n; uint8_t* dst; int64_t cap;
    int level, window, crc, single, full_zero, no_entropy, all_lit, lowmem;
    const uint8_t* dict; int64_t dict_len; uint32_t dict_id;
    int64_t result; char err[256];
    int concurrency = 0;            // WithEncoderConcurrency (0: the reference's default, GOMAXPROCS; 1: the synchronous nextBlock)
    const int64_t* cuts = nullptr;  // stream mode: Flush() after these input offsets (ascending)
    int64_t n_cuts = -1;            // -1: EncodeAll
    int dict_full = 0;              // the dictionary bytes are a full-format dictionary: WithEncoderDict (loadDict) instead of WithEncoderDictRaw
    int jobs = 0;                   // WithConcurrentBlocks(true)
    int64_t readfrom_at = -1;       // stream mode: >= 0 = the input from this offset on goes through ReadFrom(bytes.NewReader(src[at:])) instead of Write
    const int64_t* unit_off = nullptr;  // EncodeAll mode: n_units + 1 offsets into src — every unit through encodeAll on the SAME encoder, one
    int64_t n_units = 0;                // after the other, like the calls of one goroutine on one zstd.Encoder (its pool hands the encoder back)
    int64_t* out_off = nullptr;
};

void init_packages() {
    using namespace go;
    rt::Permanent perm;  // package variables outlive the call that happens to build them
    fse::go_init(); huff0::go_init(); xxhash::go_init(); compress::go_init(); zstd::go_init();
    zstd::initPredefined();  // NewWriter's first statement (encoder.go:72)
}
// zstd.NewWriter's option loop (encoder.go:73-80), the options in the order a caller would write them: level first — it sets the
// window / block size / allLitEntropy defaults — then the explicit overrides
void apply_options(zstd::Encoder& e, const Call* c) {
    using namespace go;
    e.o.setDefault();
    auto apply = [&](zstd::EOption opt) { error er = opt(&e.o); if (er != nil) panic(er); };
    if (c->level > 0) apply(zstd::WithEncoderLevel(zstd::EncoderLevel(K((long long)c->level))));
    if (c->window > 0) apply(zstd::WithWindowSize(Int(K((long long)c->window))));
    if (c->crc >= 0) apply(zstd::WithEncoderCRC(c->crc != 0));
    if (c->single >= 0) apply(zstd::WithSingleSegment(c->single != 0));
    if (c->full_zero >= 0) apply(zstd::WithZeroFrames(c->full_zero != 0));
    if (c->no_entropy >= 0) apply(zstd::WithNoEntropyCompression(c->no_entropy != 0));
    if (c->all_lit >= 0) apply(zstd::WithAllLitEntropyCompression(c->all_lit != 0));
    if (c->lowmem > 0) apply(zstd::WithLowerEncoderMem(true));
    if (c->concurrency > 0) apply(zstd::WithEncoderConcurrency(Int(K((long long)c->concurrency))));
    if (c->jobs) apply(zstd::WithConcurrentBlocks(true));
    if (c->dict != nullptr && c->dict_len > 0) {
        Slice<byte> d = make_slice<byte>(c->dict_len);
        memcpy((void*)d.p, c->dict, (size_t)c->dict_len);
        if (c->dict_full) apply(zstd::WithEncoderDict(d));
        else apply(zstd::WithEncoderDictRaw(uint32::raw(c->dict_id), d));
    }
}

void run(Call* c) {
    using namespace go;
    rt::Scope scope;  // the call's allocations (the result is copied out below)
    try {
        init_packages();
        zstd::Encoder e;
        apply_options(e, c);
        Slice<byte> src = make_slice<byte>(c->n);
        if (c->n) memcpy((void*)src.p, c->src, (size_t)c->n);
        Slice<byte> out;
        if (c->n_cuts < 0 && c->unit_off != nullptr) {
            auto enc = e.o.encoder();  // ONE encoder for all units: what it keeps from unit to unit must not show in the bytes
            c->out_off[0] = 0;
            long long total = 0;
            for (long long i = 0; i < c->n_units; i++) {  // (frames go straight to the caller's buffer: nothing accumulates here)
                Slice<byte> f = e.encodeAll(enc, src.sl(c->unit_off[i], c->unit_off[i + 1]), Slice<byte>());
                if (total + f.n > c->cap) { snprintf(c->err, sizeof c->err, "output does not fit %lld", (long long)c->cap); c->result = -2; return; }
                if (f.n) memcpy(c->dst + total, f.p, (size_t)f.n);
                total += f.n;
                c->out_off[i + 1] = total;
            }
            c->result = total;
            return;
        } else if (c->n_cuts < 0) {
            auto enc = e.o.encoder();
            out = e.encodeAll(enc, src, Slice<byte>());
        } else {
            // a stream: NewWriter(w, opts...) -> Reset(w); Write(src[a:b]) and Flush() at every cut; Close()
            struct Sink : io::WriterImpl {
                Slice<byte> buf;
                std::tuple<Int, error> Write(Slice<byte> p) override { buf = append_all(buf, p); return std::tuple<Int, error>(len(p), error()); }
            } sink;
            // NewWriter, encoder.go:81-83: job mode is switched off with a dictionary or without concurrency
            if (e.o.concurrentBlocks && (e.o.dict != nil || e.o.concurrent <= K(1LL))) e.o.concurrentBlocks = false;
            e.Reset(io::Writer(&sink));
            long long pos = 0;
            auto check = [&](error er) { if (er != nil) panic(er); };
            for (long long i = 0; i < c->n_cuts; i++) {
                if (c->cuts[i] > c->n) continue;  // (a Flush position behind the end of the input never happens: the tests' convention)
                const long long cut = c->cuts[i];
                if (cut > pos) { auto r = e.Write(src.sl(pos, cut)); check(std::get<1>(r)); pos = cut; }
                check(e.Flush());
            }
            const long long wend = (c->readfrom_at >= 0 && c->readfrom_at <= c->n) ? (c->readfrom_at > pos ? c->readfrom_at : pos) : c->n;
            if (wend > pos) { auto r = e.Write(src.sl(pos, wend)); check(std::get<1>(r)); pos = wend; }
            if (c->readfrom_at >= 0) {
                struct Source : io::ReaderImpl {  // bytes.Reader
                    Slice<byte> buf;
                    long long at = 0;
                    std::tuple<Int, error> Read(Slice<byte> p) override {
                        if (at >= buf.n) return std::tuple<Int, error>(Int(K(0LL)), io::EOF_);
                        const long long k = p.n < buf.n - at ? p.n : buf.n - at;
                        if (k > 0) memcpy((void*)p.p, (const void*)(buf.p + at), (size_t)k);
                        at += k;
                        return std::tuple<Int, error>(Int::raw(k), error());
                    }
                } source;
                source.buf = src.sl(pos, c->n);
                auto r = e.ReadFrom(io::Reader(&source));
                check(std::get<1>(r));
                if (std::get<0>(r).v != c->n - pos) panic_str("ReadFrom returned another byte count than the source holds");
            }
            check(e.Close());
            out = sink.buf;
        }
        if (out.n > c->cap) { snprintf(c->err, sizeof c->err, "output of %lld bytes does not fit %lld", out.n, (long long)c->cap); c->result = -2; return; }
        if (out.n) memcpy(c->dst, out.p, (size_t)out.n);
        c->result = out.n;
    } catch (const go::Panic& p) {
        snprintf(c->err, sizeof c->err, "panic: %s", p.msg.c_str());
        c->result = -1;
    }
}
void* thread_main(void* a) { run((Call*)a); return nullptr; }
}  // namespace

namespace {
struct S2Call { int level; const uint8_t* src; int64_t n; uint8_t* dst; int64_t cap; int64_t result; char err[256]; };
void* s2_thread(void* a) {
    using namespace go;
    S2Call* c = (S2Call*)a;
    rt::Scope scope;
    try {
        { rt::Permanent perm; s2::go_init(); }
        Slice<byte> src = make_slice<byte>(c->n);
        if (c->n) memcpy((void*)src.p, c->src, (size_t)c->n);
        Slice<byte> out;
        switch (c->level) {  // s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter / EncodeBest / EncodeSnappyBest (dst = nil)
            case 0: out = s2::Encode(Slice<byte>(), src); break;
            case 1: out = s2::EncodeBetter(Slice<byte>(), src); break;
            case 2: out = s2::EncodeSnappy(Slice<byte>(), src); break;
            case 3: out = s2::EncodeSnappyBetter(Slice<byte>(), src); break;
            case 4: out = s2::EncodeBest(Slice<byte>(), src); break;
            case 5: out = s2::EncodeSnappyBest(Slice<byte>(), src); break;
            default: c->result = -4; return nullptr;
        }
        if (out.n > c->cap) { c->result = -2; return nullptr; }
        if (out.n) memcpy(c->dst, out.p, (size_t)out.n);
        c->result = out.n;
    } catch (const go::Panic& p) {
        snprintf(c->err, sizeof c->err, "panic: %s", p.msg.c_str());
        c->result = -1;
    }
    return nullptr;
}
}  // namespace

namespace {
// s2.NewWriter(w, WriterConcurrency(1), <options>) as a stream: Write(src[..cut]) + Flush() at every cut, Close(); what w received.
// NewWriter's body (s2/writer.go:34-57) is restated here — its defaults are runtime.GOMAXPROCS(0) and crypto/rand — around the
// translated option functions, Reset, Write / writeSync, Flush, Close / closeIndex, Index.add / appendTo, skippableFrame.
struct S2StreamCall {
    const uint8_t* src; int64_t n; uint8_t* dst; int64_t cap;
    int level, snappy, block_size, add_index, padding, flush_on_write;
    const int64_t* cuts; int64_t n_cuts; int64_t result; char err[256];
    int64_t ebuf_a = -1, ebuf_b = -1;  // >= 0: Write(src[:a]); EncodeBuffer(src[a:b]); Write(src[b:]); Close() instead of the Write / Flush sequence
    int64_t readfrom_at = -1;          // >= 0: behind the Write / Flush sequence, the rest from this offset on through ReadFrom(bytes.NewReader(...))
};
thread_local long long g_s2_ebuf_a = -1, g_s2_ebuf_b = -1, g_s2_readfrom_at = -1;
void* s2_stream_thread(void* a) {
    using namespace go;
    S2StreamCall* c = (S2StreamCall*)a;
    rt::Scope scope;
    try {
        { rt::Permanent perm; s2::go_init(); }
        struct Sink : io::WriterImpl {
            Slice<byte> buf;
            std::tuple<Int, error> Write(Slice<byte> p) override { buf = append_all(buf, p); return std::tuple<Int, error>(len(p), error()); }
        } sink;
        s2::Writer w;
        w.blockSize = Int(s2::defaultBlockSize);
        w.concurrency = Int(K(1LL));
        w.randSrc = rand_::Reader;  // (padding bytes: the io.Reader stub of gort.h leaves the zeroes of make([]byte, n): a zero source)
        w.level = uint8(s2::levelFast);
        auto apply = [&](s2::WriterOption opt) { error er = opt(&w); if (er != nil) panic(er); };
        apply(s2::WriterConcurrency(Int(K(1LL))));
        if (c->level == 1) apply(s2::WriterBetterCompression());
        if (c->level == 2) apply(s2::WriterBestCompression());
        if (c->level == 3) apply(s2::WriterUncompressed());
        if (c->snappy) apply(s2::WriterSnappyCompat());
        if (c->block_size > 0) apply(s2::WriterBlockSize(Int(K((long long)c->block_size))));
        if (c->add_index) apply(s2::WriterAddIndex());
        if (c->padding > 0) apply(s2::WriterPadding(Int(K((long long)c->padding))));
        if (c->flush_on_write) apply(s2::WriterFlushOnWrite());
        w.obufLen = Int(s2::obufHeaderLen) + s2::MaxEncodedLen(w.blockSize);
        w.paramsOK = true;
        w.ibuf = make_slice<byte>(0, w.blockSize.v);
        w.Reset(io::Writer(&sink));
        Slice<byte> src = make_slice<byte>(c->n);
        if (c->n) memcpy((void*)src.p, c->src, (size_t)c->n);
        auto check = [&](error er) { if (er != nil) panic(er); };
        long long pos = 0;
        if (c->ebuf_a >= 0 && c->ebuf_a <= c->ebuf_b && c->ebuf_b <= c->n) {
            if (c->ebuf_a > 0) { auto r = w.Write(src.sl(0, c->ebuf_a)); check(std::get<1>(r)); }
            check(w.EncodeBuffer(src.sl(c->ebuf_a, c->ebuf_b)));
            pos = c->ebuf_b;
        } else {
            for (long long i = 0; i < c->n_cuts; i++) {
                const long long cut = c->cuts[i];
                if (cut > c->n) continue;
                if (cut > pos) { auto r = w.Write(src.sl(pos, cut)); check(std::get<1>(r)); pos = cut; }
                check(w.Flush());
            }
        }
        if (c->readfrom_at >= 0 && c->readfrom_at <= c->n) {
            const long long a = c->readfrom_at > pos ? c->readfrom_at : pos;
            if (a > pos) { auto r = w.Write(src.sl(pos, a)); check(std::get<1>(r)); pos = a; }
            struct Source : io::ReaderImpl {  // bytes.Reader behind a plain io.Reader (no Bytes() method: the read loop, not the EncodeBuffer shortcut)
                Slice<byte> buf;
                long long at = 0;
                std::tuple<Int, error> Read(Slice<byte> p) override {
                    if (at >= buf.n) return std::tuple<Int, error>(Int(K(0LL)), io::EOF_);
                    const long long k = p.n < buf.n - at ? p.n : buf.n - at;
                    if (k > 0) memcpy((void*)p.p, (const void*)(buf.p + at), (size_t)k);
                    at += k;
                    return std::tuple<Int, error>(Int::raw(k), error());
                }
            } source;
            source.buf = src.sl(pos, c->n);
            auto r = w.ReadFrom(io::Reader(&source));
            check(std::get<1>(r));
            pos = c->n;
        }
        if (c->n > pos) { auto r = w.Write(src.sl(pos, c->n)); check(std::get<1>(r)); }
        check(w.Close());
        Slice<byte> out = sink.buf;
        if (out.n > c->cap) { c->result = -2; return nullptr; }
        if (out.n) memcpy(c->dst, out.p, (size_t)out.n);
        c->result = out.n;
    } catch (const go::Panic& p) {
        snprintf(c->err, sizeof c->err, "panic: %s", p.msg.c_str());
        c->result = -1;
    }
    return nullptr;
}
}  // namespace

namespace {
// io.ReadAll(s2.NewReader(bytes.NewReader(src), ReaderMaxBlockSize(n))): the reference's own stream reader (sequential Read: chunk
// types, CRC check, stream identifiers of both kinds, skippable / padding / index chunks, block decode) as the judge of a stream.
// NewReader's body (s2/reader.go:31-51) is restated here around the translated option functions.
struct S2ReadCall { const uint8_t* src; int64_t n; uint8_t* dst; int64_t cap; int max_block; int ignore_crc; int64_t result; char err[256]; };
void* s2_read_thread(void* a) {
    using namespace go;
    S2ReadCall* c = (S2ReadCall*)a;
    rt::Scope scope;
    try {
        { rt::Permanent perm; s2::go_init(); }
        struct Source : io::ReaderImpl {
            Slice<byte> buf;
            long long pos = 0;
            std::tuple<Int, error> Read(Slice<byte> p) override {  // bytes.Reader.Read
                if (pos >= buf.n) return std::tuple<Int, error>(Int(K(0LL)), io::EOF_);
                const long long k = p.n < buf.n - pos ? p.n : buf.n - pos;
                if (k > 0) memcpy((void*)p.p, (const void*)(buf.p + pos), (size_t)k);
                pos += k;
                return std::tuple<Int, error>(Int::raw(k), error());
            }
        } source;
        source.buf = make_slice<byte>(c->n);
        if (c->n) memcpy((void*)source.buf.p, c->src, (size_t)c->n);
        s2::Reader nr;
        nr.r = io::Reader(&source);
        nr.maxBlock = Int(s2::maxBlockSize);
        auto apply = [&](s2::ReaderOption opt) { error er = opt(&nr); if (er != nil) panic(er); };
        if (c->max_block > 0) apply(s2::ReaderMaxBlockSize(Int(K((long long)c->max_block))));
        if (c->ignore_crc) apply(s2::ReaderIgnoreCRC());
        nr.maxBufSize = s2::MaxEncodedLen(nr.maxBlock) + Int(s2::checksumSize);
        nr.buf = make_slice<byte>((s2::MaxEncodedLen(Int(s2::defaultBlockSize)) + Int(s2::checksumSize)).v);
        nr.readHeader = nr.ignoreStreamID;
        nr.paramsOK = true;
        Slice<byte> chunk = make_slice<byte>(1 << 16);
        long long total = 0;
        for (;;) {  // io.ReadAll
            auto r = nr.Read(chunk);
            const long long k = std::get<0>(r).v;
            if (k > 0) {
                if (total + k > c->cap) { c->result = -2; return nullptr; }
                memcpy(c->dst + total, chunk.p, (size_t)k);
                total += k;
            }
            error er = std::get<1>(r);
            if (er == io::EOF_) break;
            if (er != nil) { snprintf(c->err, sizeof c->err, "%s", er.e->msg.c_str()); c->result = -5; return nullptr; }
        }
        c->result = total;
    } catch (const go::Panic& p) {
        snprintf(c->err, sizeof c->err, "panic: %s", p.msg.c_str());
        c->result = -1;
    }
    return nullptr;
}
}  // namespace

namespace {
// the translated encoders hold their tables by value, like the Go structs do on Go's heap: a thread with a large stack
long long run_on_big_stack(Call* c, char* err, int err_cap) {
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, thread_main, c) != 0) return -3;
    pthread_join(th, nullptr);
    pthread_attr_destroy(&at);
    if (err && err_cap > 0) { strncpy(err, c->err, (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    return c->result;
}
}  // namespace

namespace {
struct DecCall {
    const uint8_t* src; int64_t n; uint8_t* dst; int64_t cap; int64_t result; char err[256];
    const uint8_t* dict = nullptr; int64_t dict_len = 0; uint32_t dict_id = 0; int dict_full = 0;
};
// zstd.NewReader(nil, <WithDecoderDicts(d) | WithDecoderDictRaw(id, d)>).DecodeAll(src, nil): the reference's own decoder as the
// judge of a frame's validity (decoderOptions.dicts, a map in the reference, is a list here: ref_go/manifest.py).
void* dec_thread(void* a) {
    using namespace go;
    DecCall* c = (DecCall*)a;
    rt::Scope scope;
    try {
        init_packages();
        zstd::Decoder d;
        d.o.setDefault();  // NewReader's first statement (decoder.go:91)
        if (c->dict != nullptr && c->dict_len > 0) {  // ... then its option loop
            Slice<byte> dd = make_slice<byte>(c->dict_len);
            memcpy((void*)dd.p, c->dict, (size_t)c->dict_len);
            zstd::DOption opt = c->dict_full ? zstd::WithDecoderDicts(Slice<Slice<byte>>{dd}) : zstd::WithDecoderDictRaw(uint32::raw(c->dict_id), dd);
            error oe = opt(&d.o);
            if (oe != nil) { snprintf(c->err, sizeof c->err, "%s", oe.e->msg.c_str()); c->result = -6; return nullptr; }
        }
        Slice<byte> src = make_slice<byte>(c->n);
        if (c->n) memcpy((void*)src.p, c->src, (size_t)c->n);
        auto r = d.DecodeAll(src, Slice<byte>());
        error er = std::get<1>(r);
        if (er != nil) { snprintf(c->err, sizeof c->err, "%s", er.e->msg.c_str()); c->result = -5; return nullptr; }
        Slice<byte> out = std::get<0>(r);
        if (out.n > c->cap) { c->result = -2; return nullptr; }
        if (out.n) memcpy(c->dst, out.p, (size_t)out.n);
        c->result = out.n;
    } catch (const go::Panic& p) {
        snprintf(c->err, sizeof c->err, "panic: %s", p.msg.c_str());
        c->result = -1;
    }
    return nullptr;
}
}  // namespace

#ifdef GOREF_AMD64
// The assembly addresses the fields of these structures by the offsets Go's compiler gives them.  The translation keeps Go's field
// order and types (gort.h: I<T> is T, Slice is base / len / cap, Array is the elements), so the C++ compiler lays them out the same
// way; the sizes below are Go's (zstd/seqdec.go:59-98, seqdec_asm.go:13-58, fse_decoder_asm.go:14-25, bitreader.go:17-22,
// huff0/decompress_asm.go:16-25,131-138, huff0/bitreader.go:126-131) — a drift fails the build, not a test.
static_assert(sizeof(zstd::seqVals) == 24 && sizeof(zstd::decSymbol) == 8, "seqVals / decSymbol");
static_assert(sizeof(zstd::bitReader) == 48, "zstd bitReader: in []byte, value uint64, cursor int, bitsRead uint8");
static_assert(sizeof(zstd::fseState) == 32 && sizeof(zstd::sequenceDec) == 48, "fseState {dt []decSymbol; state decSymbol}, sequenceDec {fse, state, repeat}");
static_assert(offsetof(zstd::sequenceDecs, offsets) == 48 && offsetof(zstd::sequenceDecs, matchLengths) == 96 && offsetof(zstd::sequenceDecs, prevOffset) == 144, "sequenceDecs");
static_assert(sizeof(zstd::decodeAsmContext) == 136 && sizeof(zstd::executeAsmContext) == 128 && sizeof(zstd::decodeSyncAsmContext) == 232, "asm contexts");
static_assert(sizeof(zstd::buildDtableAsmContext) == 40, "buildDtableAsmContext");
static_assert(offsetof(zstd::fseDecoder, symbolLen) == 4096 && offsetof(zstd::fseDecoder, stateTable) == 4100 && offsetof(zstd::fseDecoder, norm) == 4612, "fseDecoder: dt [512]decSymbol, symbolLen, actualTableLog, maxBits, stateTable [256]uint16, norm [256]int16");
static_assert(sizeof(huff0::decompress4xContext) == 56 && sizeof(huff0::decompress1xContext) == 48 && sizeof(huff0::bitReaderShifted) == 48, "huff0 asm contexts");
#endif

extern "C" {
// amd64 flavour: 0 = the dispatch helpers take the routines without BMI1 / BMI2, -1 = what the host's CPU has (a no-op in the
// portable flavour, which has no such routines)
void goref_force_bmi(int mode) { cpuinfo::force() = mode; }
long long goref_zstd_decode_all_dict(const uint8_t* src, long long n, uint8_t* dst, long long cap, const uint8_t* dict, long long dict_len,
                                     unsigned dict_id, char* err, int err_cap);
// DecodeAll(src, nil) of zstd.NewReader(nil): >= 0 the decoded length, -5 the decoder's error (text in err)
long long goref_zstd_decode_all(const uint8_t* src, long long n, uint8_t* dst, long long cap, char* err, int err_cap) {
    return goref_zstd_decode_all_dict(src, n, dst, cap, nullptr, 0, 0, err, err_cap);
}
// ... with a dictionary registered: dict_id 0xFFFFFFFF = a full-format dictionary (WithDecoderDicts), else WithDecoderDictRaw(dict_id, dict)
long long goref_zstd_decode_all_dict(const uint8_t* src, long long n, uint8_t* dst, long long cap, const uint8_t* dict, long long dict_len,
                                     unsigned dict_id, char* err, int err_cap) {
    DecCall c{src, n, dst, cap, 0, {0}};
    c.dict = dict; c.dict_len = dict_len; c.dict_id = dict_id;
    if (dict_id == 0xFFFFFFFFu) { c.dict_full = 1; c.dict_id = 0; }
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, dec_thread, &c) != 0) return -3;
    pthread_join(th, nullptr);
    pthread_attr_destroy(&at);
    if (err && err_cap > 0) { strncpy(err, c.err, (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    return c.result;
}
// s2.Encode* (nil, src) of a build WITHOUT the assembly (encode_go.go: the portable Go encoders, what arm64 / noasm builds run)
long long goref_s2_encode(int level, const uint8_t* src, long long n, uint8_t* dst, long long cap, char* err, int err_cap) {
    S2Call c{level, src, n, dst, cap, 0, {0}};
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, s2_thread, &c) != 0) return -3;
    pthread_join(th, nullptr);
    pthread_attr_destroy(&at);
    if (err && err_cap > 0) { strncpy(err, c.err, (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    return c.result;
}
// The framed S2 stream of s2.NewWriter(w, WriterConcurrency(1), ...): level 0 fast / 1 better / 2 best / 3 uncompressed; block_size 0 =
// the default (1 MiB); padding 0 = none (padding bytes are zeroes: a zero WriterPaddingSrc); Flush() after the input offsets in cuts
long long goref_s2_stream(const uint8_t* src, long long n, uint8_t* dst, long long cap, int level, int snappy, int block_size, int add_index,
                          int padding, int flush_on_write, const long long* cuts, long long n_cuts, char* err, int err_cap) {
    S2StreamCall c{src, n, dst, cap, level, snappy, block_size, add_index, padding, flush_on_write, (const int64_t*)cuts, n_cuts < 0 ? 0 : n_cuts, 0, {0}};
    c.ebuf_a = g_s2_ebuf_a; c.ebuf_b = g_s2_ebuf_b; c.readfrom_at = g_s2_readfrom_at;
    g_s2_ebuf_a = g_s2_ebuf_b = g_s2_readfrom_at = -1;
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, s2_stream_thread, &c) != 0) return -3;
    pthread_join(th, nullptr);
    pthread_attr_destroy(&at);
    if (err && err_cap > 0) { strncpy(err, c.err, (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    return c.result;
}
// io.ReadAll(s2.NewReader(src)): >= 0 the decoded length, -5 the reader's error (text in err); max_block 0 = the default limit (4 MiB)
long long goref_s2_read_stream(const uint8_t* src, long long n, uint8_t* dst, long long cap, int max_block, int ignore_crc, char* err, int err_cap) {
    S2ReadCall c{src, n, dst, cap, max_block, ignore_crc, 0, {0}};
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, s2_read_thread, &c) != 0) return -3;
    pthread_join(th, nullptr);
    pthread_attr_destroy(&at);
    if (err && err_cap > 0) { strncpy(err, c.err, (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    return c.result;
}
// the NEXT goref_s2_stream call does Write(src[:a]); EncodeBuffer(src[a:b]); Write(src[b:]); Close() (one-shot)
void goref_s2_next_stream_encode_buffer(long long a, long long b) { g_s2_ebuf_a = a; g_s2_ebuf_b = b; }
// ... or feeds its input from this offset on through Writer.ReadFrom (one-shot)
void goref_s2_next_stream_readfrom(long long at) { g_s2_readfrom_at = at; }
// Encoder.MaxEncodedSize(size) of zstd.NewWriter(nil, WithEncoderLevel(level), WithWindowSize(window)) (encoder.go) and s2.MaxEncodedLen(n)
// (s2/encode.go): the size bounds of the boundary, from the reference's own code (pure arithmetic: no thread, no big stack)
long long goref_zstd_max_encoded_size(long long size, int level, int window) {
    using namespace go;
    rt::Scope scope;
    try {
        init_packages();
        zstd::Encoder e;
        Call c{nullptr, 0, nullptr, 0, level, window, -1, -1, -1, -1, -1, 0, nullptr, 0, 0, 0, {0}};
        apply_options(e, &c);
        return e.MaxEncodedSize(Int::raw(size)).v;
    } catch (const go::Panic&) { return -1; }
}
// encoderOptions.jobSize() / overlapSize() (encoder_options.go:356-371) of the options a level and a window give; calcSkippableFrame of the
// zstd package (frameenc.go:100-116) and of s2 (writer.go:858-874): which: 0 job size, 1 overlap size; pkg: 0 zstd, 1 s2
long long goref_zstd_job_geometry(int which, int level, int window) {
    using namespace go;
    rt::Scope scope;
    try {
        init_packages();
        zstd::Encoder e;
        Call c{nullptr, 0, nullptr, 0, level, window, -1, -1, -1, -1, -1, 0, nullptr, 0, 0, 0, {0}};
        apply_options(e, &c);
        return which == 0 ? e.o.jobSize().v : e.o.overlapSize().v;
    } catch (const go::Panic&) { return -1; }
}
long long goref_calc_skippable_frame(int pkg, long long written, long long want_multiple) {
    using namespace go;
    rt::Scope scope;
    try {
        init_packages();
        { rt::Permanent perm; s2::go_init(); }
        return pkg == 0 ? zstd::calcSkippableFrame(int64::raw(written), int64::raw(want_multiple)).v : s2::calcSkippableFrame(int64::raw(written), int64::raw(want_multiple)).v;
    } catch (const go::Panic&) { return -1; }
}
// s2's block-format emitters as the translated reference runs them (encode_go.go: emitLiteral / emitCopy / emitRepeat /
// emitCopyNoRepeat), for the known-answer strings of the reference's own tests (s2/s2_test.go:827-942).  kind 0: emitLiteral(dst,
// lit[:a]); 1: emitCopy(dst, a, b); 2: emitRepeat(dst, a, b); 3: emitCopyNoRepeat(dst, a, b).  Returns the bytes written.
long long goref_s2_emit(int kind, const uint8_t* lit, long long a, long long b, uint8_t* dst, long long cap) {
    using namespace go;
    rt::Scope scope;
    try {
        { rt::Permanent perm; s2::go_init(); }
        Slice<byte> d = make_slice<byte>(cap);
        Int n;
        if (kind == 0) {
            Slice<byte> l = make_slice<byte>(a);
            if (a) memcpy((void*)l.p, lit, (size_t)a);
            n = s2::emitLiteral(d, l);
        } else if (kind == 1) n = s2::emitCopy(d, Int::raw(a), Int::raw(b));
        else if (kind == 2) n = s2::emitRepeat(d, Int::raw(a), Int::raw(b));
        else if (kind == 3) n = s2::emitCopyNoRepeat(d, Int::raw(a), Int::raw(b));
        else return -4;
        if (n.v > cap) return -2;
        if (n.v) memcpy(dst, d.p, (size_t)n.v);
        return n.v;
    } catch (const go::Panic&) { return -1; }
}
// s2.Decode(nil, src) of the reference (decode.go:53 -> s2Decode, decode_other.go): one block, Snappy blocks included.
// >= 0 the decoded length, -5 the decoder's error
long long goref_s2_decode(const uint8_t* src, long long n, uint8_t* dst, long long cap) {
    using namespace go;
    rt::Scope scope;
    try {
        { rt::Permanent perm; s2::go_init(); }
        Slice<byte> in = make_slice<byte>(n);
        if (n) memcpy((void*)in.p, src, (size_t)n);
        auto r = s2::Decode(Slice<byte>(), in);
        if (std::get<1>(r) != nil) return -5;
        Slice<byte> out = std::get<0>(r);
        if (out.n > cap) return -2;
        if (out.n) memcpy(dst, out.p, (size_t)out.n);
        return out.n;
    } catch (const go::Panic&) { return -1; }
}
long long goref_s2_max_encoded_len(long long n) {
    using namespace go;
    rt::Scope scope;
    try {
        { rt::Permanent perm; s2::go_init(); }
        return s2::MaxEncodedLen(Int::raw(n)).v;
    } catch (const go::Panic&) { return -2; }
}
// EncodeAll(src, nil) of zstd.NewWriter(nil, <options>); options < 0 (or 0 for level / window / dict): the reference's defaults.
long long goref_zstd_encode_all(const uint8_t* src, long long n, uint8_t* dst, long long cap, int level, int window, int crc, int single,
                                int full_zero, int no_entropy, int all_lit, int lowmem, const uint8_t* dict, long long dict_len,
                                unsigned dict_id, char* err, int err_cap) {
    Call c{src, n, dst, cap, level, window, crc, single, full_zero, no_entropy, all_lit, lowmem, dict, dict_len, dict_id, 0, {0}};
    if (dict_id == 0xFFFFFFFFu) { c.dict_full = 1; c.dict_id = 0; }  // (a full-format dictionary carries its own id: WithEncoderDict)
    return run_on_big_stack(&c, err, err_cap);
}
// N x EncodeAll on ONE zstd.Encoder (one pooled encoder, re-used from call to call): the frames back to back, out_off[n_units + 1]
long long goref_zstd_encode_all_reuse(const uint8_t* src, const long long* unit_off, long long n_units, uint8_t* dst, long long cap, long long* out_off,
                                      int level, int window, const uint8_t* dict, long long dict_len, unsigned dict_id, char* err, int err_cap) {
    Call c{src, unit_off[n_units], dst, cap, level, window, -1, -1, -1, -1, -1, 0, dict, dict_len, dict_id, 0, {0}};
    if (dict_id == 0xFFFFFFFFu) { c.dict_full = 1; c.dict_id = 0; }
    c.unit_off = (const int64_t*)unit_off;
    c.n_units = n_units;
    c.out_off = (int64_t*)out_off;
    return run_on_big_stack(&c, err, err_cap);
}
// zstd.NewWriter(w, <options>) as a STREAM: Write(src[..cut]) + Flush() at every cut, then Close(); returns what w received.
// concurrency 1: the synchronous form of nextBlock (encoder.go:361-386), else the asynchronous one (its goroutines run where started).
long long goref_zstd_encode_stream(const uint8_t* src, long long n, uint8_t* dst, long long cap, int level, int window, int crc, int no_entropy,
                                   int all_lit, int lowmem, int concurrency, const uint8_t* dict, long long dict_len, unsigned dict_id,
                                   const long long* cuts, long long n_cuts, char* err, int err_cap) {
    Call c{src, n, dst, cap, level, window, crc, -1, -1, no_entropy, all_lit, lowmem, dict, dict_len, dict_id, 0, {0}};
    c.concurrency = concurrency & 0xFFFF;
    c.jobs = (concurrency >> 16) & 1;  // (bit 16 of the argument: WithConcurrentBlocks(true))
    if (dict_id == 0xFFFFFFFFu) { c.dict_full = 1; c.dict_id = 0; }
    c.cuts = (const int64_t*)cuts;
    c.n_cuts = n_cuts < 0 ? 0 : n_cuts;
    c.readfrom_at = g_readfrom_at;
    g_readfrom_at = -1;
    return run_on_big_stack(&c, err, err_cap);
}
// the NEXT goref_zstd_encode_stream call feeds its input from this offset on through Encoder.ReadFrom (one-shot; -1: Write only)
void goref_zstd_next_stream_readfrom(long long at) { g_readfrom_at = at; }
}
