"""What go2cpp.py translates: the reference's Go files of the zstd encode path, per package in dependency order; declarations
left out (decoder halves, goroutine variants, String() methods of debug output) and the few source patches, each with its reason."""

CFG = {
    "packages": [
        ("fse", ["fse/fse.go", "fse/bitwriter.go", "fse/bytereader.go", "fse/bitreader.go", "fse/compress.go", "fse/decompress.go"]),
        ("huff0", ["huff0/huff0.go", "huff0/bitwriter.go", "huff0/bitreader.go", "huff0/compress.go", "huff0/decompress.go", "huff0/decompress_generic.go"]),
        ("xxhash", ["zstd/internal/xxhash/xxhash.go", "zstd/internal/xxhash/xxhash_other.go"]),
        ("compress", ["compressible.go"]),
        ("s2", ["s2/s2.go", "s2/decode.go", "s2/hashtable_pool.go", "s2/dict.go", "s2/encode.go", "s2/encode_go.go", "s2/encode_all.go", "s2/encode_better.go", "s2/encode_best.go", "s2/index.go", "s2/writer.go", "s2/decode_other.go", "s2/reader.go"]),
        ("zstd", ["zstd/zstd.go", "zstd/hash.go", "zstd/matchlen_generic.go", "zstd/bitwriter.go", "zstd/seqenc.go", "zstd/fse_encoder.go",
                  "zstd/fse_predefined.go", "zstd/frameenc.go", "zstd/blockenc.go", "zstd/bytereader.go", "zstd/enc_base.go", "zstd/enc_fast.go", "zstd/enc_dfast.go",
                  "zstd/enc_better.go", "zstd/enc_best.go", "zstd/seqdec.go", "zstd/seqdec_generic.go", "zstd/dict.go", "zstd/bitreader.go", "zstd/bytebuf.go", "zstd/history.go",
                  "zstd/blockdec.go", "zstd/framedec.go", "zstd/fse_decoder.go",
                  "zstd/fse_decoder_generic.go", "zstd/encoder_options.go", "zstd/enc_jobs.go", "zstd/encoder.go", "zstd/decoder_options.go", "zstd/decoder.go"]),
    ],
    # path -> names of top-level declarations (or Type.method) that are not translated
    "skip": {
        "fse/fse.go": set(),
        "huff0/compress.go": {"Scratch.compress4Xp"},  # the goroutine variant, only reachable from an `if false` block
        "huff0/decompress.go": {"Scratch.matches"},      # a debugging aid (fmt.Fprintf to an io.Writer)
        "zstd/bytebuf.go": {"readerWrapper", "readerWrapper.*"},   # the io.Reader form of the decoder's input (DecodeAll reads a []byte: byteBuf)
        "zstd/fse_decoder.go": {"fseDecoder.mustReadFrom"},        # loads a table dump with encoding/binary.Read (a development aid)
        "s2/s2.go": {"_", "byter"},                                             # an interface assertion
        "s2/encode.go": {"EstimateBlockSize", "estblockPool", "ConcatBlocks"},  # size estimation (calcBlockSize, not an encoder), block concatenation
        "s2/dict.go": {"Dict.Decode", "MakeDict", "MakeDictManual"},            # the decoder half; dictionary construction by search
        "s2/encode_go.go": {"calcBlockSize", "calcBlockSizeSmall", "cvtLZ4BlockAsm", "cvtLZ4BlockSnappyAsm", "cvtLZ4sBlockAsm", "cvtLZ4sBlockSnappyAsm"},
        "zstd/zstd.go": {"_", "byter"},                                   # an interface assertion on bytes.Buffer (decoder input)
        "zstd/encoder_options.go": {"EncoderLevelFromString", "EncoderLevel.String"},  # strings package
        "zstd/enc_jobs.go": {"Encoder.jobWorker", "Encoder.jobFlusher"},              # the worker and flusher goroutines (see the patches)
    },
    # path -> the ONLY declarations taken from that file (the rest of it is the decoder / the streaming writer)
    "only": {
        "s2/decode.go": {"ErrCorrupt", "ErrCRC", "ErrTooLarge", "ErrUnsupported", "DecodedLen", "decodedLen", "decodeErrCodeCorrupt", "Decode"},   # error values; Decode (blocks)
        # the stream reader as the judge of a framed stream's validity: sequential Read (not DecodeConcurrent's goroutines, not the seeker)
        "s2/reader.go": {"Reader", "ReaderOption", "ReaderMaxBlockSize", "ReaderIgnoreCRC", "Reader.ensureBufferSize", "Reader.Reset", "Reader.readFull",
                         "Reader.skippable", "Reader.Read"},
        # the stream writer in its synchronous form (WriterConcurrency(1): Write / Flush / Close run writeSync in the caller — the
        # framing, the index and the padding do not depend on the concurrency) and the index it appends
        "s2/writer.go": {"Writer", "Writer.err", "Writer.Reset", "Writer.Write", "Writer.EncodeBuffer", "Writer.ReadFrom", "Writer.writeFull", "Writer.encodeBlock", "Writer.write", "Writer.writeSync",
                         "Writer.AsyncFlush", "Writer.Flush", "Writer.Close", "Writer.CloseIndex", "Writer.closeIndex", "calcSkippableFrame", "skippableFrame",
                         "errClosed", "WriterOption", "WriterConcurrency", "WriterAddIndex", "WriterBetterCompression", "WriterBestCompression",
                         "WriterUncompressed", "WriterBlockSize", "WriterPadding", "WriterSnappyCompat", "WriterFlushOnWrite",
                         "levelUncompressed", "levelFast", "levelBetter", "levelBest"},
        "s2/index.go": {"S2IndexHeader", "S2IndexTrailer", "maxIndexEntries", "minIndexDist", "indexInfo", "Index", "Index.reset", "Index.allocInfos", "Index.add",
                        "Index.reduce", "Index.appendTo"},
        "zstd/decoder.go": {"Decoder", "Decoder.DecodeAll", "Decoder.setDict"},   # the stateless DecodeAll; not the streaming reader (goroutines, channels)
        "zstd/decoder_options.go": {"DOption", "decoderOptions", "decoderOptions.setDefault", "WithDecoderDicts", "WithDecoderDictRaw"},
        "zstd/dict.go": {"dict", "dict.*", "dictMagic", "dictMaxLength", "loadDict"},   # (not InspectDictionary / BuildDict)
        # EncodeAll and the streaming writer in both modes (blocks; WithConcurrentBlocks jobs), ReadFrom included — not the
        # goroutine pool behind the public EncodeAll (the driver hands encodeAll an encoder)
        "zstd/encoder.go": {"Encoder", "encoder", "encoderState", "Encoder.encodeAll", "Encoder.MaxEncodedSize", "Encoder.Reset", "Encoder.Write",
                            "Encoder.writeBlocks", "Encoder.writeJobs", "Encoder.nextBlock", "Encoder.Flush", "Encoder.flushJobs", "Encoder.Close",
                            "Encoder.closeJobs", "Encoder.ReadFrom", "Encoder.readFromJobs"},
        # constants and the block / literals type enumerations the encoder shares with the decoder
        # initPredefined builds the predefined DECODER tables first and copies their normalised counts into the encoders
    },
    "drop_fields": {
        "s2.Writer": {"output", "buffers", "writerWg", "bufferCB", "customEnc"},
        "s2.Reader": {"skippableCB"},                      # per-id callbacks for skippable chunks (an array of funcs taking an io.Reader)   # the concurrent form's channel, buffer pool and wait group; callbacks
        "zstd.Decoder": {"decoders", "current", "syncStream", "frame", "streamWg"},
        "zstd.Encoder": {"encoders", "init"},              # the pool of encoders EncodeAll draws from (the driver hands it one)
        "zstd.encJob": {"done"},                           # job mode's worker plumbing: see the patches of zstd/enc_jobs.go
        "zstd.jobState": {"jobCh", "resultCh", "cond", "workerWg", "flusherWg", "inputPool", "outputPool", "overlapPool"},
},
    # path -> [(regular expression, replacement, why)]: source patches applied before parsing
    "patches": {
        "huff0/decompress.go": [
            (r"\tbuf, ok := d\.bufs\.Get\(\)\.\(\*\[4\]\[256\]byte\)\n\tif ok \{\n\t\treturn buf\n\t\}\n", "", "sync.Pool is an allocation cache: always the fresh buffer of the next line"),
        ],
        "zstd/blockdec.go": [
            (r"\thuffDecoderPool = sync\.Pool\{New: func\(\) any \{.*?\}\}\n", "\thuffDecoderPool sync.Pool\n", "allocation caches: Get is a fresh object (next patches), Put a no-op"),
            (r"\tfseDecoderPool = sync\.Pool\{New: func\(\) any \{.*?\}\}\n", "\tfseDecoderPool sync.Pool\n", "likewise"),
            (r"huffDecoderPool\.Get\(\)\.\(\*huff0\.Scratch\)", "&huff0.Scratch{}", "what the pool's New returns"),
            (r"fseDecoderPool\.Get\(\)\.\(\*fseDecoder\)", "&fseDecoder{}", "what the pool's New returns"),
        ],
        "zstd/decoder.go": [
            (r"\tif d\.decoders == nil \{\n\t\treturn dst, ErrDecoderClosed\n\t\}\n", "", "DecodeAll takes a block decoder from the pool (a channel): here a fresh one"),
            (r"block := <-d\.decoders\n\tframe := block\.localFrame", "block := newBlockDec(d.o.lowMem)\n\tblock.localFrame = newFrameDec(d.o)\n\tframe := block.localFrame", "what Decoder's pool holds (decoder.go:105-115)"),
            (r"\t\td\.decoders <- block\n", "", "nothing to give back"),
            (r"\tdict, ok := d\.o\.dicts\[frame\.DictionaryID\]\n",
             "\tvar dd *dict\n\tok := false\n\tfor i := len(d.o.dicts) - 1; i >= 0; i-- {\n\t\tif d.o.dicts[i].id == frame.DictionaryID {\n\t\t\tdd = d.o.dicts[i]\n\t\t\tok = true\n\t\t\tbreak\n\t\t}\n\t}\n",
             "decoderOptions.dicts is a map[uint32]*dict (the translation has no maps): kept as the list of registered dictionaries, looked up from the end (the last one registered under an id wins, like the map's overwrite)"),
            (r"frame\.history\.setDict\(dict\)", "frame.history.setDict(dd)", "the lookup's variable (previous patch)"),
        ],
        "zstd/decoder_options.go": [
            (r"\tdicts           map\[uint32\]\*dict\n", "\tdicts           []*dict\n", "see zstd/decoder.go: a list instead of a map"),
            (r"\t\tif o\.dicts == nil \{\n\t\t\to\.dicts = make\(map\[uint32\]\*dict\)\n\t\t\}\n", "", "nothing to allocate for a list"),
            (r"o\.dicts\[d\.id\] = d\n", "o.dicts = append(o.dicts, d)\n", "register = append"),
            (r"o\.dicts\[id\] = (&dict\{[^\n]*\})\n", r"o.dicts = append(o.dicts, \1)\n", "register = append"),
        ],
        # WithConcurrentBlocks (enc_jobs.go).  What decides the bytes is translated as it stands: the job cutting (writeJobs, dispatchJob,
        # flushJobs, closeJobs), the per-job encode (compressJob: ResetPrefix / Reset, then block by block) and the frame assembly.  What
        # is replaced is the plumbing that runs compressJob on worker goroutines and writes the results in order (two channels, a
        # condition variable, three buffer pools): here every job is compressed and written where it is dispatched — the one schedule a
        # translation without a scheduler can offer, and a valid one (jobs are independent; the flusher writes them in dispatch order).
        "zstd/enc_jobs.go": [
            (r"\tjs\.resultCh <- job\n\tjs\.jobCh <- job\n",
             "\te.compressJob(e.o.encoder(), job)\n\tif job.err != nil {\n\t\treturn job.err\n\t}\n\tif len(job.output) > 0 {\n\t\t_, err := s.w.Write(job.output)\n"
             "\t\tif err != nil {\n\t\t\treturn err\n\t\t}\n\t\ts.nWritten += int64(len(job.output))\n\t}\n\tjs.flushedSeq++\n",
             "dispatch = compress + write, in place of the two channel sends (jobWorker's and jobFlusher's loop bodies, enc_jobs.go:69-77, 182-222); "
             "the worker's encoder comes from e.o.encoder() like the pool's (initialize, encoder.go:89-97) — not s.encoder, whose digest carries the frame's checksum"),
            (r"\t\tdone:   make\(chan struct\{\}\),\n", "", "no worker to wait for"),
            (r"func \(e \*Encoder\) startJobWorkers\(\) \{.*?\n\}\n", "func (e *Encoder) startJobWorkers() {\n\te.state.jobs.started = true\n}\n", "no workers to start"),
            (r"func \(e \*Encoder\) shutdownJobWorkers\(\) \{.*?\n\}\n", "func (e *Encoder) shutdownJobWorkers() {\n\te.state.jobs.started = false\n}\n", "... nor to stop"),
            (r"func \(e \*Encoder\) waitAllJobs\(\) \{.*?\n\}\n", "func (e *Encoder) waitAllJobs() {\n}\n", "every dispatched job is already written"),
            (r"\tif v := js\.\w+Pool\.Get\(\); v != nil \{.*?\n\t\}\n", "", "the three sync.Pools are allocation caches: always allocate"),
            (r"\t\tjs\.\w+Pool\.Put\(&b\)\n", "", "likewise"),
        ],
        # s2.Writer: everything behind `if w.concurrency == 1 { ... return }` is the concurrent form (one goroutine per block, results
        # written in order through a channel of channels): cut, the driver only builds writers with WriterConcurrency(1)
        "s2/writer.go": [
            (r"\t// Close previous writer, if any\.\n\tif w\.output != nil \{.*?\n\t\}\n", "", "no writer goroutine to close"),
            (r"(\tif w\.concurrency == 1 \{\n\t\treturn\n\t\}\n).*?\n\}\n", r"\1}\n", "Reset: the writer goroutine of the concurrent form"),
            (r"(\tif w\.concurrency == 1 \{\n\t\treturn w\.writeSync\(p\)\n\t\}\n).*?\n\}\n", r'\1\tpanic("concurrent form not translated")\n}\n', "write"),
            (r"(\tif w\.concurrency == 1 \{\n\t\t_, err := w\.writeSync\(buf\)\n)\t\tif w\.bufferCB != nil \{\n\t\t\tw\.bufferCB\(buf\)\n\t\t\}\n(\t\treturn err\n\t\}\n).*?\n\}\n",
             r'\1\2\tpanic("concurrent form not translated")\n}\n', "EncodeBuffer"),
            (r"\tif w\.customEnc != nil \{.*?\n\t\}\n(\tif w\.snappy \{)", r"\1", "encodeBlock: no custom encoder here"),
            (r"\tif br, ok := r\.\(byter\); ok \{.*?\n\t\}\n(\tfor \{\n\t\tinbuf := )", r"\1", "ReadFrom: the source is a plain io.Reader (no Bytes() shortcut)"),
            (r"inbuf := w\.buffers\.Get\(\)\.\(\[\]byte\)\[:w\.blockSize\+obufHeaderLen\]", "inbuf := make([]byte, w.blockSize+obufHeaderLen, w.obufLen)", "sync.Pool is an allocation cache"),
            (r"\t\t\tif cap\(inbuf\) >= w\.obufLen \{\n\t\t\t\tw\.buffers\.Put\(inbuf\)\n\t\t\t\}\n", "", "likewise"),
            (r"(\t\t_, err := w\.writeSync\(inbuf\[obufHeaderLen:\]\)\n)\t\tif cap\(inbuf\) >= w\.obufLen \{\n\t\t\tw\.buffers\.Put\(inbuf\)\n\t\t\}\n(\t\treturn err\n\t\}\n).*?\n\}\n",
             r'\1\2\tpanic("concurrent form not translated")\n}\n', "writeFull"),
            (r"obuf := w\.buffers\.Get\(\)\.\(\[\]byte\)\[:w\.obufLen\]", "obuf := make([]byte, w.obufLen)", "sync.Pool is an allocation cache"),
            (r"\t\tw\.buffers\.Put\(obuf\)\n", "", "likewise"),
            (r"(\tif err := w\.AsyncFlush\(\); err != nil \{\n\t\treturn err\n\t\}\n)\tif w\.output == nil \{\n\t\treturn w\.err\(nil\)\n\t\}\n.*?\n\}\n", r"\1\treturn w.err(nil)\n}\n",
             "Flush: nothing queued anywhere in the synchronous form"),
            (r"\tif w\.output != nil \{\n\t\tclose\(w\.output\)\n\t\tw\.writerWg\.Wait\(\)\n\t\tw\.output = nil\n\t\}\n", "", "closeIndex: no writer goroutine"),
            (r"tmp = w\.buffers\.Get\(\)\.\(\[\]byte\)\[:0\]\n\t\t\t\tdefer w\.buffers\.Put\(tmp\)\n", "tmp = make([]byte, 0, w.obufLen)\n", "sync.Pool is an allocation cache"),
        ],
        "s2/reader.go": [
            (r"\tif fn := r\.skippableCB\[id-0x80\]; fn != nil \{.*?\n(\tfor n > 0 \{\n)", r"\1",
             "skippable: no callbacks registered, and the source is a plain io.Reader (not an io.ReadSeeker): what is left is the read-and-discard loop"),
        ],
        "s2/index.go": [
            (r"struct \{\n\t+compressedOffset   int64\n\t+uncompressedOffset int64\n\t+\}", "indexInfo",
             "the element type of Index.info is an anonymous struct: given a name (the translation names every struct type)"),
            (r"// Index represents an S2/Snappy index\.\n", "type indexInfo struct {\n\tcompressedOffset   int64\n\tuncompressedOffset int64\n}\n\n// Index represents an S2/Snappy index.\n",
             "... declared here"),
        ],
        "s2/hashtable_pool.go": [
            (r"= sync\.Pool\{New: func\(\) any \{ return &\w+\{\} \}\}", " sync.Pool",
             "sync.Pool is an allocation cache: its New hook is not needed when Get is replaced by a fresh table (next patch)"),
            (r"\w+\.Get\(\)\.\(\*(\w+)\)", r"&\1{}",
             "a pooled table is zeroed right after Get(): a fresh zero table is the same value"),
        ],
    },
}


def flavour(name):
    """The amd64 build of the reference (what `go build` gives an x86-64 user): the Go halves of its assembly routines in place of the
    portable ones — zstd's sequence decoder and executor (seqdec_amd64.s), buildDtable (fse_decoder_amd64.s), matchLen
    (matchlen_amd64.s, used by all four encoders), huff0's 1X / 4X decoding loops (decompress_amd64.s).  The assembly itself is
    re-spelt for the GNU assembler by oracle/ref_s2asm/plan9_to_gas.py and linked in; go2cpp.py writes the ABI0 call thunks."""
    if name != "amd64":
        raise SystemExit("manifest: unknown flavour %r" % name)
    import copy
    cfg = copy.deepcopy(CFG)
    swap = {"huff0/decompress_generic.go": ["huff0/decompress_amd64.go", "huff0/decompress_asm.go"],
            "zstd/seqdec_generic.go": ["zstd/seqdec_asm.go", "zstd/seqdec_amd64.go"],
            "zstd/fse_decoder_generic.go": ["zstd/fse_decoder_asm.go"],
            "zstd/matchlen_generic.go": ["zstd/matchlen_amd64.go"]}
    pk = []
    for pname, files in cfg["packages"]:
        out = []
        for f in files:
            out += swap.pop(f, [f])
        pk.append((pname, out))
    if swap:
        raise SystemExit("manifest: files to swap not in the manifest: %r" % sorted(swap))
    cfg["packages"] = pk
    return cfg
