"""What go2cpp.py translates: the reference's Go files of the zstd encode path, per package in dependency order; declarations
left out (decoder halves, goroutine variants, String() methods of debug output) and the few source patches, each with its reason."""

CFG = {
    "packages": [
        ("fse", ["fse/fse.go", "fse/bitwriter.go", "fse/bytereader.go", "fse/bitreader.go", "fse/compress.go"]),
        ("huff0", ["huff0/huff0.go", "huff0/bitwriter.go", "huff0/compress.go"]),
        ("xxhash", ["zstd/internal/xxhash/xxhash.go", "zstd/internal/xxhash/xxhash_other.go"]),
        ("compress", ["compressible.go"]),
        ("s2", ["s2/s2.go", "s2/decode.go", "s2/hashtable_pool.go", "s2/dict.go", "s2/encode.go", "s2/encode_go.go", "s2/encode_all.go", "s2/encode_better.go", "s2/encode_best.go"]),
        ("zstd", ["zstd/zstd.go", "zstd/hash.go", "zstd/matchlen_generic.go", "zstd/bitwriter.go", "zstd/seqenc.go", "zstd/fse_encoder.go",
                  "zstd/fse_predefined.go", "zstd/frameenc.go", "zstd/blockenc.go", "zstd/enc_base.go", "zstd/enc_fast.go", "zstd/enc_dfast.go",
                  "zstd/enc_better.go", "zstd/enc_best.go", "zstd/seqdec.go", "zstd/dict.go", "zstd/bitreader.go", "zstd/blockdec.go", "zstd/framedec.go", "zstd/fse_decoder.go",
                  "zstd/fse_decoder_generic.go", "zstd/encoder_options.go", "zstd/encoder.go"]),
    ],
    # path -> names of top-level declarations (or Type.method) that are not translated
    "skip": {
        "fse/fse.go": set(),
        "huff0/compress.go": {"Scratch.compress4Xp"},  # the goroutine variant, only reachable from an `if false` block
        "s2/s2.go": {"_", "byter", "crc", "crcTable"},                          # the stream framing's CRC32C (hash/crc32) and an interface assertion
        "s2/encode.go": {"EstimateBlockSize", "estblockPool", "ConcatBlocks"},  # size estimation (calcBlockSize, not an encoder), block concatenation
        "s2/dict.go": {"Dict.Decode", "MakeDict", "MakeDictManual"},            # the decoder half; dictionary construction by search
        "s2/encode_go.go": {"calcBlockSize", "calcBlockSizeSmall", "cvtLZ4BlockAsm", "cvtLZ4BlockSnappyAsm", "cvtLZ4sBlockAsm", "cvtLZ4sBlockSnappyAsm"},
        "zstd/zstd.go": {"_", "byter"},                                   # an interface assertion on bytes.Buffer (decoder input)
        "zstd/encoder_options.go": {"EncoderLevelFromString", "EncoderLevel.String", "WithEncoderDict"},  # strings package; loadDict (decoder tables)
    },
    # path -> the ONLY declarations taken from that file (the rest of it is the decoder / the streaming writer)
    "only": {
        "s2/decode.go": {"ErrCorrupt", "ErrCRC", "ErrTooLarge", "ErrUnsupported"},   # the package's error values
        "zstd/seqdec.go": {"seq", "seqCompMode", "compModePredefined", "compModeRLE", "compModeFSE", "compModeRepeat"},
        "zstd/dict.go": {"dict", "dict.*", "dictMagic", "dictMaxLength"},
        # EncodeAll, and the streaming writer without its job mode (enc_jobs.go: worker goroutines fed through channels)
        "zstd/encoder.go": {"Encoder", "encoder", "encoderState", "Encoder.encodeAll", "Encoder.MaxEncodedSize", "Encoder.Reset", "Encoder.Write",
                            "Encoder.writeBlocks", "Encoder.nextBlock", "Encoder.Flush", "Encoder.Close"},
        # constants and the block / literals type enumerations the encoder shares with the decoder
        "zstd/blockdec.go": {"blockType", "blockTypeRaw", "blockTypeRLE", "blockTypeCompressed", "blockTypeReserved", "literalsBlockType",
                             "literalsBlockRaw", "literalsBlockRLE", "literalsBlockCompressed", "literalsBlockTreeless", "maxCompressedBlockSize",
                             "compressedBlockOverAlloc", "maxCompressedBlockSizeAlloc", "maxBlockSize", "maxMatchLen", "maxSequences", "maxOffsetBits"},
        "zstd/bitreader.go": {"highBits"},
        "zstd/framedec.go": {"MinWindowSize", "MaxWindowSize", "frameMagic", "skippableFrameMagic"},
        # initPredefined builds the predefined DECODER tables first and copies their normalised counts into the encoders
        "zstd/fse_decoder.go": {"tablelogAbsoluteMax", "maxMemoryUsage", "maxTableLog", "maxTablesize", "maxTableMask", "minTablelog", "maxSymbolValue", "fseDecoder", "tableStep", "decSymbol", "decSymbol.*", "newDecSymbol",
                                "decSymbolValue", "fseDecoder.transform"},
    },
    "drop_fields": {
        "zstd.Encoder": {"encoders", "init"},              # the pool of encoders EncodeAll draws from (the driver hands it one)
        "zstd.encoderState": {"jobs"},                     # WithConcurrentBlocks' state (enc_jobs.go)
        "zstd.dict": {"llDec", "ofDec", "mlDec"},          # decoder tables of a loaded dictionary
"fse.Scratch": {"decTable"}, "huff0.Scratch": {"dt", "decPool"}},  # decoder halves of the shared scratch structs
    # path -> [(regular expression, replacement, why)]: source patches applied before parsing
    "patches": {
        "zstd/encoder.go": [
            (r"\tif e\.o\.concurrentBlocks \{\n\t\treturn e\.(writeJobs\(p\)|flushJobs\(\)|closeJobs\(\))\n\t\}\n", "",
             "Write / Flush / Close hand over to the job mode first: not translated (goroutines fed through channels), never taken here"),
            (r"\tif e\.o\.concurrentBlocks \{\n\t\te\.shutdownJobWorkers\(\)\n.*?\t\tjs\.started = false\n\t\}\n", "",
             "Reset's job-mode block, likewise"),
        ],
        "s2/hashtable_pool.go": [
            (r"= sync\.Pool\{New: func\(\) any \{ return &\w+\{\} \}\}", " sync.Pool",
             "sync.Pool is an allocation cache: its New hook is not needed when Get is replaced by a fresh table (next patch)"),
            (r"\w+\.Get\(\)\.\(\*(\w+)\)", r"&\1{}",
             "a pooled table is zeroed right after Get(): a fresh zero table is the same value"),
        ],
    },
}
