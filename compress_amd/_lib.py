"""ctypes binding of include/kcgpu.h (libkcgpu.so).  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KC_LIB_TAG=<tag>: a measurement build of the same sources with other compile-time constants (compress_amd/build.py, KC_BUILD_TAG)
_SO = os.path.join(_HERE, "libkcgpu%s.so" % ("_" + os.environ["KC_LIB_TAG"] if os.environ.get("KC_LIB_TAG") else ""))

KC_OK = 0
KC_ERR_BAD_ARG, KC_ERR_DST_TOO_SMALL, KC_ERR_HIP, KC_ERR_UNSUPPORTED, KC_ERR_NO_DEVICE, KC_ERR_INTERNAL = -1, -2, -3, -4, -5, -6
_NAMES = {0: "KC_OK", -1: "KC_ERR_BAD_ARG", -2: "KC_ERR_DST_TOO_SMALL", -3: "KC_ERR_HIP", -4: "KC_ERR_UNSUPPORTED",
          -5: "KC_ERR_NO_DEVICE", -6: "KC_ERR_INTERNAL"}


class KcError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("%s%s" % (_NAMES.get(status, str(status)), (": " + msg) if msg else ""))


class ZstdOpts(C.Structure):
    _fields_ = [
        ("level", C.c_int32), ("window_size", C.c_int32), ("block_size", C.c_int32), ("crc", C.c_int32),
        ("single", C.c_int32), ("full_zero", C.c_int32), ("no_entropy", C.c_int32), ("all_lit_entropy", C.c_int32),
        ("low_mem", C.c_int32), ("custom_window", C.c_int32), ("custom_block", C.c_int32), ("custom_alent", C.c_int32),
        ("dict_id", C.c_uint32), ("dict", C.c_void_p), ("dict_len", C.c_uint64),
        ("dict_offsets", C.c_uint32 * 3), ("dict_huf_len", C.c_int32), ("dict_huf_log", C.c_int32),
        ("dict_huf_val", C.c_uint16 * 256), ("dict_huf_nbits", C.c_uint8 * 256), ("concurrent", C.c_int32),
    ]


class Timings(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("match_ms", C.c_float), ("entropy_ms", C.c_float), ("other_ms", C.c_float),
                ("redo_units", C.c_uint32), ("prep_ms", C.c_float)]


# every symbol include/kcgpu.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "kc_zstd_opts_default", "kc_zstd_opts_level", "kc_zstd_opts_window", "kc_zstd_opts_crc", "kc_zstd_opts_zero_frames",
    "kc_zstd_opts_no_entropy", "kc_zstd_opts_all_lit_entropy", "kc_zstd_opts_single_segment", "kc_zstd_opts_concurrency", "kc_zstd_opts_dict_raw", "kc_zstd_opts_dict",
    "kc_zstd_max_encoded_size", "kc_ctx_create", "kc_ctx_destroy", "kc_last_error", "kc_device_info",
    "kc_zstd_encode_units", "kc_zstd_encode_units_dev", "kc_zstd_encode_streams_dev", "kc_zstd_encode_streams", "kc_zstd_encode_streams_cuts_dev", "kc_zstd_encode_streams_cuts", "kc_zstd_encode_units_submit", "kc_s2_encode_blocks_lvl_submit", "kc_wait", "kc_zstd_plan_stream_blocks", "kc_zstd_encode_units_dev_begin", "kc_zstd_encode_units_dev_end", "kc_zstd_encode_units_dev_end_at", "kc_ctx_chain_after", "kc_xxh64_units_dev", "kc_zstd_debug_parse_dev",
    "kc_s2_max_encoded_len", "kc_s2_encode_blocks", "kc_s2_encode_blocks_dev", "kc_s2_encode_stream_dev", "kc_s2_decode_blocks_dev", "kc_zstd_decode_units_dev", "kc_zstd_decode_units_dict_dev", "kc_s2_encode_block", "kc_s2_hook_stats", "kc_s2_encode_blocks_lvl", "kc_s2_encode_blocks_lvl_dev", "kc_s2_encode_blocks_lvl_dev_begin", "kc_s2_encode_blocks_lvl_dev_end_at", "kc_s2_encode_stream_lvl_dev",
    "kc_last_timings", "kc_corpus_fill", "kc_ctx_set_option", "kc_ctx_get_option", "kc_zstd_encode_jobs", "kc_zstd_job_size", "kc_zstd_overlap_size",
    "kc_probe_table_pattern", "kc_probe_pcie", "kc_ctx_trim", "kc_device_trim", "kc_s2_hook_declined", "kc_create_error", "kc_host_alloc", "kc_host_free",
]

# kc_option / KC_PATH_* (include/kcgpu.h)
PATH_AUTO, PATH_HBM, PATH_LDS = 0, 1, 2
OPT_MATCH_PATH, OPT_ZFAST_LDS_MAX_UNITS, OPT_S2_LDS_MAX_BLOCKS, OPT_SPEC_W0, OPT_SPEC_GROW, OPT_LDS_SPEC_W0 = 1, 2, 3, 4, 5, 6
OPT_HOST_SERIAL, OPT_HOST_PIPE_MIB, OPT_HOST_OVERLAP_MIN_MIB, OPT_HOST_COPY_THREADS, OPT_HOST_TRACE, OPT_HOST_CHUNK_MIB = 7, 8, 9, 10, 11, 12
OPT_K2_PROF, OPT_S2_HOOK_WAIT_US, OPT_S2_HOOK_BATCH, OPT_TEST_FEED_REDO, OPT_S2_LDS_SPEC_W0, OPT_MAX_SCRATCH_MIB, OPT_LAST_PATH, OPT_LAST_BATCHES = 13, 14, 15, 16, 17, 18, 100, 101
OPT_JOB_PRIME = 30
OPT_STAGE2_STREAM = 31
OPT_HOST_CHUNK_MIB_APPEND = 32
OPT_HOST_ROLL, OPT_HOST_ROLL_MIB = 33, 34
OPT_S2_HOOK_HOST_FIRST = 35
_PATHS = {"auto": PATH_AUTO, "hbm": PATH_HBM, "lds": PATH_LDS, None: PATH_AUTO}

_lib = None


def lib_path():
    return _SO


def load():
    """Load libkcgpu.so; raises if it has not been built (python -m compress_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise KcError(KC_ERR_NO_DEVICE, "libkcgpu.so not built: run `python -m compress_amd.build` (no CPU fallback exists)")
    # One HIP runtime per process: when PyTorch is present it must be imported first so that
    # libkcgpu.so binds to the libamdhip64.so.7 PyTorch already mapped (device pointers and
    # streams are then shareable); without PyTorch the system ROCm runtime is used.
    if os.environ.get("KC_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_SO)
    vp, u64 = C.c_void_p, C.c_uint64
    po = C.POINTER(ZstdOpts)
    L.kc_zstd_opts_default.argtypes = [po]
    L.kc_zstd_opts_default.restype = None
    for n in ("level", "window", "crc", "zero_frames", "no_entropy", "all_lit_entropy", "single_segment", "concurrency"):
        f = getattr(L, "kc_zstd_opts_" + n)
        f.argtypes = [po, C.c_int]
        f.restype = C.c_int
    L.kc_zstd_opts_dict_raw.argtypes = [po, C.c_uint32, vp, u64]
    L.kc_zstd_opts_dict_raw.restype = C.c_int
    L.kc_zstd_opts_dict.argtypes = [po, vp, u64]
    L.kc_zstd_opts_dict.restype = C.c_int
    L.kc_zstd_max_encoded_size.argtypes = [po, C.c_int64]
    L.kc_zstd_max_encoded_size.restype = C.c_int64
    L.kc_ctx_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.kc_ctx_create.restype = C.c_int
    L.kc_ctx_destroy.argtypes = [vp]
    L.kc_ctx_destroy.restype = None
    L.kc_last_error.argtypes = [vp]
    L.kc_last_error.restype = C.c_char_p
    L.kc_device_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]
    L.kc_device_info.restype = C.c_int
    for n in ("kc_zstd_encode_units", "kc_zstd_encode_units_dev", "kc_zstd_encode_streams_dev", "kc_zstd_encode_streams"):
        f = getattr(L, n)
        f.argtypes = [vp, po, vp, vp, C.c_uint32, vp, u64, vp]
        f.restype = C.c_int
    L.kc_zstd_encode_units_submit.argtypes = [vp, po, vp, vp, C.c_uint32, vp, u64, vp]
    L.kc_zstd_encode_units_submit.restype = C.c_int
    L.kc_s2_encode_blocks_lvl_submit.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, vp, u64, vp]
    L.kc_s2_encode_blocks_lvl_submit.restype = C.c_int
    L.kc_zstd_plan_stream_blocks.argtypes = [C.c_int32, u64, vp, u64, vp, u64, vp]
    L.kc_zstd_plan_stream_blocks.restype = C.c_int64
    L.kc_wait.argtypes = [vp]
    L.kc_wait.restype = C.c_int
    for n in ("kc_zstd_encode_streams_cuts_dev", "kc_zstd_encode_streams_cuts"):
        f = getattr(L, n)
        f.argtypes = [vp, po, vp, vp, C.c_uint32, vp, vp, vp, u64, vp]
        f.restype = C.c_int
    L.kc_zstd_encode_units_dev_begin.argtypes = [vp, po, vp, vp, C.c_uint32, vp, u64]
    L.kc_zstd_encode_units_dev_begin.restype = C.c_int
    L.kc_zstd_encode_units_dev_end.argtypes = [vp, vp]
    L.kc_zstd_encode_units_dev_end.restype = C.c_int
    L.kc_zstd_encode_units_dev_end_at.argtypes = [vp, vp, C.c_uint64, vp]
    L.kc_s2_encode_blocks_lvl_dev_begin.argtypes = [vp, C.c_int, vp, vp, C.c_uint32]
    L.kc_s2_encode_blocks_lvl_dev_begin.restype = C.c_int
    L.kc_s2_encode_blocks_lvl_dev_end_at.argtypes = [vp, vp, C.c_uint64, vp]
    L.kc_s2_encode_blocks_lvl_dev_end_at.restype = C.c_int
    L.kc_zstd_encode_units_dev_end_at.restype = C.c_int
    L.kc_ctx_chain_after.argtypes = [vp, vp]
    L.kc_ctx_chain_after.restype = None
    L.kc_xxh64_units_dev.argtypes = [vp, vp, vp, C.c_uint32, vp]
    L.kc_xxh64_units_dev.restype = C.c_int
    L.kc_zstd_debug_parse_dev.argtypes = [vp, po, vp, vp, C.c_uint32, vp, u64, vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.kc_zstd_debug_parse_dev.restype = C.c_int
    L.kc_s2_max_encoded_len.argtypes = [C.c_int64]
    L.kc_s2_max_encoded_len.restype = C.c_int64
    for n in ("kc_s2_encode_blocks", "kc_s2_encode_blocks_dev"):
        f = getattr(L, n)
        f.argtypes = [vp, vp, vp, C.c_uint32, vp, u64, vp]
        f.restype = C.c_int
    for n in ("kc_s2_encode_blocks_lvl", "kc_s2_encode_blocks_lvl_dev"):
        f = getattr(L, n)
        f.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, vp, u64, vp]
        f.restype = C.c_int
    L.kc_s2_encode_stream_lvl_dev.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, vp, u64, vp, C.c_int]
    L.kc_s2_encode_stream_lvl_dev.restype = C.c_int
    L.kc_s2_encode_stream_dev.argtypes = [vp, vp, vp, C.c_uint32, vp, u64, vp, C.c_int]
    L.kc_s2_encode_stream_dev.restype = C.c_int
    L.kc_zstd_decode_units_dev.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp]
    L.kc_zstd_decode_units_dev.restype = C.c_int
    L.kc_zstd_decode_units_dict_dev.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, u64]
    L.kc_zstd_decode_units_dict_dev.restype = C.c_int
    L.kc_s2_decode_blocks_dev.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp]
    L.kc_s2_decode_blocks_dev.restype = C.c_int
    L.kc_s2_encode_block.argtypes = [vp, vp, u64, vp, u64]
    L.kc_s2_encode_block.restype = C.c_int64
    L.kc_s2_hook_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.kc_s2_hook_stats.restype = None
    L.kc_last_timings.argtypes = [vp, C.POINTER(Timings)]
    L.kc_last_timings.restype = C.c_int
    L.kc_zstd_encode_jobs.argtypes = [vp, po, vp, u64, vp, u64, vp, u64, C.POINTER(u64)]
    L.kc_zstd_encode_jobs.restype = C.c_int
    L.kc_zstd_job_size.argtypes = [po]
    L.kc_zstd_job_size.restype = C.c_int64
    L.kc_zstd_overlap_size.argtypes = [po]
    L.kc_zstd_overlap_size.restype = C.c_int64
    L.kc_ctx_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.kc_ctx_set_option.restype = C.c_int
    L.kc_ctx_get_option.argtypes = [vp, C.c_int]
    L.kc_ctx_get_option.restype = C.c_int64
    L.kc_s2_hook_declined.argtypes = [vp]
    L.kc_s2_hook_declined.restype = C.c_uint64
    L.kc_host_alloc.argtypes = [C.POINTER(vp), u64]
    L.kc_host_alloc.restype = C.c_int
    L.kc_host_free.argtypes = [vp]
    L.kc_host_free.restype = None
    L.kc_create_error.argtypes = []
    L.kc_create_error.restype = C.c_char_p
    L.kc_ctx_trim.argtypes = [vp]
    L.kc_ctx_trim.restype = C.c_int
    L.kc_device_trim.argtypes = [C.c_int]
    L.kc_device_trim.restype = C.c_int
    L.kc_corpus_fill.argtypes = [C.c_int, u64, u64, C.c_uint32, C.c_uint32, vp, C.c_int]
    L.kc_corpus_fill.restype = C.c_int
    if hasattr(L, "kc_probe_pcie"):  # (the wave-emulator build of the library, tools/build_emu_lib.sh, leaves the device probes out)
        L.kc_probe_table_pattern.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.kc_probe_table_pattern.restype = C.c_int
        L.kc_probe_pcie.argtypes = [vp, u64, C.POINTER(C.c_double)]
        L.kc_probe_pcie.restype = C.c_int
    _lib = L
    return L


class Context:
    """kc_ctx: device scratch + stream.  stream: an int hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        self.L = load()
        h = C.c_void_p()
        st = self.L.kc_ctx_create(C.byref(h), device, C.c_void_p(stream) if stream else None)
        if st != KC_OK:
            why = self.L.kc_create_error().decode(errors="replace")
            raise KcError(st, "kc_ctx_create: %s (is a gfx950 GPU visible? there is no CPU fallback)" % (why or "no reason recorded"))
        self.h = h
        self._apply_env()

    # Measurement knobs: the library itself reads no environment variable (include/kcgpu.h); this harness maps the KC_* variables the
    # tools/ scripts set onto kc_ctx_set_option, once per context.
    _ENV_OPTS = (("KC_MATCH_PATH", 1), ("KC_ZFAST_LDS_MAX_UNITS", 2), ("KC_S2_LDS_MAX_BLOCKS", 3), ("KC_SPEC_W0", 4), ("KC_SPEC_GROW", 5),
                 ("KC_LDS_SPEC_W0", 6), ("KC_S2_LDS_SPEC_W0", 17), ("KC_HOST_PIPE_MIB", 8), ("KC_HOST_OVERLAP_MIN_MIB", 9),
                 ("KC_HOST_COPY_THREADS", 10), ("KC_S2_HOOK_WAIT_US", 14), ("KC_S2_HOOK_BATCH", 15), ("KC_S2_HOOK_LANES", 29),
                 ("KC_ZFAST_EPOCH", 22), ("KC_ZFAST_XSEG_K", 23), ("KC_FUSE_RAW_XXH", 24), ("KC_ZFAST_FILTER", 25), ("KC_XXH_FIN_MODE", 26),
                 ("KC_ZFAST_VARIANT", 27), ("KC_ZFAST_PRESCAN", 28), ("KC_JOB_PRIME", 30), ("KC_HOST_ROLL", 33), ("KC_HOST_ROLL_MIB", 34), ("KC_S2_HOOK_HOST_FIRST", 35), ("KC_BETTER_DICT_EPOCH", 21), ("KC_BEST_SLOTS", 19), ("KC_MAX_SCRATCH_MIB", 18))
    _ENV_FLAGS = (("KC_HOST_SERIAL", 7), ("KC_HOST_TRACE", 11), ("KC_K2_PROF", 13))  # set by their presence

    def _apply_env(self):
        for name, key in self._ENV_OPTS:
            v = os.environ.get(name)
            if v not in (None, ""):
                try:
                    self.set_option(key, int(v))
                except (ValueError, KcError) as e:  # say WHICH variable
                    raise ValueError("environment variable %s=%r is not a valid value of option %d: %s" % (name, v, key, e)) from None
        for name, key in self._ENV_FLAGS:
            if os.environ.get(name) not in (None, "", "0"):  # '0' and the empty string mean off
                self.set_option(key, 1)
        v = os.environ.get("KC_HOST_CHUNKS_MIB")
        if v:
            sizes = [int(x) for x in v.split(",") if x.strip()]
            self.set_option(OPT_HOST_CHUNK_MIB, sizes[0])
            for x in sizes[1:]:
                self.set_option(OPT_HOST_CHUNK_MIB_APPEND, x)

    def close(self):
        if getattr(self, "h", None):
            self.L.kc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st != KC_OK:
            raise KcError(st, self.L.kc_last_error(self.h).decode(errors="replace"))

    def set_option(self, key, value):
        """kc_ctx_set_option (keys: OPT_* above)."""
        self.check(self.L.kc_ctx_set_option(self.h, int(key), int(value)))

    def get_option(self, key):
        return int(self.L.kc_ctx_get_option(self.h, int(key)))

    def set_path(self, path):
        """'auto' | 'hbm' | 'lds': which kernel family serves SpeedFastest / s2.Encode batches (KC_OPT_MATCH_PATH)."""
        self.set_option(OPT_MATCH_PATH, _PATHS[path] if not isinstance(path, int) else path)

    def last_path(self):
        return {PATH_HBM: "hbm", PATH_LDS: "lds"}.get(self.get_option(OPT_LAST_PATH), "none")

    def device_info(self):
        ncu, lds, clk = C.c_int32(), C.c_int32(), C.c_int32()
        name = C.create_string_buffer(128)
        self.check(self.L.kc_device_info(self.h, C.byref(ncu), C.byref(lds), C.byref(clk), name, 128))
        return {"n_cu": ncu.value, "lds_per_cu": lds.value, "clock_khz": clk.value, "arch": name.value.decode()}

    def probe_table_pattern(self, n_tables=32768, table_bytes=131072, waves=4096, iters=512):
        """kc_probe_table_pattern: requests per second of the match finders' table traffic on this device."""
        out = (C.c_double * 3)()
        self.check(self.L.kc_probe_table_pattern(self.h, n_tables, table_bytes, waves, iters, out))
        return {"pairs_per_s": out[0], "reads_per_s": out[1], "stores_per_s": out[2]}

    def probe_pcie(self, nbytes=1 << 30):
        """kc_probe_pcie: pinned H2D / D2H GB/s (alone, both at once) and the pageable <-> pinned host copy rates."""
        out = (C.c_double * 7)()
        self.check(self.L.kc_probe_pcie(self.h, nbytes, out))
        return {"h2d_GBps": out[0], "d2h_GBps": out[1], "h2d_bidir_GBps": out[2], "d2h_bidir_GBps": out[3],
                "host_copy_in_GBps": out[4], "host_copy_out_GBps": out[5], "copy_threads": int(out[6])}

    def trim(self):
        """kc_ctx_trim: free this context's device scratch (it grows back with the next call)."""
        self.check(self.L.kc_ctx_trim(self.h))

    def timings(self):
        t = Timings()
        self.check(self.L.kc_last_timings(self.h, C.byref(t)))
        return {"total_ms": t.total_ms, "match_ms": t.match_ms, "entropy_ms": t.entropy_ms, "other_ms": t.other_ms,
                "redo_units": t.redo_units, "prep_ms": t.prep_ms}


class PinnedBuffer:
    """kc_host_alloc: page-locked host memory as a numpy uint8 array (`.a`); the host-buffer entry points DMA straight from / into it."""

    def __init__(self, nbytes):
        import numpy as np
        self.L = load()
        p = C.c_void_p()
        st = self.L.kc_host_alloc(C.byref(p), int(nbytes))
        if st != KC_OK:
            raise KcError(st, "kc_host_alloc(%d)" % nbytes)
        self.p = p
        self.a = np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(p.value))

    def free(self):
        if getattr(self, "p", None):
            self.a = None
            self.L.kc_host_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_trim(device=0):
    """kc_device_trim: free what the device's rolling host pipeline holds (slots + lane scratch); raises while calls are in flight."""
    st = load().kc_device_trim(int(device))
    if st != KC_OK:
        raise KcError(st, "kc_device_trim")


def corpus_fill(kind, seed, first_unit, n_units, unit_size, threads=None):
    """Deterministic synthetic corpus (kc_corpus_fill): returns a numpy uint8 array of n_units*unit_size bytes."""
    import numpy as np
    L = load()
    buf = np.empty(int(n_units) * int(unit_size), dtype=np.uint8)
    if threads is None:
        threads = os.cpu_count() or 1
    st = L.kc_corpus_fill(ord(kind), seed, first_unit, n_units, unit_size, buf.ctypes.data, threads)
    if st != KC_OK:
        raise KcError(st, "kc_corpus_fill")
    return buf
