"""compress_amd — MI355X-native block-parallel zstd / S2 encode engine.

Host-side mirror of the klauspost/compress encoder API for the encode hot path
(`zstd.Encoder.EncodeAll`, `s2.Encode`, `s2.WriterCustomEncoder`) over the C ABI in
include/kcgpu.h (libkcgpu.so: hand-written HIP kernels for gfx950).  There is no CPU
fallback in this package: without the HIP library or a GPU every encode call raises.
"""
from . import zstd, s2  # noqa: F401
from ._lib import KcError, lib_path, load as load_library  # noqa: F401

__all__ = ["zstd", "s2", "KcError", "lib_path", "load_library"]
