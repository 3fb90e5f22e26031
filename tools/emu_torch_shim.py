"""pytest plugin for tools/emu_host_check.py: the `-m gpu` tests run against the WHOLE library built for the wave emulator
(compress_amd/libkcgpu_emu.so: host code + kernels compiled by g++, `hipMalloc` = poisoned malloc blocks), where "device memory" is
host memory — so CPU tensors stand in for CUDA tensors.  TEST INFRASTRUCTURE for hunting memory errors under AddressSanitizer;
it proves nothing about the device and is not part of the CPU or GPU suites."""
import os

assert os.environ.get("KC_LIB_TAG", "").startswith("emu"), "only for the emulator build of the library"
import torch

torch.cuda.is_available = lambda: True
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.device_count = lambda: 1
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.empty_cache = lambda: None


def _to_device(self, *a, **k):
    """A "device" copy whose last 16-byte granule is whole: the kernels read sources in ALIGNED 16-byte granules that hold at least one
    readable byte (kc_zstd_match.hip: "never leaves the 16-byte granule of a readable byte") — on the hardware such a load cannot
    leave the page of its first byte; under AddressSanitizer the tail of the granule has to exist.  Anything wider is reported."""
    n = self.numel() * self.element_size()
    if self.dim() != 1 or n == 0:
        return self.clone()
    pad = torch.zeros((n + 15) // 16 * 16 // self.element_size() + (1 if (16 % self.element_size()) else 0), dtype=self.dtype)
    pad[:self.numel()].copy_(self)
    return pad[:self.numel()]


torch.Tensor.cuda = _to_device


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, *a):
        pass


torch.cuda.Stream = _Stream
torch.cuda.current_stream = lambda *a, **k: _Stream()


def _strip(fn):
    def f(*a, **k):
        d = k.get("device")
        if d is not None and "cuda" in str(d):
            k.pop("device")
        return fn(*a, **k)
    return f


for _n in ("empty", "zeros", "ones", "tensor", "arange", "full", "empty_like", "zeros_like", "randint", "frombuffer"):
    setattr(torch, _n, _strip(getattr(torch, _n)))
