// kc_probe.hip (measurement probes behind the C ABI: host code and kernels in one file) for the whole-library emulator build, a
// translation unit of its own (its anonymous namespace has names the kernel files use too).  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include "../../compress_amd/csrc/kc_probe.hip"
