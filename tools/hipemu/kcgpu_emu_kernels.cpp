// Every kernel file of compress_amd/csrc in ONE translation unit for the hipemu CPU emulator (the device functions of the shared
// headers are plain functions here), for the whole-library emulator build: tools/emu_host_check.py.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <vector>
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
#include "../../compress_amd/csrc/kc_s2_best.hip"
#include "../../compress_amd/csrc/kc_zstd_match_best.hip"
#include "../../compress_amd/csrc/kc_zstd_decode.hip"
#include "../../compress_amd/csrc/kc_s2_decode.hip"
