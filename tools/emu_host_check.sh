#!/bin/bash
# The `-m gpu` tests against the WHOLE library built for the wave emulator under AddressSanitizer: host code (batch cutting, scratch
# sizing, re-run bookkeeping, host pipelines) + every kernel, each hipMalloc a poisoned malloc block of its own — a memory-error hunt
# for the C ABI as a whole (an out-of-bounds access of a device buffer is an ASan report with the source line; a read of a buffer
# nothing wrote sees 0xA7 / pseudo-random bytes instead of a fresh GPU's zeros).  TEST INFRASTRUCTURE; proves nothing about the
# device (streams are synchronous here) and is not part of the CPU or GPU suites.  Tests too large for the emulator time out
# (--timeout) and show as crashed workers.  The device DECODERS (kc_zstd_decode.hip, kc_s2_decode.hip) are verifiers written for the
# hardware's lockstep waves (a wavefront fence between one lane's table build and the other lanes' reads, no rendezvous the emulator
# could see): their tests fail here by construction — deselect them (-k "not decode").
#   tools/emu_host_check.sh [pytest args...]      e.g.  tools/emu_host_check.sh tests/test_gpu_s2.py -k "not full_size and not decode"
#   HIPEMU_POISON=rand|<byte>   NOASAN=1   JOBS=7   TIMEOUT=600   SKIP_BUILD=1 (a second run beside a running one)
cd "$(dirname "$0")/.."
PRE=
[ -z "$NOASAN" ] && PRE=$(gcc -print-file-name=libasan.so)
if [ -z "$SKIP_BUILD" ]; then
  if [ -z "$NOASAN" ]; then ASAN=1 tools/build_emu_lib.sh > /dev/null || exit 1; else tools/build_emu_lib.sh > /dev/null || exit 1; fi
fi
[ $# -eq 0 ] && set -- tests
LD_PRELOAD=$PRE ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 KC_LIB_TAG=emu KC_EMU_PROCS=1 PYTHONPATH=tools \
  python -X faulthandler -m pytest -m gpu -p emu_torch_shim -q -n ${JOBS:-7} --timeout ${TIMEOUT:-600} --timeout-method=thread -p no:cacheprovider "$@"
