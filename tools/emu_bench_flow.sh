#!/bin/bash
# bench.py's own call sequence (two contexts Begin / End, chained streams, the all-frames device round trip through the decode kernels,
# the host-buffer pipeline of `end_to_end`, the oracle byte compare) at a few MiB per configuration, on the whole-library emulator
# build under AddressSanitizer (tools/emu_host_check.sh has the details).  TEST INFRASTRUCTURE; the numbers it prints are meaningless.
cd "$(dirname "$0")/.."
[ -z "$SKIP_BUILD" ] && { ASAN=1 tools/build_emu_lib.sh > /dev/null || exit 1; }
PRE=$(gcc -print-file-name=libasan.so)
rc=0
for spec in ${CONFIGS:-C2 C2H C3 C4 C4A C5 B4 C4:--s2-level=1 C4:--s2-level=4 C2:--no-pipeline C2:--path=lds}; do
  cfg=${spec%%:*}; extra=${spec#*:}; [ "$extra" = "$spec" ] && extra=
  out=$(LD_PRELOAD=$PRE ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 KC_LIB_TAG=emu KC_EMU_PROCS=1 PYTHONPATH=tools timeout ${TIMEOUT:-1800} \
    python -c "import sys, runpy, emu_torch_shim; sys.argv = ['bench.py'] + sys.argv[1:]; runpy.run_path('bench.py', run_name='__main__')" \
    --config $cfg --gib ${GIB:-0.00390625} --steps 3 --warmup 1 --no-also --cpu-sample-units 16 $extra 2>&1)
  line=$(echo "$out" | grep '^{' | tail -1)
  if [ -z "$line" ]; then echo "$spec: NO LINE"; echo "$out" | grep -v dist-packages | tail -25; rc=1; continue; fi
  echo "$spec: $(echo "$line" | python -c "import sys, json; j = json.loads(sys.stdin.read()); print('bit_exact', j['bit_exact_vs_oracle_on_sample'], 'roundtrip', j['device_roundtrip_all_frames'], 'e2e_same', (j.get('end_to_end') or {}).get('same_bytes_as_device_path'), 'e2e_err', (j.get('end_to_end') or {}).get('error'), 'path', j['match_path'], 'ratio', j['ratio'])")"
done
exit $rc
