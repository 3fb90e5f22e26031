#!/bin/bash
# end-to-end (pageable host -> frames in pageable host) time of a configuration for several kernel-chunk schedules (MiB, last repeats)
# usage: tools/sweep_host_chunks.sh [config] [schedules...]
CFG=${1:-C2}; shift
SCHED=("$@"); [ ${#SCHED[@]} -eq 0 ] && SCHED=("1024" "512" "512,512,1024,2048" "2048")
for sch in "${SCHED[@]}"; do
  echo "== $CFG KC_HOST_CHUNKS_MIB=$sch"
  KC_HOST_TRACE=1 KC_HOST_CHUNKS_MIB=$sch python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-device-verify 2>&1 | grep -E "^\[kc host\] drained" | tail -1
done
