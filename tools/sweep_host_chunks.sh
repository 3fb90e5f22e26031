#!/bin/bash
# end-to-end (pageable host -> frames in pageable host) rate of the C2 workload for several kernel-chunk schedules
for sch in "512,512,1024,2048" "1024" "768" "256,512,768,2560" "1024,1024,2048" "1536,2560" "2048"; do
  echo "== KC_HOST_CHUNKS_MIB=$sch"
  KC_HOST_TRACE=1 KC_HOST_CHUNKS_MIB=$sch python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "drained" | tail -1
done
