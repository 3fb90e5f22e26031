#!/bin/bash
# Round 4: per-phase clocks of the S2 LDS kernel's fused step (one 64 KiB block of 'J' and of 'T').
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
python - <<PY
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import corpora
open("/tmp/J.bin", "wb").write(corpora.corpus("J", 1, 65536).tobytes())
open("/tmp/T.bin", "wb").write(corpora.corpus("T", 1, 65536).tobytes())
PY
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -I compress_amd/csrc tools/s2_lds_prof.hip"
/opt/rocm/bin/hipcc $F -DKC_S2_PROF -o /tmp/s2prof 2>/dev/null
/opt/rocm/bin/hipcc $F -o /tmp/s2noprof 2>/dev/null
for k in J T; do
  echo "== $k, instrumented"; /tmp/s2prof /tmp/$k.bin 0
  echo "== $k, plain"; /tmp/s2noprof /tmp/$k.bin 0; /tmp/s2noprof /tmp/$k.bin 1
done 2>&1 | tee gpurun_out/r4f/s2prof.txt
