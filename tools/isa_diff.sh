#!/bin/bash
# Device code of the working tree against a commit's, kernel file by kernel file: hipcc -S for gfx950 of both, comments / debug
# directives / compile-unit ids dropped, md5 of the rest.  "same" = the change did nothing on the device (emulator-only macros,
# comments, host code).  No GPU needed.  Usage: bash tools/isa_diff.sh <commit>
set -e
C=${1:?usage: tools/isa_diff.sh <commit>}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$R" archive "$C" compress_amd/csrc | tar -x -C "$T"
F='^\s*;\|\.file\|\.ident\|\.loc\|\.Ltmp\|\.cfi\|__hip_cuid'
rc=0
for f in $(cd "$R/compress_amd/csrc" && ls *.hip); do
  if [ ! -f "$T/compress_amd/csrc/$f" ]; then echo "$f new"; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -S --cuda-device-only "$T/compress_amd/csrc/$f" -o "$T/o.s" 2>/dev/null &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -S --cuda-device-only "$R/compress_amd/csrc/$f" -o "$T/n.s" 2>/dev/null
  wait
  a=$(grep -v "$F" "$T/o.s" | md5sum | cut -c1-12); b=$(grep -v "$F" "$T/n.s" | md5sum | cut -c1-12)
  if [ "$a" = "$b" ]; then echo "$f same $a"; else echo "$f DIFFERENT $a $b"; rc=1; fi
done
rm -rf "$T"
exit $rc
