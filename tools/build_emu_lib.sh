#!/bin/bash
# The WHOLE product library (host code + every kernel) compiled by g++ for the wave emulator -> compress_amd/libkcgpu_emu.so
# (loaded with KC_LIB_TAG=emu).  ASAN=1: with AddressSanitizer (run python with LD_PRELOAD=$(gcc -print-file-name=libasan.so)).
# TEST INFRASTRUCTURE: see tools/emu_host_check.py.
set -e
cd "$(dirname "$0")/.."
T=${EMU_TAG:-emu}
B=tools/_build/$T${ASAN:+_asan}
mkdir -p $B
FL="-O1 -g -std=c++17 -fPIC -x c++ -DKC_HIPEMU_HOST -I tools/hipemu -Wall -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas ${ASAN:+-fsanitize=address -fno-omit-frame-pointer}"
pids=()
for f in compress_amd/csrc/kc_*.cpp tools/hipemu/kcgpu_emu_kernels.cpp tools/hipemu/kcgpu_emu_probe.cpp tools/hipemu/hipemu.cpp; do
  g++ $FL -c $f -o $B/$(basename $f).o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
g++ -shared ${ASAN:+-fsanitize=address} -o $B/libkcgpu_$T.so $B/*.o -ldl -lpthread
ln -sf ../$B/libkcgpu_$T.so compress_amd/libkcgpu_$T.so   # KC_LIB_TAG=$T loads it (EMU_TAG=emu2: a second build beside a running one)
echo $B/libkcgpu_$T.so
