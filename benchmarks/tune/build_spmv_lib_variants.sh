#!/bin/bash
# Builds libhpcla_exp<N>.so = the production library with spmv.hip recompiled under -DHPCLA_EXP=N (all other objects are
# the production build's).  Used by benchmarks/tune_spmv_lib.py to time kernel changes against the shipped kernel in ONE process.
set -e
cd "$(dirname "$0")/../.."
make -s -C linearalgebrampi.jl_amd/csrc
B=linearalgebrampi.jl_amd/csrc/_build
F=${HPCLA_EXP_FILE:-spmv}        # which source file is rebuilt under -DHPCLA_EXP=N (spmv or spmm)
for n in "$@"; do
  extra=""
  if [ $((n & 16)) -ne 0 ]; then extra="-mllvm -amdgpu-kernarg-preload-count=16"; fi      # bit 16: kernel arguments preloaded into SGPRs
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -DHPCLA_EXP=$n $extra \
      -c linearalgebrampi.jl_amd/csrc/$F.hip -o benchmarks/tune/${F}_exp$n.o
  objs=$(ls $B/*.o | grep -v "/$F.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o benchmarks/tune/libhpcla_${F}_exp$n.so benchmarks/tune/${F}_exp$n.o $objs -ldl -Wl,-rpath,/opt/rocm/lib
  rm -f benchmarks/tune/${F}_exp$n.o
done
