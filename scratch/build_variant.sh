#!/bin/bash
# A variant of libevdr.so that differs from the product build by extra -D flags on ONE source (the other objects are the product
# build's): bash scratch/build_variant.sh <name> [--src maxsim_bwd] -DEVDR_REFILL_FRONT=1 ...   ->  scratch/ab/libevdr_<name>.so
set -e
name=$1; shift
src=maxsim_fwd16
if [ "$1" = "--src" ]; then src=$2; shift 2; fi
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/efficient-visual-document-retrieval_amd
# The other five objects are the PRODUCT build's (build/*.o).  This script never rebuilds or relinks the in-tree libevdr.so: until round 5 it
# called build.build() first, which silently rebuilt the product library from whatever the working tree held -- a half-edited experimental
# source reached one fuzz call that way (profiles/r05_experiments.txt, item 3).  Build the product first, on a clean tree.
for o in maxsim_fwd maxsim_fwd16 maxsim_bwd topk prep qcache evdr_capi; do
    [ -f "$P/build/$o.o" ] || { echo "missing $P/build/$o.o: build the product library first (python -m evdr_amd.build) on a clean tree" >&2; exit 1; }
done
mkdir -p $R/scratch/ab /tmp/evdr_variant_$name
extra=""; [ "$src" = "maxsim_fwd16" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -fno-honor-nans -std=c++17 -Wall -Wno-unused-function $extra "$@" \
    -c $P/csrc/$src.hip -o /tmp/evdr_variant_$name/$src.o
objs=""
for o in maxsim_fwd maxsim_fwd16 maxsim_bwd topk prep qcache evdr_capi; do
    if [ "$o" = "$src" ]; then objs="$objs /tmp/evdr_variant_$name/$o.o"; else objs="$objs $P/build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/ab/libevdr_$name.so $objs
echo $R/scratch/ab/libevdr_$name.so
