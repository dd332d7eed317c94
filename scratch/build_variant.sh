#!/bin/bash
# A variant of libevdr.so that differs from the product build by extra -D flags on maxsim_fwd16.hip only (the other objects are the
# product build's): bash scratch/build_variant.sh <name> -DEVDR_REFILL_FRONT=1 ...   ->  scratch/ab/libevdr_<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/efficient-visual-document-retrieval_amd
python3 -c "import sys; sys.path.insert(0, '$R'); import evdr_amd; from evdr_amd import build; build.build(verbose=False)"
mkdir -p $R/scratch/ab /tmp/evdr_variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -fno-honor-nans -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c $P/csrc/maxsim_fwd16.hip -o /tmp/evdr_variant_$name/maxsim_fwd16.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/ab/libevdr_$name.so /tmp/evdr_variant_$name/maxsim_fwd16.o \
    $P/build/maxsim_fwd.o $P/build/maxsim_bwd.o $P/build/topk.o $P/build/prep.o $P/build/evdr_capi.o
echo $R/scratch/ab/libevdr_$name.so
