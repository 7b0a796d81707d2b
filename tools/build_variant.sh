#!/bin/bash
# build_variant.sh NAME "-DFOO=1 ..." : libdrrt_hip variant with extra defines -> adjointnonlinearraytracing_amd/csrc/_variants/NAME.so
# (same-box A/B runs: DRRT_HIP_LIB=.../NAME.so python bench.py ..., tools/ab_variants.sh)
set -e
cd "$(dirname "$0")/../adjointnonlinearraytracing_amd/csrc"
N=$1; D=$2
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function"
mkdir -p _variants/_o_$N
for f in drrt_kernels drrt_sort; do /opt/rocm/bin/hipcc $FL $D -DDRRT_SRC_ID=\"variant:$N\" -c $f.hip -o _variants/_o_$N/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/$N.so _variants/_o_$N/drrt_kernels.o _variants/_o_$N/drrt_sort.o _build/drrt_sensor.o _build/drrt_source.o
rm -rf _variants/_o_$N
echo built _variants/$N.so
