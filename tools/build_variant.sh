#!/bin/bash
# build_variant.sh NAME "-DFOO=1 ..." ["ring-unit flags"] : libdrrt_hip variant with extra defines
#   -> adjointnonlinearraytracing_amd/csrc/_variants/NAME.so
# (same-box A/B runs: DRRT_HIP_LIB=.../NAME.so python bench.py ..., tools/ab_variants.sh).  The third argument replaces the
# ring-window unit's scheduling option (default: the Makefile's -mllvm -amdgpu-sched-strategy=max-ilp).
set -e
cd "$(dirname "$0")/../adjointnonlinearraytracing_amd/csrc"
N=$1; D=$2; RF=${3--mllvm -amdgpu-sched-strategy=max-ilp}
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function"
make -s _build/drrt_sensor.o _build/drrt_source.o
mkdir -p _variants/_o_$N
for f in drrt_api drrt_forward drrt_adjoint_box drrt_cable drrt_sort; do /opt/rocm/bin/hipcc $FL $D -DDRRT_SRC_ID=\"variant:$N\" -c $f.hip -o _variants/_o_$N/$f.o & done
/opt/rocm/bin/hipcc $FL $RF $D -c drrt_adjoint_ring.hip -o _variants/_o_$N/drrt_adjoint_ring.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/$N.so _variants/_o_$N/*.o _build/drrt_sensor.o _build/drrt_source.o
rm -rf _variants/_o_$N
echo built _variants/$N.so
