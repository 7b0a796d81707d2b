#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]  -> adjointnonlinearraytracing_amd/csrc/_variants/NAME.so  (development A-B builds)
set -e
cd "$(dirname "$0")/../adjointnonlinearraytracing_amd/csrc"
name=$1; shift
mkdir -p _variants/o_$name
for f in drrt_kernels drrt_sort drrt_sensor drrt_source; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine --offload-arch=gfx950 -fvisibility=hidden "$@" -c $f.hip -o _variants/o_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/$name.so _variants/o_$name/*.o
echo built _variants/$name.so
