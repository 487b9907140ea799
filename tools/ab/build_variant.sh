#!/bin/bash
# build a variant of libevhip.so with extra -D flags into tools/ab/<name>.so:  tools/ab/build_variant.sh name -DX=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../../evenvizion_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
mkdir -p /tmp/evh_var_$name
for f in evh_api evh_detect evh_image evh_match evh_ransac evh_sift evh_surf; do
  if [ $f = ${AB_FILE:-evh_detect} ] || [ ! -f $f.o ]; then /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o /tmp/evh_var_$name/$f.o; else cp $f.o /tmp/evh_var_$name/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/$name.so /tmp/evh_var_$name/*.o
echo built tools/ab/$name.so
