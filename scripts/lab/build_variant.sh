#!/bin/bash
# LAB: build an A/B copy of the product library with extra compiler flags on some sources:
#   scripts/lab/build_variant.sh NAME "-DFOO=1" mlp_bwd.hip gemm.hip   ->  diffulab_amd/csrc/variants/libdiffulab_hip_NAME.so
# (load it with DIFFULAB_HIP_LIB=<that path>; variants/*.so is git-ignored but not gpurun-ignored: it travels to the GPU box)
set -e
cd "$(dirname "$0")/../../diffulab_amd/csrc"
name=$1; flags=$2; shift 2
make -s ../libdiffulab_hip.so
mkdir -p build/lab_$name variants
objs=""
for src in elementwise gemm gemm_ln gemm_w4 mlp_bwd norm attention embed unet tokens block f32; do
  if [[ " $* " == *" $src.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-function \
      -Wno-unused-variable $flags -c $src.hip -o build/lab_$name/$src.o &
    objs="$objs build/lab_$name/$src.o"
  else
    objs="$objs build/$src.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/libdiffulab_hip_$name.so
echo "built diffulab_amd/csrc/variants/libdiffulab_hip_$name.so"
