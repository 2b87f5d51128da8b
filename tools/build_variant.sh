#!/bin/bash
# Builds a profiling variant of the library with other tile shapes:
#   tools/build_variant.sh <name> '<extra -D flags>'
# -> build/variants/<name>/libeigen_amd_blas.so (run a tool against it with LD_LIBRARY_PATH).
set -e
name=$1; shift
flags="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out/obj
cd $root/eigen_amd/csrc
for f in gemm_inst_f32 gemm_inst_f32s gemm_inst_f64 gemm_inst_f64s gemm_inst_c32 gemm_inst_c32s gemm_inst_c64 gemm_inst_c64s lu trsm syrk level3x pgemm xfer blas_abi; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $f.hip -o $out/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libeigen_amd_blas.so $out/obj/*.o -ldl
rm -rf $out/obj
echo "built $out/libeigen_amd_blas.so"
