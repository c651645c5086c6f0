#!/bin/bash
# One library per setting of lab switches of the product kernel (pa_spmv_kernel.h): only pa_csr.hip (the product launch) is rebuilt,
# the other translation units come from csrc/obj (run `make -C partitionedarrays.jl_amd/csrc` first).
#   bash tools/probe/build_k1_variants.sh name "-DPA_K1_REC=0 ..." [name flags ...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/partitionedarrays.jl_amd/csrc
mkdir -p $R/tools/probe/build
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$R/include -I$C -I/opt/rocm/include $flags -x hip -c $C/pa_csr.hip -o /tmp/pa_csr_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/pa_csr_$name.o $(ls $C/obj/*.o | grep -v pa_csr.o) -o $R/tools/probe/build/libpa_hip_$name.so -ldl
  echo built $name
done
