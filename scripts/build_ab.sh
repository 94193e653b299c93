#!/bin/bash
# builds A/B variants of libpt_hip.so into build/ab/: build_ab.sh name "-DFLAG1 -DFLAG2" [name2 "flags2" ...]
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/ab
D=$R/pathtracer-0_amd/csrc/hip
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wall -Wno-unused-value $flags -o $R/build/ab/$name.so $D/pt_hip.hip $D/pt_bvh.hip &
done
wait
ls -la $R/build/ab/
