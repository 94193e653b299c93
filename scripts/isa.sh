#!/bin/bash
# ISA statistics of the production intersect kernel (k_extend_persist<false, short, 256, false>) for a set of -D flags:  isa.sh tag "-DFLAGS"
R=$(cd $(dirname $0)/.. && pwd)
T=/tmp/isa_$1; mkdir -p $T; cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -c --save-temps $2 $R/pathtracer-0_amd/csrc/hip/pt_hip.hip -o pt_hip.o 2>/dev/null
S=pt_hip-hip-amdgcn-amd-amdhsa-gfx950.s
K=${3:-_ZN12_GLOBAL__N_116k_extend_persistILb0EsLi256ELb0EEEvN3ptd8DevSceneENS_5StateEPKjiiPNS_7ControlEiiii}
a=$(grep -n "^$K:" $S | cut -d: -f1); b=$(grep -n "\.amdhsa_kernel $K" $S | cut -d: -f1)
sed -n "${a},${b}p" $S > k.s
echo "$1: lines $(wc -l < k.s) VALU $(grep -c '^\s*v_' k.s) (v_mov $(grep -c '^\s*v_mov' k.s), v_pk $(grep -c '^\s*v_pk' k.s)) SALU $(grep -c '^\s*s_' k.s) DS $(grep -c '^\s*ds_' k.s) VMEM $(grep -c '^\s*global_' k.s) waitcnt $(grep -c 's_waitcnt' k.s)"
grep -A30 "\.amdhsa_kernel $K" $S | grep -E "next_free_vgpr|next_free_sgpr|group_segment" 
