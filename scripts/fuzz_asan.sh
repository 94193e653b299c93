#!/bin/bash
# HOST-side AddressSanitizer run of the api fuzz: libpt_hip.so rebuilt with -fsanitize=address -fno-gpu-sanitize (device code and code objects untouched: no xnack, no GPU ASan),
# the ASan runtime preloaded into python.  Looks for a host heap overflow in the library's scheduler code.   fuzz_asan.sh <tag> <first> <count>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzasan}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
[ -f build/libpt_hip_asan.so ] || { echo "build/libpt_hip_asan.so missing"; exit 1; }
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:verify_asan_link_order=0:symbolize=1:log_path=$O/asan
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
PT_HIP_LIB=$R/build/libpt_hip_asan.so LD_PRELOAD=$RT API_FUZZ_TRACE=$O/ops.txt timeout -k 10 1000 python3 -X faulthandler scripts/api_fuzz.py ${2:-100001} ${3:-1500} > $O/out.txt 2>&1; echo "rc=$?"
tail -15 $O/out.txt | cut -c1-200
ls $O; for f in $O/asan*; do [ -f "$f" ] && head -80 "$f" | cut -c1-220; done
