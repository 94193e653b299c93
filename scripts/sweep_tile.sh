run() { v=$1; shift; bash scripts/ab.sh -r 1 -t -f "$*" $v; }
PT_HIP_LIB=build/ab/tpb1024.so timeout -k 10 120 python3 scripts/asm_check.py C3 65536 extend_blocks_per_cu=4 extend_cache_bytes=114688 | tail -1
for s in 2 1; do
run base --streams $s
run base --streams $s --extend-blocks-per-cu 4 --extend-cache 28000
run base --streams $s --extend-blocks-per-cu 3 --extend-cache 40000
run base --streams $s --extend-blocks-per-cu 2 --extend-cache 65536
run tpb512 --streams $s --extend-blocks-per-cu 8 --extend-cache 16384
run tpb512 --streams $s --extend-blocks-per-cu 6 --extend-cache 28672
run tpb512 --streams $s --extend-blocks-per-cu 4 --extend-cache 57344
run tpb1024 --streams $s --extend-blocks-per-cu 8 --extend-cache 32768
run tpb1024 --streams $s --extend-blocks-per-cu 4 --extend-cache 114688
run tpb1024 --streams $s --extend-blocks-per-cu 4 --extend-cache 65536
done
