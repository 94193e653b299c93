run() { bash scripts/ab.sh -r 1 -t -f "$*" final; }
run ""
run --extend-blocks-per-cu 5 --extend-cache 20480
run --extend-blocks-per-cu 6 --extend-cache 12288
run --extend-blocks-per-cu 6 --extend-cache 16384
run --extend-blocks-per-cu 7 --extend-cache 12288
run --extend-blocks-per-cu 7 --extend-cache 8192
run --extend-blocks-per-cu 8 --extend-cache 8192
run --refill-min 16
run --refill-min 32
run --none-min 1
run --none-min 4
run ""
