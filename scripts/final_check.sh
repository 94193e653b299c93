cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02y
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r02y/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02y/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02y/bench_default_c3.json 2> gpurun_out/r02y/bench_default_c3.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r02y/bench_default_c3.json
