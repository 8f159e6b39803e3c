# usage: bash tools/run_r06_x3.sh  -- the split-bf16 parity arm on the persistent kernels: tests, bench line, kernel stats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_x3fast
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_precision_gpu.py tests/test_gemm_gpu.py tests/test_abi.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout -k 10 300 python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-clap --no-clock > $O/bench_x3.json 2> $O/bench_x3.err; cut -c1-300 $O/bench_x3.json
ADT_X3_TILED=1 timeout -k 10 300 python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-clap --no-clock > $O/bench_x3_tiled.json 2> /dev/null; cut -c1-300 $O/bench_x3_tiled.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --precision bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-clap --no-clock > $O/prof.log 2>&1
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_x3.csv; head -16 $f | cut -c1-160
rm -rf $O/prof
