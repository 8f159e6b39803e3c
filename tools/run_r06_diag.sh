# usage: bash tools/run_r06_diag.sh "<libs>"  -- per-kernel stats of the training step under alternative builds + GEMM tests on them
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_diag
mkdir -p $O
LIBS=${1:-"hip exp_c5"}
cd $R
for v in $LIBS; do
  ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_network_gpu.py -x -q -m gpu > $O/pytest_$v.txt 2>&1; tail -2 $O/pytest_$v.txt
done
cd /tmp && export TMPDIR=/tmp
for v in $LIBS; do
  export ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm --no-e2e > $O/prof_$v.log 2>&1
  grep -o '"final_loss": [0-9.]*' $O/prof_$v.log | head -1
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$v.csv
  rm -rf $O/prof_$v
done
ls $O
