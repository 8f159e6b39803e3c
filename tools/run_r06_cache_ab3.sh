# usage: bash tools/run_r06_cache_ab3.sh "<libs>"  -- correctness (GEMM tests) then the step bench per build, alternating, with final_loss
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_cache_ab
mkdir -p $O
cd $R
LIBS=${1:-"hip exp_c exp_c1 exp_c2 exp_c3 exp_c5"}
F=$O/gemm_store_policy_ab3.txt
for v in $LIBS; do
  echo "== tests lib $v" >> $F
  ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -1 >> $F
done
for rep in 1 2; do
  for v in $LIBS; do
    echo "== step rep $rep lib $v" >> $F
    ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'roofline_ms', d['roofline'].get('kernel_ms'), d['roofline']['frac'], 'loss', d['final_loss'], 'sclk', d['clock']['sclk_mhz']['median'])" >> $F
  done
done
cat $F
