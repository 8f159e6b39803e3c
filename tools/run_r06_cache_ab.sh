# usage: bash tools/run_r06_cache_ab.sh  -- cache-policy arms of the persistent NT kernel (tools/build_variant.sh a/c/ac/b), alternating processes on one box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_cache_ab
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in hip exp_a exp_c exp_ac exp_b; do
    echo "== rep $rep lib $v" >> $O/gemm_cache_policy_ab.txt
    ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 200 python tools/exp_gemm_tile.py 2>&1 | grep -v "^ADT_LIB_PATH\|override" >> $O/gemm_cache_policy_ab.txt
  done
done
for rep in 1 2; do
  for v in hip exp_a exp_c exp_ac; do
    echo "== step rep $rep lib $v" >> $O/gemm_cache_policy_ab.txt
    ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-clap --no-fp32-arm --no-parity-arm --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'roofline_ms', d['roofline'].get('kernel_ms'), d['roofline']['frac'])" >> $O/gemm_cache_policy_ab.txt
  done
done
cat $O/gemm_cache_policy_ab.txt
