# usage: bash tools/run_r06_clap_ab.sh  -- the C = 384 MLP in one launch (ADT_HTSAT_MLP384=1, default) against LN -> fc1 -> GELU + fc2 GEMM (=0): tests, kernels, CLAP bench line, alternating
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_clap
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_htsat_fused_gpu.py tests/test_clap_encoder_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 200 python tools/probe/rowblock384.py 2>&1 | grep -v amdgpu.ids | tee $O/rowblock384.txt
for rep in 1 2; do
  for v in 0 1; do
    echo "== rep $rep ADT_HTSAT_MLP384=$v"
    ADT_HTSAT_MLP384=$v timeout -k 10 300 python bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
  done
done 2>&1 | tee $O/clap_mlp384_ab.txt
