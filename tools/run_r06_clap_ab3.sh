# usage: bash tools/run_r06_clap_ab3.sh  -- the C = 384 MLP in one launch (ADT_HTSAT_MLP384=1) against two launches (=0) with the attention halves fused: kernels alone, tests, CLAP bench line alternating
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python tools/probe/rowblock384.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
  for v in 0 1; do
    echo -n "rep $rep ADT_HTSAT_MLP384=$v: "
    ADT_HTSAT_MLP384=$v timeout -k 10 300 python bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline']['kernel_ms'],3))"
  done
done
ADT_HTSAT_MLP384=1 timeout -k 10 300 python -m pytest tests/test_clap_encoder_gpu.py tests/test_htsat_fused_gpu.py -x -q -m gpu 2>&1 | tail -2
