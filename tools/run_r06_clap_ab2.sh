# usage: bash tools/run_r06_clap_ab2.sh  -- the attention half of stages 1 / 2 in one launch (ADT_HTSAT_ATTN_BIG=192 / 384 / 1) against the three launches (=0): CLAP bench line, alternating
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
  for v in 0 192 384 1; do
    echo -n "rep $rep ADT_HTSAT_ATTN_BIG=$v: "
    ADT_HTSAT_ATTN_BIG=$v timeout -k 10 300 python bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline']['kernel_ms'],3))"
  done
done
timeout -k 10 300 env ADT_HTSAT_ATTN_BIG=1 python -m pytest tests/test_clap_encoder_gpu.py -x -q -m gpu 2>&1 | tail -2
