# usage (GPU box): bash tools/run_r06_layer_ab.sh  -- a third-stage HTSAT layer in ONE launch (adt_htsat_layer_block, ADT_HTSAT_LAYER384=1, the default)
# against attention half + MLP half (=0), alternating, one library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for rep in 1 2 3; do
  for v in 0 1; do
    ADT_HTSAT_LAYER384=$v timeout -k 10 300 python $R/bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline > $O/clap_layer$v.json 2> $O/clap_layer$v.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$O/clap_layer$v.json')); print('rep $rep ADT_HTSAT_LAYER384=$v: embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline'].get('kernel_ms'),3))"
  done
done
