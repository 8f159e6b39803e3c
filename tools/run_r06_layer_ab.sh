# usage (GPU box): bash tools/run_r06_layer_ab.sh [192|384]  -- an HTSAT layer of stage 2 (C = 384, the default) or stage 1 (C = 192) in ONE launch
# (adt_htsat_layer_block, ADT_HTSAT_LAYER<C>=1, the default) against attention half + MLP half (=0), alternating, one library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; CC=${1:-384}
for rep in 1 2 3; do
  for v in 0 1; do
    env ADT_HTSAT_LAYER$CC=$v timeout -k 10 300 python $R/bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline > $O/clap_layer$v.json 2> $O/clap_layer$v.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$O/clap_layer$v.json')); print('rep $rep ADT_HTSAT_LAYER$CC=$v: embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline'].get('kernel_ms'),3))"
  done
done
