# usage (GPU box): bash tools/run_r06_small_ab.sh  -- the small launches around the CLAP tower (clips read through a pointer table instead of a concatenation pass,
# final LayerNorm + token mean in one pass, cosine arg-max with the row in registers): libadt_exp_head.so + the previous Python (git stash is not available on
# the box: the previous tree is a copy under gpurun_out/prev) against the tree, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for rep in 1 2 3; do
  for v in prev new; do
    D=$R; [ $v = prev ] && D=$R/gpurun_prev
    (cd $D && timeout -k 10 300 python bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline > $O/clap_small_$v.json 2> $O/clap_small_$v.err) || exit 1
    python3 -c "import json,sys; d=json.load(open('$O/clap_small_$v.json')); print('rep $rep $v: embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline'].get('kernel_ms'),3))"
  done
done
