# usage (GPU box): bash tools/run_r06_resid_ab.sh  -- the residual carried in the output accumulators of the fused HTSAT kernels (no second read of the
# token rows): libadt_exp_head.so (before) against the in-tree library, alternating, alone and inside the tower
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
HEAD=$R/adt_str_amd/libadt_exp_head.so; NEW=$R/adt_str_amd/libadt_hip.so
for rep in 1 2; do
  for v in head new; do
    L=$HEAD; [ $v = new ] && L=$NEW
    echo "== rep $rep $v"
    ADT_LIB_PATH=$L timeout -k 10 120 python $R/tools/probe/attn_big.py 2>&1 | grep "C=" | grep -v "update max" || exit 1
    ADT_LIB_PATH=$L timeout -k 10 120 python $R/tools/probe/rowblock384.py 2>&1 | grep "whole MLP" | sed "s/.*whole MLP/whole MLP/" || exit 1
    ADT_LIB_PATH=$L timeout -k 10 300 python $R/bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline > $O/clap_$v.json 2> $O/clap_$v.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$O/clap_$v.json')); print('embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', d['roofline'].get('kernel_ms'))"
  done
done
