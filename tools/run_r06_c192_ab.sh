# usage (GPU box): bash tools/run_r06_c192_ab.sh  -- builds of the C = 192 layer kernel (libadt_exp_head.so = the previous commit) against the in-tree library,
# alternating: the attention half alone (tools/probe/attn_big.py) and the tower
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
HEAD=$R/adt_str_amd/libadt_exp_head.so; NEW=$R/adt_str_amd/libadt_hip.so
for rep in 1 2 3; do
  for v in head new; do
    L=$HEAD; [ $v = new ] && L=$NEW
    [ $rep = 1 ] && ADT_LIB_PATH=$L timeout -k 10 120 python $R/tools/probe/attn_big.py 2>&1 | grep "C=192" | grep -v "update max"
    ADT_LIB_PATH=$L timeout -k 10 300 python $R/bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline > $O/clap_$v.json 2> $O/clap_$v.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$O/clap_$v.json')); print('rep $rep $v: embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'tower_ms', round(d['roofline'].get('kernel_ms'),3))"
  done
done
