cd $GRAFT_REPO_ROOT
for m in split fused split fused; do ADT_ATTN_BWD=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-clap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],3), round(d['value'],1))"; done
