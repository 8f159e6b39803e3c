cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_attention_gpu.py -m gpu -x -q 2>&1 | tail -3
cp adt_str_amd/libadt_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in A B; do
  cp _exp/lib$v.so adt_str_amd/libadt_hip.so
  echo "== $v"
  python tools/bench_kernels.py attn 2>&1 | grep "attn"
done
done
cp /tmp/lib_orig.so adt_str_amd/libadt_hip.so
