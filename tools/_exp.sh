cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_attention_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_kernels.py attn 2>&1 | grep attn
