set -x
mkdir -p gpurun_out
./tools/probe/probe_mfma > gpurun_out/probe.log 2>&1; cat gpurun_out/probe.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1; tail -3 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*stats*" | head
