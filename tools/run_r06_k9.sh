# usage: bash tools/run_r06_k9.sh  -- K9 on K1's second-generation FFT passes (product build) against the first-generation kernel (libadt_exp_k9old.so): tests, kernel time, CLAP line
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests/test_clap_frontend.py tests/test_clap_encoder_gpu.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2; do for v in exp_k9old hip; do
  echo -n "rep $rep lib $v: "
  ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 300 python - <<'PY' 2>&1 | grep -v "ADT_LIB_PATH\|amdgpu"
import numpy as np, torch
from adt_str_amd.clap_frontend import ClapLogMel
rng = np.random.default_rng(7)
clips = [torch.from_numpy((rng.standard_normal(int(n)) * 0.2).astype(np.float32)).cuda() for n in rng.integers(4800, 96001, 512)]
fe = ClapLogMel("cuda:0")
for _ in range(3): m = fe.mel(clips)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m = fe.mel(clips)
e1.record(); torch.cuda.synchronize()
print(f"mel of 512 clips {e0.elapsed_time(e1) / 10:.3f} ms; checksum {float(m.double().sum()):.6e}")
PY
done; done
for rep in 1 2; do for v in exp_k9old hip; do
  echo -n "rep $rep lib $v: "
  ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 300 python bench.py --workload clap --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('embeds/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3))"
done; done
