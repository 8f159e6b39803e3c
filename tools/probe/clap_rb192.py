"""Stage-1 LN + q|k|v row-block launch (htsat_rowblock_kernel<192, 0>) alone, 512 clips: ADT_HTSAT_RB192_TPC=1 halves the LDS-DMA chunk."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd.clap_encoder import HtsatEncoder, random_init_clap_model, rowblock

dev = "cuda:0"
model = random_init_clap_model(0)
enc = HtsatEncoder(model.state_dict(), model.config.audio_config, dev)
L = enc.stages[1]["layers"][0]
C = 192
x = torch.randn(512 * 1024, C, device=dev)
qkv = torch.empty(512 * 1024, 3 * C, dtype=torch.bfloat16, device=dev)
def run():
    rowblock(0, x, L["qkv_pk"], 3 * C // 32, L["bqkv"], ln=L["ln1"], eps=1e-5, out16=qkv)
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print(f"tpc={os.environ.get('ADT_HTSAT_RB192_TPC', '2')}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", float(qkv.float().abs().mean()))
