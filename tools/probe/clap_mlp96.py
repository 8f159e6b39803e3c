"""Stage-0 MLP half (htsat_mlp_kernel<96>) alone, 512 clips: ADT_HTSAT_MLP96_SPC=2 doubles the LDS-DMA chunk (two workgroups per CU instead of three: +4 %)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd.clap_encoder import HtsatEncoder, random_init_clap_model, rowblock

dev = "cuda:0"
model = random_init_clap_model(0)
enc = HtsatEncoder(model.state_dict(), model.config.audio_config, dev)
L = enc.stages[0]["layers"][0]
x = torch.randn(512 * 4096, 96, device=dev)
def run():
    rowblock(2, x, L["mlp_pk"], 96 // 8, L["b1"], ln=L["ln2"], eps=1e-5, bias2=L["b2"])
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print(f"spc={os.environ.get('ADT_HTSAT_MLP96_SPC', '1')}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", float(x.abs().mean()))
