import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd.network import ADTModel, ADTModelConfig
torch.manual_seed(0)
m = ADTModel(ADTModelConfig(input_sec=10.0, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=4, dec_layers=4, nhead=6, d_query=128,
                            dropout=0.1, tgt_vocab_size=1400, plain=True, n_mels=128)).cuda().eval()
src = torch.randn(8, 160000, device="cuda") * 0.1
eng = m.engine
for mode in ("graph", "cached", "full"):
    for L in (64, 256, 1000):
        if mode == "full" and L > 256:
            continue
        m.sample(src, None, None, max_length=8, use_cache=mode != "full")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "full":
            out = m.sample(src, None, None, max_length=L, end_token=-1, use_cache=False)     # end_token -1: never finishes -> L-1 steps
        else:
            mem16, B, S = eng.encode(src)
            out = eng.greedy_decode_cached(mem16, B, S, L, 2, -1, use_graph=mode == "graph")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{mode:7s} max_length={L}: {dt*1e3:.1f} ms  ({dt/(L-1)*1e3:.3f} ms/step), out {tuple(out.shape)}")
