import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adt_str_amd.network import ADTModel, ADTModelConfig
torch.manual_seed(0)
m = ADTModel(ADTModelConfig(input_sec=10.0, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=4, dec_layers=4, nhead=6, d_query=128,
                            dropout=0.1, tgt_vocab_size=1400, plain=True, n_mels=128)).cuda().eval()
for B in (1, 8):
    src = torch.randn(B, 160000, device="cuda") * 0.1
    for L in (64, 256):
        m.sample(src, None, None, max_length=L, end_token=-1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            out = m.sample(src, None, None, max_length=L, end_token=-1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"ADTModel.sample: {B} x 10 s clip, max_length {L}: {dt * 1e3:.1f} ms (log-mel + encoder + {L - 1} decode steps, graph captured per call)", flush=True)
