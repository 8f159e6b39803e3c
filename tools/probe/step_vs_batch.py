import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from adt_str_amd.network import ADTModel, ADTModelConfig
from adt_str_amd.trainer import FlatTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = ADTModelConfig(input_sec=10.0, time_res=0.01, win_length=2048, sample_rate=16000, dropout=0.1, plain=True, **bench.SETTING1)
model = ADTModel(cfg).to(dev).train()
tr = FlatTrainer(model, lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, total_steps=10000, warmup_ratio=0.1)
rng = np.random.default_rng(0)
for B in (64, 48, 32, 16, 8):
    tok, tl = bench.synthetic_tokens(rng, B, 128)
    tokens, lens = torch.from_numpy(tok).to(dev), torch.from_numpy(tl).to(dev)
    wavs = torch.randn(B, 160000, device=dev) * 0.1
    for _ in range(3): tr.train_step(wavs, tokens, lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.train_step(wavs, tokens, lens)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"batch {B}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} clips/s (B x F = {B * 986}, % 64 = {B * 986 % 64})", flush=True)
