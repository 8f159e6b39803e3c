import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adt_str_amd.network import ADTModel, ADTModelConfig
torch.manual_seed(0)
m = ADTModel(ADTModelConfig(input_sec=10.0, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=4, dec_layers=4, nhead=6, d_query=128,
                            dropout=0.1, tgt_vocab_size=1400, plain=True, n_mels=128)).cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
src = torch.randn(B, 160000, device="cuda") * 0.1
eng = m.engine
mem16, B, S = eng.encode(src)
out = eng.greedy_decode_cached(mem16, B, S, 101, 2, -1, use_graph=False)
torch.cuda.synchronize()
