import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda:0"
dbg = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
os.environ["ADT_ATTN_DBG"] = str(dbg.data_ptr())
from adt_str_amd import kernels as K
B, H, S = 64, 6, 986
d = H * 128
qkv = torch.randn((B * S, 3 * d), device=dev).bfloat16()
scale = 1 / math.sqrt(128)
drop = (0.1, 5) if len(sys.argv) > 1 else None
for _ in range(3):
    o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, scale, drop=drop)
torch.cuda.synchronize()
t = dbg.cpu().view(16, 16).numpy()
names = ["S0 issue+wait", "valu0", "tr wait a", "PVa+tr b", "PVb issue", "S1", "valu1", "tr a", "PVa+tr b", "PVb", "vmcnt", "barrier+flush"]
for j in range(4, 10):
    r = t[j]
    print(j, [int(r[k + 1] - r[k]) for k in range(12)], "tile cycles", int(t[j + 1][0] - r[0]), "ticks", int(t[j + 1][13] - r[13]))
print(names)
