import os, sys, torch
sys.path.insert(0, '/root/repo')
os.environ["GIMS_ATTN_PROF"] = "1"
from gims_amd import hip
n, pairs = 4096, 8
rows = 2 * n * pairs
g = torch.Generator().manual_seed(1)
x = torch.randn(rows, 768, generator=g) * 0.3
x[:, :256] *= hip.ATTN_Q_SCALE
probs = []
for p in range(pairs):
    o = 2 * n * p
    probs += [(o, n, o + n, n), (o + n, n, o, n)]
pr = torch.tensor(probs, dtype=torch.int32, device="cuda")
out = torch.empty((rows, 512), dtype=torch.bfloat16, device="cuda")
for name, t, f16 in (("bf16", x.to(torch.bfloat16), False), ("f16", x.to(torch.float16).view(torch.bfloat16), True))[: int(os.environ.get("PROBE_N", "2"))]:
    q = t.cuda()
    for _ in range(2):
        print("----", name, file=sys.stderr, flush=True)
        hip.attention(q, pr, n, 4, None, out_split=out, q_prescaled=True, f16=f16)
torch.cuda.synchronize()
