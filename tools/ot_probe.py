#!/usr/bin/env python3
"""GPU probe: Sinkhorn solve time, streamed kernels vs the on-chip resident kernel, plus their agreement."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gims_amd import hip


def make(n, np_, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    items = []
    for _ in range(np_):
        z = (torch.randn(n, (n + 3) // 4 * 4, generator=g) * 4).cuda()
        items.append(dict(scores=z, n=n, m=n, matches0=torch.empty(n, dtype=torch.int64, device="cuda"),
                          matches1=torch.empty(n, dtype=torch.int64, device="cuda"), mscores0=torch.empty(n, device="cuda"),
                          mscores1=torch.empty(n, device="cuda"), uv=torch.empty(2 * n + 3, device="cuda")))
    return items


def run(items, iters, resident, reps=3):
    os.environ["GIMS_OT_RESIDENT"] = str(int(resident))        # 0 streamed, 1 on-chip where the plan says so, 2 on-chip forced
    probs = hip.make_ot_problems(items)
    work = torch.empty(hip.sinkhorn_workspace_bytes(probs), dtype=torch.uint8, device="cuda")
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        hip.sinkhorn_match(probs, 1.0, iters, 0.2, work)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return min(ms), [it["uv"].clone() for it in items], [it["matches0"].clone() for it in items]


if __name__ == "__main__":
    hip.load()
    cases = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(1022, 32), (4096, 8), (4096, 2), (1024, 8), (300, 4), (2000, 6)]
    for n, np_ in cases:
        items = make(n, np_)
        t0, uv0, m0 = run(items, 100, False)
        t1, uv1, m1 = run(items, 100, True)
        du = max(float((a[:-1] - b[:-1]).abs().max()) for a, b in zip(uv0, uv1))
        st = max(float(b[-1]) for b in uv1)
        same = all(torch.equal(a, b) for a, b in zip(m0, m1))
        print(f"n={n} x{np_}: streamed {t0:8.3f} ms   resident {t1:8.3f} ms   max|du,dv| {du:.2e}  status {st}  matches equal {same}", flush=True)
