#!/usr/bin/env python3
"""GPU probe: Sinkhorn loop timing (event vs wall) on the default stream vs a side stream."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gims_amd import hip

def run(n, np_, iters, stream=None, reps=3):
    items = []
    for _ in range(np_):
        z = torch.randn(n, n, device="cuda") * 4
        items.append(dict(scores=z, n=n, m=n, matches0=torch.empty(n, dtype=torch.int64, device="cuda"),
                          matches1=torch.empty(n, dtype=torch.int64, device="cuda"), mscores0=torch.empty(n, device="cuda"),
                          mscores1=torch.empty(n, device="cuda"), uv=torch.empty(2 * n + 3, device="cuda")))
    probs = hip.make_ot_problems(items)
    work = torch.empty(hip.sinkhorn_workspace_bytes(probs), dtype=torch.uint8, device="cuda")
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    res = []
    with ctx:
        for r in range(reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); a.record()
            hip.sinkhorn_match(probs, 1.0, iters, 0.2, work)
            t1 = time.perf_counter(); b.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
            res.append((a.elapsed_time(b), (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    return res

if __name__ == "__main__":
    hip.load()
    for n, np_ in ((4096, 2), (1024, 16), (1024, 1)):
        for name, st in (("default-stream", None), ("side-stream", torch.cuda.Stream())):
            r = run(n, np_, 100, st)
            print(f"n={n} x{np_} {name:15s} event_ms/host_enqueue_ms/wall_ms: " + "  ".join(f"{e:.2f}/{h:.2f}/{w:.2f}" for e, h, w in r), flush=True)
