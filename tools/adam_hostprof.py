import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
from gims_amd import GMatcher, synth
from gims_amd.optim import Adam
m = GMatcher({}); m.load_state_dict(synth.make_state_dict(123)); m = m.cuda().train()
ps = list(m.parameters())
for which in ("fused", "torch"):
    opt = Adam(ps, lr=1e-4) if which == "fused" else torch.optim.Adam(ps, lr=1e-4)
    def once():
        for p in ps: p.grad = torch.ones_like(p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.step(); t1 = time.perf_counter(); opt.zero_grad(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2
    for _ in range(3): once()
    r = [once() for _ in range(10)]
    print(which, "step host %.3f ms, zero_grad %.3f ms, drain %.3f ms" % tuple(1e3 * sorted(x[i] for x in r)[5] for i in range(3)))
opt = Adam(ps, lr=1e-4)
for p in ps: p.grad = torch.ones_like(p)
opt.step()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): opt.step()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
