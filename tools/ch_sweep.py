import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gims_amd import synth
from gims_amd.carhynet import CARHyNet
torch.set_grad_enabled(False)
m = CARHyNet().eval(); m.load_state_dict(synth.make_carhynet_state_dict(321))
base = synth.make_patches(256, 5)
n = 16384
patches = torch.from_numpy(np.tile(base, (n // 256 + 1, 1, 1, 1))[:n]).cuda()
x = patches.permute(0, 3, 1, 2)
for chunk in (1024, 2048, 4096, 8192, 16384):
    m.chunk = chunk
    for _ in range(2): m(x)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("chunk %5d: %.3f ms  %.3f M patches/s" % (chunk, 1e3 * sorted(ts)[2], n / sorted(ts)[2] / 1e6), flush=True)
