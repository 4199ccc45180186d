#!/usr/bin/env python3
"""GPU probe: where the HOST time of one pair through forward() goes (cProfile over 40 calls).   python tools/b1_hostprof.py 4096"""
import cProfile, os, pstats, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gims_amd import GMatcher, synth
from helpers import pair_to_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = GMatcher({}).eval(); m.load_state_dict(synth.make_state_dict(123))
pair = synth.make_pair(n, 1000)
datas = [pair_to_data(pair, 15, 2, 7, device="cuda") for _ in range(45)]
for d in datas[:5]:
    m(d)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for d in datas[5:]:
    m(d)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
