"""Dev probe: does a preceding training engine change the generation step time?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.gen_bench import run
from tools import quick_bench
mode = sys.argv[1]
if mode == "train_first":
    quick_bench.run("bf16", steps=1)
    torch.cuda.empty_cache()
elif mode == "alloc_first":
    x = torch.empty(16 << 30, dtype=torch.uint8, device="cuda:0"); x.zero_(); torch.cuda.synchronize(); del x
    torch.cuda.empty_cache()
run("bf16", 4)
