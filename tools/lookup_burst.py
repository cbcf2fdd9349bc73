import os, sys
sys.path.insert(0, "/root/repo")
import tcs_paths; tcs_paths.add_product_path()
import torch, bench
dev = torch.device("cuda:0")
with torch.no_grad():
    for B in (1, 4):
        print("LPB", os.environ.get("TCS_LOOKUP_LPB", "4"), "B", B, "us per launch", round(bench.lookup_burst_us(dev, B), 3), flush=True)
