#!/usr/bin/env python3
"""Diagnostic: HIP-event time per corr-lookup launch (bench.lookup_burst_us) for 1 and 4 sequences per launch.
TCS_LOOKUP_LPB = 1 / 2 / 4 forces the pyramid levels per workgroup, TCS_LOOKUP_ORDER = 1 the level-major grid order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
import bench
dev = torch.device("cuda:0")
with torch.no_grad():
    for B in (1, 4):
        print("levels per workgroup", os.environ.get("TCS_LOOKUP_LPB", "auto"), "| sequences", B, "| us per launch",
              round(bench.lookup_burst_us(dev, B), 3), flush=True)
