#!/usr/bin/env python3
"""bench.py's `evaluators` block alone (one JSON object), for A/B runs with PXL_LIB_PATH."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
r = bench.gpu_evaluators(dev)
print(json.dumps({k: (v.get("ms"), v.get("frac")) for k, v in r.items()}))
