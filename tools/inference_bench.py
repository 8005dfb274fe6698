"""Forward-only frames/s (bench.inference_measure) for the given configs, eager and as HIP-graph replays: python tools/inference_bench.py C3 C2 C1"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda", 0)
for c in sys.argv[1:] or ["C3"]:
    for g in (False, True):
        r = bench.inference_measure(c, 200, dev, cameras=16, graph=g)
        print(json.dumps(r), flush=True)
