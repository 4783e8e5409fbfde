import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import cactus_amd.hip as H
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(1)
print(json.dumps(bench.qwen3_decode_via_graph(H, gen), indent=1))
