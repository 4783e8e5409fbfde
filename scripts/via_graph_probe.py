import sys, json
sys.path.insert(0, '/root/repo')
import torch
import cactus_amd.hip as H
H.init(0)
import bench
gen = torch.Generator(device="cuda").manual_seed(1)
print(json.dumps(bench.qwen3_decode_via_graph(H, gen, 512, 24), indent=1))
