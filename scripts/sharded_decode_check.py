#!/usr/bin/env python3
"""Runs bench.sharded_decode under a one-rank RCCL group (the per-rank work of a 2-GPU run) to check
the code path on a one-GPU box:  python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1
--nproc-per-node 1 scripts/sharded_decode_check.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import cactus_amd.hip as H
from bench import sharded_decode
H.init(0)
gen = torch.Generator(device="cuda").manual_seed(42)
print(json.dumps(sharded_decode(H, gen, 2)))
dist.destroy_process_group()
