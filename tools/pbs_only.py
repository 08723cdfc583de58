#!/usr/bin/env python3
"""bench.py's per_block_size leg alone in a fresh process (is the in-bench figure held back by what ran before it?)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
import bench
import jampack_amd as jam
from jampack_amd import corpus
dev = torch.device("cuda", 0)
r = bench.per_block_size(jam, corpus, torch, dev, 0, 8)
for k, v in r.items():
    if isinstance(v, dict):
        print(k, v["compress_MBps"], v["decompress_MBps"], v["same_bytes"])
