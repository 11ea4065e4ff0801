#!/usr/bin/env python3
"""The eval_stream leg of bench.py on its own (serial / pipelined / prefetch_thread schedules).  usage: eval_stream.py [images]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
r = bench.eval_stream_bench(dcl, torch.device("cuda:0"), images=int(sys.argv[1]) if len(sys.argv) > 1 else 40)
for tag in r:
    print(tag, {k: v["ms_per_image"] for k, v in r[tag].items() if isinstance(v, dict) and "ms_per_image" in v}, r[tag]["stages_alone_ms"])
