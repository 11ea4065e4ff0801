"""bench.py's `eval_stream` leg alone (crop builder -> forward -> ADD-S -> table, serial / pipelined / builder thread)."""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
print(json.dumps(bench.eval_stream_bench(dcl, torch.device("cuda:0")), indent=1))
