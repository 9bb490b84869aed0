"""Experiment: the sharded training step with its RCCL all-reduce captured into one hipGraph (1-rank group)."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import queue
from test_parallel_gpu import _graph_worker
class Q:
    def put(self, r):
        (l0, s0), (l1, s1) = r
        print("eager loss", l0, "graph loss", l1)
_graph_worker(36211, Q())
