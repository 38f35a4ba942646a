#!/usr/bin/env python3
"""Does a HIP graph of the step's kernel sequence beat eager launches at the 8192-row shard?  forward_backward +
SGD/prox (no Adam: its bias corrections are host-computed kernel arguments) captured with torch.cuda.CUDAGraph and
replayed, against the same sequence launched eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
eng, X = bench.init_engine(dev, rows, seed=100)
eps = torch.randn(1, rows, bench.K, device=dev)

def seq():
  eng.forward_backward(X, eps, 1.0, lam=bench.LAM)
  eng.sgd_prox_step(bench.WS_LR, bench.WS_LR)

for _ in range(3):
  seq()
torch.cuda.synchronize()

def timeit(fn, n=50):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e3

print("eager  %.4f ms" % timeit(seq))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  seq()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
  seq()
torch.cuda.synchronize()
print("graph  %.4f ms" % timeit(g.replay))
print("eager  %.4f ms" % timeit(seq))
