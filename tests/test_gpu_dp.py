"""The overlapped two-bucket gradient all-reduce on the GPU: a one-rank RCCL
group exercises the stream hand-offs of parallel.BucketedGradSync (collective on
the communicator's stream while the encoder backward runs, compute stream
waiting before the optimizers); the values must be what the plain step gives.
The multi-rank arithmetic itself is covered by the gloo test in test_host_logic."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from oi_vae_amd import parallel
from tests.test_gpu_scale import _engine
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[1], rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
eng = _engine(16, 64, 64, 64, 64, seed=3)
g = torch.Generator(device="cuda").manual_seed(11)
B = 4096
X = torch.rand(B, 16 * 64, generator=g, device="cuda")
eps = torch.randn(1, B, 64, generator=g, device="cuda")
out = eng.forward_backward(X, eps, 1.0)
ref = eng.flat_grads.clone(); ref_s = out["scalars"].clone()
for rep in range(3):
  eng.flat_grads.zero_()
  sync = parallel.BucketedGradSync(eng, force=True)
  assert sync.active and sync.enc.numel() + sync.dec.numel() == eng.flat_grads.numel()
  out = eng.forward_backward(X, eps, 1.0, grad_sync=sync)
  sync.wait()
  eng.adam_step(0.0)   # consumers after wait() must see the reduced buffer
  torch.cuda.synchronize()
  cut = eng.seg_off["gen_W1"]
  assert torch.equal(eng.flat_grads[cut:], ref[cut:]), "decoder-side gradients changed"
  err = ((eng.flat_grads[:cut] - ref[:cut]).abs().max() / ref[:cut].abs().max()).item()
  assert err < 1e-5, err      # encoder gradients go through the dz atomics: order-dependent in the last bits
  assert torch.allclose(out["scalars"][:6], ref_s[:6], rtol=1e-6)
dist.destroy_process_group()
print("ok")
'''


def test_bucketed_grad_sync_on_one_rank_rccl(tmp_path):
  script = tmp_path / "dp1.py"
  script.write_text(_WORKER % {"root": ROOT})
  port = str(29600 + (os.getpid() % 2000))
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  r = subprocess.run([sys.executable, str(script), port], capture_output=True, text=True, timeout=110, env=env)
  assert r.returncode == 0, r.stdout + r.stderr


_WORKER2 = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from oi_vae_amd import parallel
from tests.test_gpu_scale import _engine
rank, port = int(sys.argv[1]), sys.argv[2]
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=2)
G, B, K, steps = 6, 512, 64, 6
g = torch.Generator(device="cuda").manual_seed(21)
Xs = [torch.rand(B, G * 64, generator=g, device="cuda") for _ in range(steps)]
Es = [torch.randn(1, B, K, generator=g, device="cuda") for _ in range(steps)]
lo, hi = parallel.shard_rows(B, rank, 2)
t_prox = 7.5   # rows of Ws ~ N(0,1)^64 have norm ~8: a good third of them dies at the first step

def run(eng, rows, sync_on):
  scal = []
  sync = parallel.BucketedGradSync(eng) if sync_on else None   # one object for all steps, as bench.py uses it
  for X, eps in zip(Xs, Es):
    out = eng.forward_backward(X[rows].contiguous(), eps[:, rows].contiguous(), 1.0, lam=1.0, grad_sync=sync)
    sc = out["scalars"].clone()
    if sync_on:   # per-bucket optimizers, as bench.py runs them: the decoder side under the encoder bucket's all-reduce
      sync.wait_decoder()
      eng.adam_step(1e-3, part="dec")
      eng.sgd_prox_step(1e-3, t_prox)
      sync.wait_encoder()
      eng.adam_step(1e-3, part="enc")
      parallel.reduce_scalars_(sc)
    else:         # one Adam launch over everything: the split form must give the same bits
      eng.adam_step(1e-3)
      eng.sgd_prox_step(1e-3, t_prox)
    scal.append(sc[:6].double().cpu())
  torch.cuda.synchronize()
  return torch.stack(scal)

eng = _engine(G, 64, K, 64, 64, seed=5)
sc_dp = run(eng, slice(lo, hi), True)
mine = eng.flat_params.clone()
# both ranks must hold bit-identical parameters (and so the same zero set) after every step's all-reduce
both = [torch.empty_like(mine), torch.empty_like(mine)]
dist.all_gather(both, mine)
assert torch.equal(both[0], both[1]), "ranks diverged"
if rank == 0:
  ref = _engine(G, 64, K, 64, 64, seed=5)
  sc_1 = run(ref, slice(0, B), False)
  assert float(((sc_dp - sc_1).abs() / sc_1.abs().clamp(min=1.0)).max()) < 1e-5, (sc_dp, sc_1)
  for n in ref.seg_off:
    o, m = ref.seg_off[n], ref.seg_numel[n]
    if m:
      a, b = mine[o:o + m], ref.flat_params[o:o + m]
      assert float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)) < 2e-5, n
  wa, wb = eng.p["Ws"], ref.p["Ws"]
  dead_a, dead_b = (wa == 0).all(-1), (wb == 0).all(-1)
  assert int(dead_b.sum()) > 0, "the threshold should kill some rows"
  assert int((dead_a != dead_b).sum()) <= 2     # a row within rounding of the threshold may fall either way
dist.barrier()
dist.destroy_process_group()
print("ok")
'''


def test_two_ranks_on_one_gpu_match_the_single_rank_step(tmp_path):
  """World size 2 over gloo with both ranks on cuda:0: each rank runs the real kernels on its half of the batch,
  the two gradient buckets are averaged as in a multi-GPU run, optimizers and the proximal step follow.  Both ranks
  must end with bit-identical parameters, and those must be the single-rank full-batch step's within rounding."""
  script = tmp_path / "dp2.py"
  script.write_text(_WORKER2 % {"root": ROOT})
  port = str(31600 + (os.getpid() % 2000))
  env = dict(os.environ)
  procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            text=True, env=env) for r in range(2)]
  outs = []
  for p in procs:
    try:
      o, _ = p.communicate(timeout=110)
    except subprocess.TimeoutExpired:
      p.kill()
      o, _ = p.communicate()
    outs.append(o)
  assert all(p.returncode == 0 for p in procs), "\n".join(outs)


@pytest.mark.gpu
def test_adam_in_two_parts_gives_the_bits_of_one_launch():
  """Engine.adam_step(part="dec") then (part="enc") -- the per-bucket form of the data-parallel step -- against one
  launch over every Adam parameter, three steps with moving gradients."""
  import torch
  from tests.test_gpu_scale import _engine
  a, b = _engine(3, 64, 64, 64, 64, seed=8), _engine(3, 64, 64, 64, 64, seed=8)
  g = torch.Generator(device="cuda").manual_seed(4)
  for _ in range(3):
    grads = torch.randn(a.flat_grads.shape, generator=g, device="cuda")
    a.flat_grads.copy_(grads); b.flat_grads.copy_(grads)
    a.adam_step(1e-2)
    b.adam_step(1e-2, part="dec")
    b.adam_step(1e-2, part="enc")
  torch.cuda.synchronize()
  assert a.adam_t == b.adam_t == 3
  assert torch.equal(a.flat_params, b.flat_params)
  assert torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)


def test_bench_starts_its_own_ranks_for_n_gpus():
  """`python bench.py --gpus 2` with no launcher around it (the form the driver uses): bench.py starts the two ranks
  itself, as fresh children under torch.distributed.run, and rank 0 prints the line.  Rehearsal mode
  (OIVAE_BENCH_REHEARSE=1: both ranks on cuda:0 over gloo) makes that runnable on a one-GPU box: a code-path check of
  the N > 1 bench -- sharding, the two gradient buckets, barriers, max over ranks -- whose numbers mean nothing."""
  import json
  env = dict(os.environ, OIVAE_BENCH_REHEARSE="1")
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                     capture_output=True, text=True, timeout=380, env=env)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
  assert line["n_gpus"] == 2 and line["config"]["rows_per_gpu"] == 32768 and line["scaling"] == "strong"
  assert "rehearsal" in line and line["comm"]["backend"] == "gloo"
  assert line["value"] > 0 and set(line["comm"]["exposed_wait_ms"]) == {"decoder_side", "encoder"}
