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
