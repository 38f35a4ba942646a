"""CPU-only tests of the host side: the C-ABI library loads and exports what
include/oivae_hip.h declares, the product refuses to run without a GPU, model
recognition reads the reference-style heads correctly, and the data-parallel
gradient mean over two gloo ranks equals the global-batch gradient."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  from oi_vae_amd import _lib
  header = open(os.path.join(ROOT, "include", "oivae_hip.h")).read()
  declared = sorted(set(re.findall(r"\b(oivae_[a-z0-9_]+)\s*\(", header)))
  assert declared == sorted(_lib.SYMBOLS)
  L = _lib.load()
  for sym in declared:
    assert hasattr(L, sym), sym
  assert b"gfx950" in L.oivae_version()


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "oi_vae_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".h", ".cpp")):
        text = open(os.path.join(dirpath, f)).read()
        assert "import oracle" not in text and "from oracle" not in text, f
        assert "prox_oracle" not in text or f.endswith(".hip"), f


def _bars_model(holder, K=8):
  import oi_vae_amd as oi
  inference_net = oi.NormalNet(
    mu_net=torch.nn.Sequential(torch.nn.Linear(64, K)),
    sigma_net=torch.nn.Sequential(torch.nn.Linear(64, K), oi.Lambda(torch.exp),
                                  oi.Lambda(lambda x: x * holder["c"] + 1e-3)))

  def make_gen():
    log_sigma = torch.log(1e-2 * torch.ones(8)).requires_grad_(True)
    return oi.NormalNet(mu_net=torch.nn.Linear(1, 8),
                        sigma_net=oi.Lambda(lambda x, ls: torch.exp(ls.expand(x.size(0), -1)) + 1e-3,
                                            extra_args=(log_sigma,)))
  gen = oi.BayesianGroupLassoGenerator([make_gen() for _ in range(8)], group_input_dim=1, dim_z=K)
  return inference_net, gen


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
  import oi_vae_amd as oi
  holder = {"c": 0.1}
  inf, gen = _bars_model(holder)
  with pytest.raises(oi.OivaeError):
    oi.OIVAE(inf, gen, oi.Normal(torch.zeros(1, 8), torch.ones(1, 8)), oi.NormalPriorTheta(1.0), 1.0, [])
  with pytest.raises(oi.OivaeError):
    gen.proximal_step(0.1)
  with pytest.raises(oi.OivaeError):
    oi.Engine(oi.ModelSpec(D=64, K=8, H=1, group_dims=[8] * 8, gen_type="linear"))


def test_head_probing_reads_the_lambda_constants():
  from oi_vae_amd import adopt
  holder = {"c": 0.1}
  inf, gen = _bars_model(holder)
  c, off = adopt.probe_encoder_head(inf)
  assert abs(c - 0.1) < 1e-15 and abs(off - 1e-3) < 1e-15
  holder["c"] = 1  # the scripts flip stddev_multiple after 1000 iterations (bars_data_oivae.py:258-259)
  c, off = adopt.probe_encoder_head(inf)
  assert abs(c - 1.0) < 1e-15
  ls, goff = adopt._probe_gen_sigma(gen.group_generators[0].sigma_net, 8)
  assert abs(goff - 1e-3) < 1e-15 and ls.numel() == 8
  assert adopt._gen_structure(gen.group_generators[0])[0] == "linear"


def test_unsupported_heads_are_refused():
  import oi_vae_amd as oi
  from oi_vae_amd import adopt
  bad = oi.NormalNet(mu_net=torch.nn.Sequential(torch.nn.Linear(4, 2)),
                     sigma_net=torch.nn.Sequential(torch.nn.Linear(4, 2), oi.Lambda(torch.sigmoid)))
  with pytest.raises(oi.OivaeError):
    adopt.probe_encoder_head(bad)
  g = oi.NormalNet(mu_net=torch.nn.Sequential(torch.nn.Linear(2, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3)),
                   sigma_net=None)
  with pytest.raises(oi.OivaeError):
    adopt._gen_structure(g)


def test_modelspec_validation():
  import oi_vae_amd as oi
  with pytest.raises(ValueError):
    oi.ModelSpec(D=10, K=2, H=1, group_dims=[4, 4], gen_type="linear")
  with pytest.raises(ValueError):
    oi.ModelSpec(D=8, K=2, H=1, group_dims=[4, 4], gen_type="mlp")
  s = oi.ModelSpec(D=8, K=2, H=3, group_dims=[5, 3], gen_type="mlp", M=7)
  assert s.seg_shapes()["gen_W2"] == (8, 7) and s.seg_shapes()["gen_W1"] == (2, 7, 3)


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from oracle import oivae_oracle as O
from oi_vae_amd import parallel
rank = int(sys.argv[1]); world = 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(1)
spec = dict(D=24, K=16, H=8, M=24, gen_type="mlp", group_dims=[5, 12, 7])
params = O.init_params(spec, seed=3)
rng = np.random.default_rng(0)
B = 32
X = rng.random((B, 24), dtype=np.float32); eps = rng.standard_normal((2, B, 16)).astype(np.float32)

def flat_grads(Xs, es):
  m = O.OracleModel(spec, params)
  t = O.forward_terms(m, torch.from_numpy(Xs), torch.from_numpy(es), 0.5, 1.0, 1.0)
  t["loss"].backward()
  g = m.export_grads()
  parts = [g[k].ravel() if not isinstance(g[k], list) else np.concatenate([a.ravel() for a in g[k]]) for k in sorted(g)]
  sc = np.array([float(t[k]) for k in ("z_kl", "loglik_term", "logprob_theta", "logprob_W", "loss", "elbo")], np.float32)
  return np.concatenate(parts), sc

lo, hi = parallel.shard_rows(B, rank, world)
g_local, sc_local = flat_grads(X[lo:hi], eps[:, lo:hi])
g = torch.from_numpy(g_local.copy()); sc = torch.from_numpy(sc_local.copy())
parallel.allreduce_mean_(g); parallel.reduce_scalars_(sc)
g_full, sc_full = flat_grads(X, eps)
err = np.max(np.abs(g.numpy() - g_full)) / np.max(np.abs(g_full))
serr = np.max(np.abs(sc.numpy() - sc_full) / np.maximum(np.abs(sc_full), 1))
assert err < 2e-6, err
assert serr < 2e-6, serr
assert parallel.is_active()
# the overlapped two-bucket form must give the same mean as the single all-reduce
class Holder: pass
h = Holder(); h.flat_grads = torch.from_numpy(g_local.copy()); h.seg_off = {"gen_W1": 1000}
sync = parallel.BucketedGradSync(h)
sync.decoder_grads_ready(); sync.encoder_grads_ready(); sync.wait()
assert torch.equal(h.flat_grads, g), float((h.flat_grads - g).abs().max())
sync.wait()  # idempotent once drained
assert torch.equal(h.flat_grads, g)
# ... and so must the per-bucket waits (bench.py: the decoder side's optimizers run between them)
h.flat_grads.copy_(torch.from_numpy(g_local))
sync.decoder_grads_ready(); sync.encoder_grads_ready()
sync.wait_decoder()
assert torch.equal(h.flat_grads[1000:], g[1000:])
sync.wait_encoder()
assert torch.equal(h.flat_grads, g)
dist.destroy_process_group()
print("rank", rank, "ok", err, serr)
'''


def test_data_parallel_mean_equals_global_batch_gradient_gloo(tmp_path):
  script = tmp_path / "dp_worker.py"
  script.write_text(_DP_WORKER % {"root": ROOT})
  port = str(29500 + (os.getpid() % 2000))
  procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True) for r in range(2)]
  outs = [p.communicate(timeout=240)[0] for p in procs]
  for p, o in zip(procs, outs):
    assert p.returncode == 0, o


def test_shard_rows_rejects_uneven_split():
  from oi_vae_amd import parallel
  assert parallel.shard_rows(64, 1, 2) == (32, 64)
  with pytest.raises(ValueError):
    parallel.shard_rows(65, 0, 2)


def test_first_layer_sparse_decoder_has_no_cpu_fallback():
  import oi_vae_amd as oi
  dec = oi.FirstLayerSparseDecoder([torch.nn.Linear(1, 4, bias=False) for _ in range(3)], [1, 1, 1], 5)
  z = torch.randn(2, 5)
  assert dec(z).shape == (2, 12)                       # forward is plain torch, as in the reference
  assert len(list(dec.parameters())) == 6
  assert dec.group_lasso_penalty().item() > 0
  with pytest.raises(oi.OivaeError):
    dec.proximal_step(0.1)                             # the proximal step is the HIP kernel or nothing


def test_bench_reports_pmc_traffic_only_for_the_profiled_decoder_sources(tmp_path, monkeypatch):
  """bench.py's `roofline.traffic` comes from committed rocprofv3 PMC passes; a profile taken from OTHER decoder sources
  must give null (a stale figure is worse than none).  The committed file matches this tree; a copy with another stamp,
  or one that lacks the kernel, does not count."""
  import json
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench
  t = bench.pmc_traffic()
  assert t is not None and t.get("bytes"), "profiles/pmc_per_kernel_latest.json is stale: re-run tools/pmc_collect.sh + pmc_summarize.py"
  assert t["decoder_source_digest"] == bench.decoder_source_digest()
  assert 17.2e9 < t["read_bytes"] < 23e9 and 8.6e9 < t["write_bytes"] < 10e9     # X once + re-reads of z; dz adds + slabs
  d = json.load(open(os.path.join(bench.ROOT, bench.PMC_FILE)))
  d["_decoder_source_digest"] = "0" * 16
  stale = tmp_path / "pmc.json"
  stale.write_text(json.dumps(d))
  monkeypatch.setattr(bench, "PMC_FILE", str(stale))
  t = bench.pmc_traffic()
  assert t["bytes"] is None and t["stale"] is True
  stale.write_text(json.dumps({"_decoder_source_digest": bench.decoder_source_digest()}))
  assert bench.pmc_traffic() is None


def test_bench_flop_accounting_follows_the_survey():
  """3 G (2KH + 2HM + 2M D_g) = 75.50 MFLOP per row for the decoder, 2 (4 D K) = 33.55 for the encoder;
  the issued 16-bit flops are 2.0 x the algorithmic ones (6 composed products of 9, three terms each)."""
  import bench
  assert abs(bench.decoder_flops_per_row() / 1e6 - 75.50) < 0.01
  assert abs(bench.encoder_flops_per_row() / 1e6 - 33.55) < 0.01
  rows = bench.GLOBAL_BATCH
  assert abs(bench.decoder_f16_flops(rows) / (bench.decoder_flops_per_row() * rows) - 2.0) < 1e-9
  assert abs((bench.decoder_flops_per_row() + bench.encoder_flops_per_row()) * rows / 1e12 - 7.147) < 0.001


def test_c_abi_header_is_plain_c(tmp_path):
  """include/oivae_hip.h is the boundary a C, C++ or FFI host binds: it must compile as C99 and as C++ on its own."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  src = tmp_path / "hdr.c"
  src.write_text('#include "oivae_hip.h"\nint main(void) { return (int)sizeof(oivae_scalars_t) - 32; }\n')
  inc = os.path.join(root, "include")
  subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)
  subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)], check=True)


def test_dz_layout_follows_the_padded_decoder_rule():
  """`oivae_dz_layout` (host logic, no GPU call): two-layer generators up to 64 wide with one output width take the
  matrix decoder (dz transposed, 1) -- always when every dimension is 32 or 64, otherwise when the padded launch pays:
  K H + H M + M D_g >= 1536 per group or S B G >= 524288 row-groups (csrc/common.h kPadMinWork / kPadMinRowGroups,
  measured by tools/narrow_dims.py); ragged groups, widths above 64, a D_g that is no multiple of 4 and single-layer
  generators keep [N, K] (0)."""
  import ctypes
  from oi_vae_amd import _lib
  L = _lib.load()
  offs = (ctypes.c_int32 * 2)(0, 1)   # the layout rule never reads the offsets; the pointer only has to be non-null

  def layout(K, H, M, dg, G, B, S=1, gen="mlp", uniform=True):
    sh = _lib.Shape(D=dg * G, K=K, G=G, H=H, M=M, gen_type=_lib.GEN_TYPES[gen], max_group_dim=dg,
                    uniform_group_dim=dg if uniform else 0, group_off=ctypes.cast(offs, ctypes.c_void_p).value)
    return L.oivae_dz_layout(ctypes.byref(sh), B, S)

  assert layout(64, 64, 64, 64, 4, 64) == 1
  assert layout(32, 64, 32, 32, 3, 1) == 1                  # any mix of 32 and 64, whatever the size
  assert layout(40, 24, 56, 44, 5, 64) == 1                 # narrow, enough work per group (4768)
  assert layout(16, 16, 16, 16, 8, 256) == 0                # narrow, 768 < 1536, 2048 row-groups
  assert layout(16, 16, 16, 16, 256, 2048) == 1             # the same shape at 524288 row-groups
  assert layout(16, 16, 16, 16, 256, 1024, S=2) == 1        # ... counted in decoder rows S B
  assert layout(16, 16, 16, 16, 256, 2047) == 0
  assert layout(64, 64, 64, 42, 4, 4096) == 0               # D_g no multiple of 4: X is read 16 bytes per lane
  assert layout(64, 64, 96, 64, 4, 4096) == 0               # wider than 64
  assert layout(64, 64, 64, 64, 4, 4096, uniform=False) == 0   # ragged groups
  assert layout(10, 1, 0, 64, 4, 4096, gen="linear") == 0   # single-layer generators
