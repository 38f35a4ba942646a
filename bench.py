#!/usr/bin/env python3
"""bench.py -- oi-VAE train samples/s (fwd + bwd + optimizers + prox) on MI355X.

Workload (BASELINE.json configs[3], "synthetic scale-up"): G=1024 groups x 64
dims, K=64, two-layer generators Linear(64,64)->Tanh->Linear(64,64), global
batch 65536, S=1, synthetic X ~ U[0,1) resident in HBM, random-init weights.
One step = encoder fwd + reparam + KL, grouped decoder fwd+bwd, encoder bwd,
weight prior + penalty, [gradient all-reduce over RCCL when N>1], fused Adam,
fused SGD + proximal step.  N>1: one process per GPU (torchrun), the global
batch is sharded over ranks ("strong" scaling, as BASELINE.json states it).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

G, DG, K, H, M = 1024, 64, 64, 64, 64
GLOBAL_BATCH = 65536
LR, WS_LR, LAM = 1e-3, 1e-4, 1.0
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
F16_MATRIX_PEAK_TFLOPS = 16 * 157.3  # dense fp16/bf16 MFMA = 16 x the fp32 matrix rate, same table (2516.8)
# what the chip sustains on v_mfma_f32_16x16x32_f16 alone with random operands at its power limit
# (tools/probe/mfma_shape_power.hip: 2006 TFLOP/s; the 32x32x16 shape reaches 1772-1852, all-zero operands 2445)
F16_MATRIX_SUSTAINED_TFLOPS = 2006.0
PMC_FILE = os.path.join("profiles", "pmc_per_kernel_latest.json")


DEC_SOURCES = ("oi_vae_amd/csrc/dec_x3_64s.h", "oi_vae_amd/csrc/mfma_decoder.hip", "oi_vae_amd/csrc/dec_shared.h",
               "oi_vae_amd/csrc/x3_split.h")


def decoder_source_digest():
  import hashlib
  h = hashlib.sha256()
  for f in DEC_SOURCES:
    with open(os.path.join(ROOT, f), "rb") as fh:
      h.update(fh.read())
  return h.hexdigest()[:16]


def pmc_traffic():
  """HBM bytes per decoder launch from the committed rocprofv3 PMC passes (profiles/README.md names the commands).
  Counters cannot be read from inside this process, so this is a committed measurement -- valid for THIS tree only if
  it was taken from the same decoder sources: tools/pmc_summarize.py stamps the file with a digest of them, and a
  stamp that does not match the tree's gives null (a stale figure is worse than none)."""
  try:
    d = json.load(open(os.path.join(ROOT, PMC_FILE)))
    stamp = d.get("_decoder_source_digest")
    if stamp != decoder_source_digest():
      return {"bytes": None, "stale": True, "source": PMC_FILE,
              "note": "profile taken from other decoder sources (digest %s, tree %s)" % (stamp, decoder_source_digest())}
    name, k = next((n, v) for n, v in sorted(d.items(), reverse=True) if "dec_x3_64s" in n)
    return {"bytes": k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"], "read_bytes": k["hbm_read_bytes_corrected"],
            "write_bytes": k["hbm_write_bytes"], "kernel": name, "source": PMC_FILE, "decoder_source_digest": stamp}
  except Exception:
    return None


def decoder_flops_per_row():
  # ALGORITHMIC flops: fwd + dgrad + wgrad of the three per-group products
  return 3 * G * (2 * K * H + 2 * H * M + 2 * M * DG)


def decoder_executed_flops(rows):
  """fp32-equivalent flops the decoder kernels issue: Ws[g] and the generator's first Linear compose
  (no nonlinearity between them), so the row loop has 6 products of 2*64*64 flops per row
  and group instead of 9, plus three 64^3 products per group outside it."""
  return rows * G * 6 * 2 * 64 * 64 + G * 3 * 2 * 64 ** 3


def decoder_f16_flops(rows):
  """16-bit matrix flops dec_x3_64s issues in the row loop: each of the six fp32 products (64 x 32 x 64 per 32-row tile:
  eight 16 x 16 blocks, two reduction steps of 32) as three fp16 products: 288 v_mfma_f32_16x16x32_f16 of 2*16*16*32
  flops per 32-row tile and group (dec_x3_64g issued the same flops as 144 32x32x16 instructions)."""
  return (rows + 31) // 32 * G * 288 * (2 * 16 * 16 * 32)


def encoder_flops_per_row():
  return 2 * (4 * (G * DG) * K)


def init_engine(device, batch_rows, seed):
  from oi_vae_amd import Engine, ModelSpec
  spec = ModelSpec(D=G * DG, K=K, H=H, M=M, group_dims=[DG] * G, gen_type="mlp")
  eng = Engine(spec, device=device)
  gen = torch.Generator(device=device).manual_seed(1234)  # same parameters on every rank

  def uniform_(t, fan_in):
    bound = 1.0 / (fan_in ** 0.5)
    t.copy_((torch.rand(t.shape, generator=gen, device=device) * 2 - 1) * bound)

  D = spec.D
  uniform_(eng.p["enc_Wmu"], D); uniform_(eng.p["enc_bmu"], D)
  uniform_(eng.p["enc_Wsig"], D); uniform_(eng.p["enc_bsig"], D)
  uniform_(eng.p["gen_W1"], H); uniform_(eng.p["gen_b1"], H)
  uniform_(eng.p["gen_W2"], M); uniform_(eng.p["gen_b2"], M)
  eng.p["log_sigma"].fill_(float(torch.log(torch.tensor(1e-2))))
  eng.p["Ws"].copy_(torch.randn(eng.p["Ws"].shape, generator=gen, device=device))
  xg = torch.Generator(device=device).manual_seed(seed)
  X = torch.rand(batch_rows, D, generator=xg, device=device)
  return eng, X


def prox_roofline(device):
  """Secondary measurement (BASELINE.json configs[4], wide-prox stress): the fused
  SGD + proximal kernel on G=8192 x K=256 x H=8, threshold at the median row norm
  (about half of the rows die).  HBM-bound: 12 B/element (Ws read+write, dWs read)
  + 5 B/row (mask byte + norm)."""
  import ctypes
  from oi_vae_amd import _lib
  L = _lib.init_device(device.index)
  Gp, Kp, Hp = 8192, 256, 8
  gen = torch.Generator(device=device).manual_seed(3)
  W0 = torch.randn(Gp, Kp, Hp, generator=gen, device=device)
  dW = torch.randn(Gp, Kp, Hp, generator=gen, device=device)
  t = float(torch.sqrt((W0 ** 2).sum(-1)).median())
  mask = torch.zeros(Gp, Kp, dtype=torch.uint8, device=device)
  norms = torch.zeros(Gp, Kp, dtype=torch.float32, device=device)
  st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
  reps = 20
  Ws = [W0.clone() for _ in range(reps + 2)]  # fresh input per launch so every launch does the same work
  evs = []
  for i, W in enumerate(Ws):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(L.oivae_sgd_prox_f32(W.data_ptr(), dW.data_ptr(), Gp * Kp, Hp, 1e-4, t, 1, mask.data_ptr(),
                                    norms.data_ptr(), st), "oivae_sgd_prox_f32")
    b.record()
    evs.append((a, b))
  torch.cuda.synchronize()
  ms = sorted(a.elapsed_time(b) for a, b in evs[2:])
  avg = sum(ms) / len(ms)
  nbytes = 12 * Gp * Kp * Hp + 5 * Gp * Kp
  return {"workload": "wide-prox: SGD+proximal step on Ws [8192,256,8], t = median row norm",
          "bound": "hbm", "achieved": nbytes / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
          "frac": nbytes / (avg * 1e-3) / 1e9 / 8000.0, "avg_launch_ms": avg, "bytes_per_launch": nbytes,
          "zeroed_fraction": float(mask.float().mean())}


def api_leg(device, batches, rows_list, steps=10, warmup=3):
  """The same workload through the drop-in API: `OIVAE.step` (oi_vae_amd/oivae.py, the reference's signature) on a model
  built from the reference-shaped objects (NormalNet, 1024 group generators, BayesianGroupLassoGenerator), engine-native
  Adam and SGD, X resident, noise drawn on the device.  ms per step at each row count."""
  from oi_vae_amd.experiments import Experiment
  torch.manual_seed(4321)
  exp = Experiment(dim_x=G * DG, dim_z=K, group_dims=[DG] * G, group_input_dim=H, gen_type="mlp", hidden=M,
                   lr=LR, Ws_lr=WS_LR, lam=LAM, device=device, device_eps=True)
  exp.stddev_multiple = 1.0
  exp.iteration = 2000
  out = {}
  for rows in rows_list:
    xs = [b[:rows] for b in batches]
    for i in range(warmup):
      exp.step(xs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
      exp.step(xs[i % 2])
    torch.cuda.synchronize()
    out[str(rows)] = (time.perf_counter() - t0) / steps * 1e3
  del exp
  return out


def cpu_baseline(batch=1024, steps=8):
  """The oracle (CPU restatement with the reference's per-group loop) on the host
  cores, same shapes at a reduced batch; reported, not a target."""
  import numpy as np
  from oracle import oivae_oracle as O
  torch.set_num_threads(max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))))
  spec = dict(D=G * DG, K=K, H=H, M=M, gen_type="mlp", group_dims=[DG] * G)
  params = O.init_params(spec, seed=0)
  tr = O.OracleTrainer(spec, params, lr=LR, Ws_lr=WS_LR, lam=LAM)
  rng = np.random.default_rng(0)
  X = rng.random((batch, spec["D"]), dtype=np.float32)
  eps = rng.standard_normal((1, batch, K)).astype(np.float32)
  t0 = time.perf_counter()
  for _ in range(steps):
    tr.step(X, eps, c=1.0, prox_step_size=WS_LR * LAM, want_grads=False)
  dt = time.perf_counter() - t0
  return {"value": batch * steps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
          "sample": "%d oracle steps (per-group loop, torch CPU autograd + Adam/SGD + prox) of the same shapes "
                    "at batch %d instead of 65536 (%.1f s)" % (steps, batch, dt)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-extras", action="store_true", help="skip the comparator kernels, the range-change and the sustained run")
  ap.add_argument("--sustain-seconds", type=float, default=10.0)
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    # `python bench.py --gpus N` by itself: start the N ranks as fresh child processes (one per GPU, the same command
    # line under torch.distributed.run) and pass their output through.  Nothing in THIS process has touched the GPU --
    # it only waits -- and nothing is exec'ed over it.
    import socket
    import subprocess
    with socket.socket() as sk:
      sk.bind(("127.0.0.1", 0))
      port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)
  if world != args.gpus:
    raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
  # OIVAE_BENCH_REHEARSE=1: all ranks on cuda:0 over gloo -- walks the N > 1 code path (sharding, the two gradient
  # buckets, barriers, max over ranks, rank 0's line) on a one-GPU box; its numbers mean nothing and the line says so
  rehearse = os.environ.get("OIVAE_BENCH_REHEARSE") == "1"
  if rehearse:
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=False, file=sys.stderr)   # a hang names its line
  dev_index = 0 if rehearse else local_rank
  torch.cuda.set_device(dev_index)
  device = torch.device("cuda", dev_index)
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearse:
      dist.init_process_group("gloo")
    else:
      dist.init_process_group("nccl", device_id=device)
  from oi_vae_amd import parallel

  rows = GLOBAL_BATCH // world
  eng, X = init_engine(device, rows, seed=100 + rank)
  # two resident minibatches, alternated: a step never sees the X of the step before (the fp16 kernels keep
  # max|X| of the previous pass in the workspace; same distribution, so the power of two does not move)
  X2 = torch.rand(rows, G * DG, generator=torch.Generator(device=device).manual_seed(900 + rank), device=device)
  batches = [X, X2]
  egen = torch.Generator(device=device).manual_seed(7 + rank)
  t_prox = WS_LR * LAM
  events = []   # (name, start, end) of the decoder entry point in the timed steps
  diag = []     # ... of every entry point in the diagnostic steps
  sync = parallel.BucketedGradSync(eng) if world > 1 else None
  counter = [0]
  comm_marks = []

  def step(timed, xs=None, diagnose=False):
    """timed: the decoder entry point's events go to `events`; diagnose (the untimed steps behind the timed region):
    every entry point's events go to `diag`, and the waits for the gradient buckets are bracketed too."""
    xs = batches if xs is None else xs
    Xs = xs[counter[0] % len(xs)]
    counter[0] += 1
    eps = torch.randn(1, rows, K, generator=egen, device=device)
    if diagnose:
      eng.trace, eng.trace_names = diag, None
    elif timed:
      eng.trace, eng.trace_names = events, {"decoder"}
    eng.forward_backward(Xs, eps, 1.0, lam=LAM, grad_sync=sync)
    eng.trace, eng.trace_names = None, None
    if sync is not None:
      # every bucket is averaged over ranks before ITS optimizers run: the decoder-side bucket (reduced under the
      # encoder backward GEMM) first, so that its Adam, SGD and prox run under the encoder bucket's all-reduce
      # (diagnostic steps: events on the compute stream around each wait = the time that stream stood still for a
      # bucket, i.e. the communication the step did NOT hide)
      marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if diagnose else None
      if diagnose: marks[0].record()
      sync.wait_decoder()
      if diagnose: marks[1].record()
      eng.adam_step(LR, part="dec")
      eng.sgd_prox_step(WS_LR, t_prox)
      if diagnose: marks[2].record()
      sync.wait_encoder()
      if diagnose:
        marks[3].record()
        comm_marks.append(marks)
      eng.adam_step(LR, part="enc")
    else:
      eng.adam_step(LR)
      eng.sgd_prox_step(WS_LR, t_prox)

  def barrier():
    if world > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  def timed_steps(n, timed=True, xs=None):
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
      step(timed, xs)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
      tmax = torch.tensor([dt], dtype=torch.float64, device=device)
      torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
      dt = float(tmax.item())
    return dt

  for _ in range(args.warmup):
    step(False)
  # Inside the timed region only the decoder entry point carries HIP events (the roofline needs its launch time): a
  # pair of events per entry point costs ~0.05 ms per step (tools/trace_cost.py: 1.7 % of the step at an eighth of
  # the batch).  The other entry points' times come from three more, untimed, steps behind it.
  dt = timed_steps(args.steps)
  barrier()
  for _ in range(3):
    step(False, diagnose=True)
  barrier()

  extras = {}
  if world == 1 and not args.no_extras:
    # same-run comparators on full-width arithmetic (test hooks of the library)
    for tag, env in (("bf16x3_exact_operands", {"OIVAE_ENC": "bf16", "OIVAE_DEC": "p"}),
                     ("fp32_matrix_instructions", {"OIVAE_ENC": "fp32", "OIVAE_DEC": "v4"})):
      from oi_vae_amd import _lib
      _lib.set_test_hooks(**env)
      eng._ws_key = None
      for _ in range(2):
        step(False)
      extras["ms_per_step_" + tag] = timed_steps(4, timed=False) / 4 * 1e3
      _lib.set_test_hooks(**{k: None for k in env})
    eng._ws_key = None
    for _ in range(2):
      step(False)
    # worst case for the range guards: max|X| crosses a power of two every step, so every kernel that scales X
    # (encoder forward, decoder, encoder backward) runs twice
    X3 = X2 * 2.5
    for _ in range(2):
      step(False, [X, X3])
    extras["ms_per_step_max_x_changes_every_step"] = timed_steps(6, timed=False, xs=[X, X3]) / 6 * 1e3
    del X3
    for _ in range(2):
      step(False)
    # the drop-in API path (OIVAE.step) beside the engine path the timed region drives, at the full batch and at an
    # eighth of it (the per-GPU shard of an 8-GPU run, where host work per step would show first)
    small = GLOBAL_BATCH // 8
    xs_small = [X[:small], X2[:small]]
    rows_saved = rows

    def engine_ms(xs_, n_rows, n):
      nonlocal rows
      rows = n_rows
      try:
        for _ in range(3):
          step(False, xs_)
        return timed_steps(n, timed=False, xs=xs_) / n * 1e3
      finally:
        rows = rows_saved

    extras["engine_ms_per_step"] = {str(GLOBAL_BATCH): dt / args.steps * 1e3, str(small): engine_ms(xs_small, small, 20)}
    eng._ws_key = None
    extras["api_ms_per_step"] = api_leg(device, batches, [GLOBAL_BATCH, small])
    extras["api_over_engine"] = {k: extras["api_ms_per_step"][k] / extras["engine_ms_per_step"][k] for k in extras["api_ms_per_step"]}
    for _ in range(2):
      step(False)
    if args.sustain_seconds > 0:
      n = max(args.steps, int(args.sustain_seconds / (dt / args.steps)) + 1)
      ds = timed_steps(n, timed=False)
      extras["sustained"] = {"steps": n, "seconds": ds, "ms_per_step": ds / n * 1e3, "samples_per_s": GLOBAL_BATCH * n / ds}

  if rank == 0:
    ms = dt / args.steps * 1e3
    value = GLOBAL_BATCH * args.steps / dt
    dec_ms = [a.elapsed_time(b) for (name, a, b) in events if name == "decoder"]
    dec_avg = sum(dec_ms) / max(1, len(dec_ms))
    sec = dec_avg * 1e-3
    alg = decoder_flops_per_row() * rows
    exe = decoder_executed_flops(rows)
    f16 = decoder_f16_flops(rows)
    tf = lambda fl: fl / sec / 1e12 if sec > 0 else 0.0
    per_kernel = {}
    for name, a, b in diag:
      per_kernel.setdefault(name, []).append(a.elapsed_time(b))
    traffic = pmc_traffic() if world == 1 else None
    out = {
      "metric": "oi-VAE train samples/sec (fwd+bwd+prox)", "value": value, "unit": "samples/s",
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
      "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
      "data": "synthetic",
      "config": {"workload": "synthetic scale-up: G=1024 groups x 64 dims, K=64, H=M=64 two-layer generators, "
                             "global batch 65536, S=1", "global_batch": GLOBAL_BATCH,
                 "parallelism": "dp%d" % world, "rows_per_gpu": rows},
      # the dominant kernel against the pipe it runs on: v_mfma_f32_16x16x32_f16
      "roofline": {"bound": "mfma", "kernel": "dec_x3_64s (grouped decoder fwd+bwd, fp32 products as 3 fp16 MFMAs)",
                   "achieved": tf(alg), "peak": F16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": tf(alg) / F16_MATRIX_PEAK_TFLOPS,
                   "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
                   "avg_launch_ms": dec_avg,
                   "algorithmic_flops_per_launch": alg,
                   "issued_f16_flops_per_launch": f16, "issued_tflops": tf(f16),
                   "issued_frac": tf(f16) / F16_MATRIX_PEAK_TFLOPS,
                   "issued_frac_of_sustained": tf(f16) / F16_MATRIX_SUSTAINED_TFLOPS, "sustained_peak": F16_MATRIX_SUSTAINED_TFLOPS,
                   "fp32_equivalent": {"executed_flops_per_launch": exe, "executed_tflops": tf(exe),
                                       "algorithmic_over_fp32_matrix_peak": tf(alg) / FP32_MATRIX_PEAK_TFLOPS,
                                       "fp32_matrix_peak": FP32_MATRIX_PEAK_TFLOPS},
                   "note": "achieved/frac: the reference algorithm's flops (SURVEY 8d: 3 G (2KH + 2HM + 2M D_g) per row = "
                           "75.5 MFLOP, x rows per launch) over the launch time, against the dense fp16 matrix peak -- the "
                           "pipe the kernel runs on.  issued_*: the 16-bit matrix flops the kernel ISSUES (288 "
                           "v_mfma_f32_16x16x32_f16 per 32-row tile and group): each fp32 product costs three fp16 "
                           "products, and W^(g) z composed with the generator's first Linear makes 6 products of the "
                           "9, so issued = 2.0 x algorithmic.  issued_frac_of_sustained: against what the power-limited "
                           "chip reaches on that instruction alone with random operands.  fp32_equivalent: the same "
                           "work priced on the fp32 matrix pipe the kernel does not use.  avg_launch_ms is the decoder "
                           "ENTRY POINT timed with HIP events on its stream: dec_x3_64s plus ~0.06 ms of helpers (max|z|, "
                           "the image of z, the range guard and its gated second launch); profiles/ holds the "
                           "rocprofv3 kernel trace of the same command.  The kernel is not matrix-bound: the chain "
                           "wave's dependent sequence and the dz float atomics (8.6 GB per launch) bound it, DESIGN 3.0"},
      "step_model_tflops": (decoder_flops_per_row() + encoder_flops_per_row()) * GLOBAL_BATCH / (ms * 1e-3) / 1e12 / world,
      "entry_ms": {k: sum(v) / len(v) for k, v in per_kernel.items()},
      "entry_ms_note": "three untimed steps behind the timed region, every entry point bracketed by HIP events; inside "
                       "the timed region only the decoder's are recorded (roofline.avg_launch_ms)",
      "extra": extras,
      "arithmetic": "fp32 in, fp32 out, fp32 accumulation everywhere.  Encoder GEMMs and decoder products: two fp16 "
                    "terms per operand after an exact power-of-two scaling by the operand's maximum (or a bound of "
                    "it), three v_mfma_f32_16x16x32_f16 products per fp32 product (x3_split.h); every kernel that "
                    "scales X checks the maximum it assumed against what it read and repeats itself otherwise.  "
                    "Error against float64 no larger than the fp32 matrix instructions' (tests/test_gpu_x3.py).  "
                    "extra.ms_per_step_bf16x3_exact_operands: OIVAE_ENC=bf16 OIVAE_DEC=p (operands carried exactly as "
                    "three bf16 terms, six products); extra.ms_per_step_fp32_matrix_instructions: OIVAE_ENC=fp32 "
                    "OIVAE_DEC=v4",
    }
    if sync is not None and comm_marks:
      n = len(comm_marks)
      out["comm"] = {"backend": torch.distributed.get_backend(),
                     "bucket_bytes": {"decoder_side": int(sync.dec.numel()) * 4, "encoder": int(sync.enc.numel()) * 4},
                     "exposed_wait_ms": {"decoder_side": sum(m[0].elapsed_time(m[1]) for m in comm_marks) / n,
                                         "encoder": sum(m[2].elapsed_time(m[3]) for m in comm_marks) / n},
                     "note": "rank 0's compute stream, HIP events around the waits for the two gradient buckets: the "
                             "decoder-side bucket is reduced under the encoder backward GEMM, the encoder bucket under "
                             "the decoder side's Adam, SGD and proximal step"}
    if rehearse:
      out["rehearsal"] = "all %d ranks on one GPU over gloo: code-path check only, the numbers are meaningless" % world
    if world == 1:
      out["prox_roofline"] = prox_roofline(device)
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
  if world > 1:
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
