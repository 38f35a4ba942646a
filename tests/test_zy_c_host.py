"""The C ABI from a host that is not Python: examples/c_host_prox.cpp (built in-tree by oi_vae_amd/csrc/Makefile) links
liboivae_hip.so, allocates with the HIP runtime, calls oivae_init / oivae_prox_f32 / oivae_check_errors with plain
pointers and a HIP stream, and checks the result against the reference's formula on the host."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_calls_the_proximal_step_through_the_c_abi():
  exe = os.path.join(ROOT, "oi_vae_amd", "c_host_prox")
  assert os.path.exists(exe), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
  r = subprocess.run([exe], capture_output=True, text=True, timeout=100)
  assert r.returncode == 0, r.stdout + r.stderr
  assert r.stdout.strip().startswith("ok:"), r.stdout
