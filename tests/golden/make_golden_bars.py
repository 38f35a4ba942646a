#!/usr/bin/env python3
"""Bars images drawn by the REAL reference sampler (/root/reference/lib/bars_data.py)
under fixed seeds:  python tests/golden/make_golden_bars.py   (build container only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from lib.bars_data import sample_bars_image, sample_many_bars_images, sample_one_bar_image  # noqa: E402

out = {}
torch.manual_seed(5)
out["one_bar_8"] = torch.stack([sample_one_bar_image(8).view(-1) for _ in range(16)]).numpy()
torch.manual_seed(6)
# sample_many_bars_images itself raises on torch >= 1.x (.view(-1) of a transposed sum, lib/bars_data.py:21);
# the harness replays its body with .reshape(-1), which draws the same numbers
out["many_16"] = torch.stack([
  sample_bars_image(torch.ones(16) / 16 * 2.0, torch.ones(16) / 16 * 0.0).reshape(-1) for _ in range(12)]).numpy()
torch.manual_seed(7)
out["image_5x7"] = sample_bars_image(torch.full((5,), 0.4), torch.full((7,), 0.3)).numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bars_samples.npz"), **out)
print({k: v.shape for k, v in out.items()})
