"""Bars images for the bars experiment (counterpart of /root/reference/lib/bars_data.py:4-23).

The images are tiny and made on the host.  What has to match the reference is the
stream of random draws, not the arithmetic around them: every function below takes
its draws from torch's CPU default generator in the reference's order (one
`bernoulli` over the row probabilities, then one over the column probabilities, per
image; one `randperm` per one-bar image), so a run seeded with `torch.manual_seed`
sees the same data on either code base (tests/test_data_formats.py compares with
images the reference drew).  The images themselves are formed by broadcasting.
"""
import torch


def _bar_draws(row_probs, col_probs):
  """(row indicator [m], column indicator [n]) of one image: rows are drawn first."""
  rows = torch.bernoulli(row_probs)
  cols = torch.bernoulli(col_probs)
  return rows, cols


def sample_bars_image(row_probs, col_probs):
  """[m, n] image: pixel (i, j) = (row i lit + column j lit) / 2."""
  rows, cols = _bar_draws(row_probs, col_probs)
  return 0.5 * (rows.unsqueeze(1) + cols.unsqueeze(0))


def sample_one_bar_image(image_size):
  """[image_size, image_size] image with one uniformly chosen row at 0.5."""
  lit = torch.randperm(image_size)[0]
  rows = torch.zeros(image_size)
  rows[lit] = 0.5
  return rows.unsqueeze(1).expand(image_size, image_size).clone()


def sample_many_bars_images(num_images, image_size, row_mul, col_mul):
  """[num_images, image_size^2]: each row and column lit with probability mul / image_size.
  (The reference's own version of this function raises on torch >= 1.x: it calls .view(-1)
  on a non-contiguous sum.)"""
  # the probabilities in the reference's fp32 association, (1 / n) * mul: rounding mul / n once instead differs by an
  # ulp for some (n, mul) -- e.g. n = 7, mul = 3 -- and a Bernoulli draw compares against exactly this number
  unit = torch.ones(image_size) / image_size
  p_row, p_col = unit * row_mul, unit * col_mul
  draws = [_bar_draws(p_row, p_col) for _ in range(num_images)]       # the reference's draw order
  rows = torch.stack([r for r, _ in draws])                           # [num_images, image_size]
  cols = torch.stack([c for _, c in draws])
  return (0.5 * (rows.unsqueeze(2) + cols.unsqueeze(1))).reshape(num_images, image_size * image_size)
