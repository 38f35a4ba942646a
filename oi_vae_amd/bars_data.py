"""Bars images, as the reference samples them (/root/reference/lib/bars_data.py:4-23).

Host-side data generation on the CPU default generator: the same torch calls in
the same order, so a seeded run draws the same images as the reference."""
import torch


def sample_bars_image(row_probs, col_probs):
  """Bernoulli row bars + Bernoulli column bars, each worth 0.5 (lib/bars_data.py:4-8)."""
  m = row_probs.numel()
  n = col_probs.numel()
  return (torch.bernoulli(row_probs).repeat(n, 1).t() +
          torch.bernoulli(col_probs).repeat(m, 1)) / 2


def sample_one_bar_image(image_size):
  """One random row set to 0.5 (lib/bars_data.py:10-14)."""
  i = torch.randperm(image_size)[0]
  A = torch.zeros(image_size, image_size)
  A[i, :] = 0.5
  return A


def sample_many_bars_images(num_images, image_size, row_mul, col_mul):
  """[num_images, image_size^2] flattened bars images (lib/bars_data.py:16-23)."""
  return torch.stack([
    sample_bars_image(
      torch.ones(image_size) / image_size * row_mul,
      torch.ones(image_size) / image_size * col_mul
    ).reshape(-1)  # the reference's .view(-1) raises on torch >= 1.x for this non-contiguous sum; same values
    for _ in range(num_images)
  ])
