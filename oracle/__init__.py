"""CPU oracle for the oi-VAE training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and only as the checker.  The product
(``oi_vae_amd``) never imports this package and fails loudly when its HIP
library is missing.
"""
