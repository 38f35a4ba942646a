#!/usr/bin/env python3
"""Host-side profile (cProfile) of the bars training loop of scripts/bars_data_oivae.py: where the time of a 0.15 ms
step goes when the GPU part is a 0.034 ms launch.  usage: python tools/profile_bars_host.py [iterations]"""
import cProfile
import os
import pstats
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "5000"
sys.argv = ["bars", "--iterations", n, "--out", "/tmp/out_bars_prof", "--log-every", "100000"]
cProfile.run('runpy.run_path(%r, run_name="__main__")' % os.path.join(root, "scripts", "bars_data_oivae.py"), "/tmp/bars_prof.out")
st = pstats.Stats("/tmp/bars_prof.out")
st.sort_stats("cumulative").print_stats("oi_vae_amd|scripts|bars", 40)
st.sort_stats("tottime").print_stats(25)
