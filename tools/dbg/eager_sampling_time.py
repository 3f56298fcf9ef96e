"""Per-call wall time of eager GaussianDiffusion.sampling at small shapes (host-side overhead hunting)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from helpers import hot_path_configs, write_stats  # noqa: E402
import mixgan_tts_amd as mg  # noqa: E402

dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as d:
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
    for B, L in ((1, 1000), (4, 256), (4, 250), (2, 1000)):
        gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats)).to(dev).eval()
        gd.cond = torch.randn(B, 256, L, device=dev)
        gd.spk_emb = None
        times = []
        for i in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gd.sampling(keep_trace=False)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        print("B=%d L=%d:" % (B, L), " ".join("%.2f" % t for t in times), flush=True)
        del gd
        torch.cuda.empty_cache()
