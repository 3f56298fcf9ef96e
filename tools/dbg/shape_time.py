"""p_sample and T=4 sampling() times at given (B, L) shapes:  python tools/dbg/shape_time.py 8x1000 4x1000 ..."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from helpers import hot_path_configs, write_stats  # noqa: E402
import mixgan_tts_amd as mg  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8, 1000)]
with tempfile.TemporaryDirectory() as d:
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
    for B, L in shapes:
        gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats)).to(dev).eval()
        gd.cond = torch.randn(B, 256, L, device=dev)
        gd.spk_emb = None
        x = torch.randn(B, 80, L, device=dev)
        t = torch.full((B,), 2, device=dev, dtype=torch.long)

        def timeit(fn, n):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        ps = timeit(lambda: gd._p_sample_bml(x, t, gd.cond, None, None), 40)
        sm = timeit(lambda: gd.sampling(keep_trace=False), 20)
        print("B=%d L=%d: p_sample %.3f ms (%.1f TFLOP/s), sampling T=4 %.3f ms" % (B, L, ps, 23805952.0 * B * L / ps / 1e9, sm), flush=True)
        del gd
        torch.cuda.empty_cache()
