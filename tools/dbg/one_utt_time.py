"""One-utterance latency: p_sample steps and the T=4 sampling loop at B=1, L=1000, teams on / off (denoiser_team16.h)."""
import os, sys, tempfile, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats)).cuda().eval()
with torch.no_grad():
    gd.denoise_fn.output_projection.conv.weight.normal_(0, 0.02)
for B, L in ((1, 1000), (2, 1000), (1, 2000), (4, 500), (4, 250)):
    cond = torch.randn(B, 256, L, device="cuda"); x = torch.randn(B, 80, L, device="cuda"); o = torch.empty_like(x)
    t = torch.full((B,), 2, device="cuda")
    for team in ("", "0"):
        if team:
            os.environ["MG_PERSIST_TEAM"] = team
        else:
            os.environ.pop("MG_PERSIST_TEAM", None)
        for _ in range(5): gd._p_sample_bml(x, t, cond, None, None, True, out=o)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): gd._p_sample_bml(x, t, cond, None, None, True, out=o)
        e1.record(); torch.cuda.synchronize()
        step = e0.elapsed_time(e1) / 50
        gd.cond, gd.spk_emb = cond, None
        gd.sampling(keep_trace=False); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): gd.sampling(keep_trace=False)
        torch.cuda.synchronize(); samp = (time.perf_counter() - t0) / 10 * 1e3
        print("B=%d L=%d teams=%s: p_sample %.3f ms (%.1f TFLOP/s), sampling T=4 %.3f ms" % (B, L, team or "auto", step, 23.805952e6 * B * L / step / 1e9, samp))
