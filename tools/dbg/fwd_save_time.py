import sys, tempfile, torch, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))); sys.path.insert(0, sys.path[0] + "/tests")
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
_, pre, mc, _ = hot_path_configs(stats_dir=stats)
den = mg.Denoiser(pre, mc).cuda()
with torch.no_grad():
    den.output_projection.conv.weight.normal_(0, 0.05)
def timeit(B, save, n=20):
    x = torch.randn(B, 80, 1000, device="cuda"); c = torch.randn(B, 256, 1000, device="cuda"); t = torch.randint(0, 4, (B,), device="cuda")
    f = (lambda: den.run(x, t, c, None, save=True)) if save else (lambda: den.run(x, t, c, None))
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B, save in [(8, False), (8, True), (16, False), (16, True)]:
    print("B=%d save=%s: %.3f ms" % (B, save, timeit(B, save)))
x = torch.randn(8, 80, 1000, device="cuda"); x2 = torch.randn(8, 80, 1000, device="cuda")
c = torch.randn(8, 256, 1000, device="cuda"); t = torch.randint(0, 4, (8,), device="cuda")
def pair():
    r = den.run_pair(x, t, x2, t, c, None); r[2]._mg_busy = False
for _ in range(3): pair()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): pair()
e1.record(); torch.cuda.synchronize()
print("pair 8+8 (second half saved): %.3f ms" % (e0.elapsed_time(e1) / 20))
