"""JCUDiscriminator forward + backward time at the training shard's shape (2B = 16 items of 1000 frames)."""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
D = mg.JCUDiscriminator(pre, mc, tr).cuda()
B, L = 16, 1000
x1 = torch.randn(B, L, 80, device="cuda", requires_grad=True); x2 = torch.randn(B, L, 80, device="cuda")
t = torch.randint(0, 4, (B,), device="cuda")
def run():
    c, u = D(x1, x2, None, t)
    (c[-1].sum() + u[-1].sum()).backward()
for _ in range(3): run()
print("scratch in use:", len(mg.ops._SPLIT_SCRATCH))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("JCU fwd+bwd %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
