import os, sys, tempfile, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5]*80, [2.0]*80)
B, L = int(sys.argv[1]), 1000
args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
G = mg.GaussianDiffusion(args, pre, mc, tr); D = mg.JCUDiscriminator(pre, mc, tr)
gen = torch.Generator().manual_seed(2)
with torch.no_grad():
    for p in G.parameters():
        fan = p[0].numel() if p.dim() > 1 else 1
        p.copy_(torch.randn(p.shape, generator=gen) * (fan ** -0.5 if p.dim() > 1 else 0.1))
G, D = G.to(dev), D.to(dev)
trainer = mg.HotPathTrainer(G, D, tr, mc)
mel = torch.rand(B, L, 80, device=dev) * 13.5 - 11.5
cond = torch.randn(B, L, 256, device=dev)
pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
for _ in range(2): trainer.step(mel, cond, None, pad)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): trainer.step(mel, cond, None, pad)
torch.cuda.synchronize(); print("step ms", (time.perf_counter() - t0) / 3 * 1e3)
