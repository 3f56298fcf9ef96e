"""Short stability soak of the GAN step: 200 steps at the cfg4 per-GPU shard, losses finite, no memory growth."""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5] * 80, [2.0] * 80, n_speakers=218)
B, L = 8, 1000
args, pre, mc, tr = hot_path_configs("naive", 4, multi_speaker=True, stats_dir=stats)
G = mg.GaussianDiffusion(args, pre, mc, tr).to(dev); D = mg.JCUDiscriminator(pre, mc, tr).to(dev)
trainer = mg.HotPathTrainer(G, D, tr, mc)
gen = torch.Generator(device=dev).manual_seed(0)
mem0 = None
for step in range(200):
    mel = torch.rand(B, L, 80, device=dev, generator=gen) * 13.5 - 11.5
    cond = torch.randn(B, L, 256, device=dev, generator=gen)
    spk = torch.randn(B, 256, device=dev, generator=gen)
    pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
    out = trainer.step(mel, cond, spk, pad)
    if step % 50 == 0 or step == 199:
        vals = {k: float(v) for k, v in out.items()}
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
        mem = torch.cuda.memory_allocated() / 2**20
        print(step, {k: round(v, 4) for k, v in vals.items()}, "MiB", round(mem), flush=True)
        if step == 50: mem0 = mem
        if step > 50: assert mem <= mem0 * 1.05 + 64, (mem, mem0)
assert all(torch.isfinite(p).all() for p in list(G.parameters()) + list(D.parameters()))
print("soak ok")
