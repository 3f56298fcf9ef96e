"""Does the HIP training step learn?  A task the denoiser can solve: the conditioner carries the (normalised) target
mel in its first 80 channels, so x0 can be read off it; the mel L1 must fall well below its starting value."""
import os, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
B, L = 8, 256
args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
tr = dict(tr); tr["optimizer"] = dict(tr["optimizer"], init_lr_G=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
G = mg.GaussianDiffusion(args, pre, mc, tr).to(dev); D = mg.JCUDiscriminator(pre, mc, tr).to(dev)
trainer = mg.HotPathTrainer(G, D, tr, mc)
gen = torch.Generator(device=dev).manual_seed(0)
first = last = None
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    # smooth random "spectrograms": low-pass noise along time, inside the stats range
    z = torch.randn(B, L // 8 + 1, 80, device=dev, generator=gen)
    mel = torch.nn.functional.interpolate(z.transpose(1, 2), size=L, mode="linear").transpose(1, 2) * 3.0 - 5.0
    mel = mel.clamp(-11.5, 2.0).contiguous()
    cond = torch.zeros(B, L, 256, device=dev)
    cond[:, :, :80] = (mel + 11.5) / 13.5 * 2 - 1
    pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
    out = trainer.step(mel, cond, None, pad)
    if step % 25 == 0:
        print(step, {k: round(float(v), 4) for k, v in out.items()}, flush=True)
    if step == 0: first = float(out["mel_loss"])
    last = float(out["mel_loss"])
print("mel_loss %.3f -> %.3f" % (first, last))
