"""HiFi-GAN V1 generator on ONE 1000-frame utterance: the shape of a synthesize.py call (profile target)."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from oracle import refmath as R
G = mg.vocoder.Generator(types.SimpleNamespace(**R.HIFIGAN_V1)).cuda().eval()
G.remove_weight_norm()
mel = torch.empty(1, 80, 1000, device="cuda").uniform_(-11.5, 2.0)
with torch.no_grad():
    for _ in range(3): G(mel)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): G(mel)
    e1.record(); torch.cuda.synchronize()
print("vocoder B=1 L=1000: %.3f ms" % (e0.elapsed_time(e1) / 10))
