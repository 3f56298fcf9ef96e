"""Which lines of the package issue the small torch ops of the training step (copies, adds, fills)?
A TorchDispatchMode logs every aten op of one step with the innermost package frame that caused it."""
import collections, os, sys, tempfile, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(); stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
B, L = 8, 1000
args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
G = mg.GaussianDiffusion(args, pre, mc, tr); D = mg.JCUDiscriminator(pre, mc, tr)
G, D = G.to(dev), D.to(dev)
trainer = mg.HotPathTrainer(G, D, tr, mc)
mel = torch.rand(B, L, 80, device=dev) * 13.5 - 11.5
cond = torch.randn(B, L, 256, device=dev)
pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
for _ in range(3): trainer.step(mel, cond, None, pad)
torch.cuda.synchronize()
seen = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if any(k in name for k in ("copy", "fill", "zero", "add", "mul", "div", "cat", "clone", "sub", "sum", "bitwise", "ones", "full", "neg", "clamp", "reciprocal")):
            fr = [f for f in traceback.extract_stack() if "mixgan-tts_amd" in f.filename]
            where = "%s:%d" % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else "(autograd engine / torch)"
            seen[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    trainer.step(mel, cond, None, pad)
torch.cuda.synchronize()
for (name, where), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d  %-28s %s" % (n, name, where))
