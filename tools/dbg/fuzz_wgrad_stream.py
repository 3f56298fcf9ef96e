"""Randomised shapes for the streaming weight-gradient kernel (and its folded bias sums) against the split kernel."""
import ctypes, os, random, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mixgan_tts_amd as mg
lib = mg._lib.lib()
cp = lambda t: ctypes.c_void_p(t.data_ptr())
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
gen = torch.Generator(device="cuda").manual_seed(1)
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    K = rnd.choice([1, 3])
    Co = rnd.choice([128, 256, 384, 512]); Ci = rnd.choice([128, 256, 512]) if K == 3 else rnd.choice([256, 512])
    G = rnd.choice([1, 1, 2, 5])
    L = rnd.choice([4, 8, 36, 60, 64, 68, 100, 128, 256, 1000, 1004, 2052])
    FT = 64 if K == 3 else 32
    tiles = G * (Co // (128 if K == 3 or Co % 256 or Ci % 256 else 256)) * (Ci // (128 if K == 3 else 256))
    need = 2048 // max(1, tiles * -(-L // FT)) + 1
    B = min(max(need, rnd.randint(1, 6)), 4000)
    if B * G * (Co + Ci) * L > 3e8: continue
    shared = rnd.random() < 0.3
    dy = torch.randn(B, G * Co, L, device="cuda", generator=gen)
    x = torch.randn(G, B, Ci, L, device="cuda", generator=gen)
    scratch = torch.empty(lib.mg_conv1d_wgrad_grouped_scratch_floats(Co, Ci, K, G), device="cuda")
    res = {}
    for stream in ("1", "0"):
        os.environ["MG_WGRAD_STREAM"] = stream
        dw = torch.empty(G, Co, Ci, K, device="cuda"); db = torch.zeros(G, Co, device="cuda")
        mg._lib.check(lib.mg_conv1d_wgrad_grouped_bias(cp(dy), G * Co * L, 0 if shared else Co * L, cp(x), Ci * L, B * Ci * L, cp(dw), 0,
                                                       cp(db), 0, cp(scratch), G, B, Co, Ci, L, L, K, 1, (K - 1) // 2, 1.0, 0, None))
        res[stream] = (dw, db)
    same = torch.equal(res["1"][0], res["0"][0])
    e_w = (res["1"][0] - res["0"][0]).abs().max().item() / res["0"][0].abs().max().item()
    e_b = (res["1"][1] - res["0"][1]).abs().max().item() / max(res["0"][1].abs().max().item(), 1e-9)
    worst = max(worst, e_w, e_b)
    print("K=%d Co=%d Ci=%d G=%d B=%d L=%d shared=%d  streamed=%s  dw %.1e  db %.1e" % (K, Co, Ci, G, B, L, shared, not same, e_w, e_b), flush=True)
    assert e_w < 2e-5 and e_b < 2e-5
print("worst", worst)
