#!/usr/bin/env python3
"""Where the single-launch denoiser kernel spends its cycles: runs the TIMING instantiation (lane 0 of every workgroup
stamps clock64() at each phase boundary of every layer) at B=16, L=1000 and prints per-phase means.  Tool, not product."""
import ctypes
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mixgan_tts_amd as mg  # noqa: E402
from helpers import hot_path_configs, write_stats  # noqa: E402

B, L, NL = int(os.environ.get("B", 16)), int(os.environ.get("L", 1000)), 20
with tempfile.TemporaryDirectory() as d:
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=write_stats(d, [-11.5] * 80, [2.0] * 80)))
gen = torch.Generator().manual_seed(1)
with torch.no_grad():
    for p in gd.denoise_fn.parameters():
        p.copy_(torch.randn(p.shape, generator=gen) * (p[0].numel() ** -0.5 if p.dim() > 1 else 0.1))
gd = gd.cuda().eval()
den = gd.denoise_fn
x = torch.randn(B, 80, L, device="cuda")
cond = torch.randn(B, 256, L, device="cuda")
t = torch.full((B,), 3, device="cuda", dtype=torch.long)
NT = int(os.environ.get('MG_PERSIST_NT', 64 if B * ((L + 63) // 64) > 128 else 32))
tiles = B * ((L + NT - 1) // NT)
# CPROJ=1: a step that READS the loop's conditioner projections instead of computing them
cproj = den.cond_projection(cond) if os.environ.get("CPROJ") == "1" else None


def launch():
    if cproj is None:
        den.run(x, t, cond, None)
    else:
        gd._p_sample_bml(x, t, cond, None, None, cproj=cproj)


for _ in range(3):
    launch()
stamps = torch.zeros(tiles * (NL + 2) * 12, dtype=torch.int64, device="cuda")
lib = mg.lib()
lib.mg_debug_persist_stamps.argtypes = [ctypes.c_void_p]
lib.mg_debug_persist_stamps(ctypes.c_void_p(stamps.data_ptr()))
launch()
torch.cuda.synchronize()
lib.mg_debug_persist_stamps(ctypes.c_void_p(0))
s = stamps.cpu().numpy().reshape(tiles, NL + 2, 12).astype(np.float64)
t0 = s[:, 0, 0].min()
lay = s[:, 1:NL + 1, :]                       # [tiles, NL, 12]
names = [  # NT-independent
         "GEMM1 (init + k loop)", "barrier (prev GEMM3 done)", "h write + barrier", "publish + GEMM2 centre",
         "halo sweep (wave 0)", "barrier (halo in place)", "GEMM2 outer", "barrier (hT read done)", "gate + addends",
         "barrier (g complete)", "GEMM3"]
d = np.diff(lay, axis=2)                      # [tiles, NL, 11]
out = {"B": B, "L": L, "tiles": tiles,
       "kernel_cycles": float(s[:, NL + 1, 1].max() - t0),
       "prologue_cycles_mean": float((s[:, 0, 1] - s[:, 0, 0]).mean()),
       "start_skew_cycles": float(s[:, 0, 0].max() - t0),
       "layer_cycles_mean": float((lay[:, :, 11] - lay[:, :, 0]).mean()),
       "between_layers_mean": float((lay[:, 1:, 0] - lay[:, :-1, 11]).mean()),
       "tail_cycles_mean": float((s[:, NL + 1, 1] - s[:, NL + 1, 0]).mean()),
       "phases_mean_cycles": {n: float(d[:, :, i].mean()) for i, n in enumerate(names)},
       "phases_p95_cycles": {n: float(np.percentile(d[:, :, i], 95)) for i, n in enumerate(names)},
       "mfma_ideal_cycles_alone": {"GEMM1": 256 * 64, "GEMM2 centre": 512 * 64, "GEMM2 outer": 1024 * 64, "GEMM3": 512 * 64}}
pc = lambda q: {n: float(np.percentile(d[:, :, i], q)) for i, n in enumerate(names)}  # noqa: E731
out["phases_p5_cycles"], out["phases_p50_cycles"] = pc(5), pc(50)
hw = s[:, 0, 2].astype(np.int64)
xcc = s[:, 0, 3].astype(np.int64) & 0xF
cu = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)
slots = {}
for c, h in zip(cu.tolist(), hw.tolist()):
    slots.setdefault(c, []).append(h & 0xF)
out["distinct_cus"] = len(slots)
out["wgs_per_cu_hist"] = {str(k): int(v) for k, v in zip(*np.unique([len(v) for v in slots.values()], return_counts=True))}
out["slot_pairs_hist"] = {str(k): int(v) for k, v in zip(*np.unique([str(sorted(v)) for v in slots.values()], return_counts=True))}
out["flags"] = os.environ.get("MG_PERSIST_FLAGS", "default")
out["layer_cycles_by_role"] = {str(r): float((lay[(hw & 1) == r][:, :, 11] - lay[(hw & 1) == r][:, :, 0]).mean()) for r in (0, 1)}
out["tile_total_cycles"] = {"mean": float((s[:, NL + 1, 1] - s[:, 0, 0]).mean()), "max": float((s[:, NL + 1, 1] - s[:, 0, 0]).max()),
                            "min": float((s[:, NL + 1, 1] - s[:, 0, 0]).min())}
for k in ("kernel_cycles", "start_skew_cycles"):
    out.pop(k)
print(json.dumps(out, indent=1))
