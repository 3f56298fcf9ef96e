#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (not the driver's bench): the other BASELINE.json configs
on one MI355X.  Prints one JSON line per config."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mixgan_tts_amd as mg
from helpers import hot_path_configs, write_stats


def seeded_(module, seed):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if not p.requires_grad:
                continue
            fan = p[0].numel() if p.dim() > 1 else 1
            p.copy_(torch.randn(p.shape, generator=gen) * (fan ** -0.5 if p.dim() > 1 else 0.1))


def timeit(fn, warm, iters):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp()
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80, n_speakers=218)
    out = []
    # cfg1-shape / cfg2-shape inference steps, eager vs graph
    for name, model, T, B, L in (("one utterance naive T=4 B=1 L=1000", "naive", 4, 1, 1000),
                                 ("cfg1 naive T=4 B=4 L=256", "naive", 4, 4, 256),
                                 ("cfg2 naive T=4 B=16 L=1000", "naive", 4, 16, 1000),
                                 ("cfg3 shallow T=100 B=32 L=1000", "shallow", 100, 32, 1000)):
        gd = mg.GaussianDiffusion(*hot_path_configs(model, T, stats_dir=stats))
        seeded_(gd, 1)
        gd = gd.to(dev).eval()
        cond = torch.randn(B, L, 256, device=dev)
        pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
        coarse = (torch.rand(B, L, 80, device=dev) * 13.5 - 11.5) if model == "shallow" else None
        with torch.no_grad():
            gd(None, cond, None, pad, coarse)
        iters = 3 if T >= 100 else 20
        te = timeit(lambda: gd.sampling(keep_trace=False), 1 if T >= 100 else 3, iters)
        tg = timeit(lambda: gd.sampling(keep_trace=False, use_graph=True), 2, iters)
        out.append({"config": name, "sampling_eager_ms": round(te * 1e3, 3), "sampling_hipgraph_ms": round(tg * 1e3, 3),
                    "denoiser_steps_per_s_eager": round(T / te, 1), "denoiser_steps_per_s_graph": round(T / tg, 1),
                    "tflops_graph": round(23805952.0 * B * L * T / tg / 1e12, 1)})
        print(json.dumps(out[-1]), flush=True)
        del gd
        torch.cuda.empty_cache()
    # cfg4-shape training step per GPU: multi-speaker naive, B=8, L=1000 (global 64 over 8 GPUs)
    for name, ms, B, L in (("cfg4 per-GPU shard: multi-speaker naive train step B=8 L=1000", True, 8, 1000),
                           ("train step B=16 L=1000 single speaker", False, 16, 1000)):
        args, pre, mc, tr = hot_path_configs("naive", 4, multi_speaker=ms, stats_dir=stats)
        G = mg.GaussianDiffusion(args, pre, mc, tr)
        D = mg.JCUDiscriminator(pre, mc, tr)
        seeded_(G, 2)
        G, D = G.to(dev), D.to(dev)
        trainer = mg.HotPathTrainer(G, D, tr, mc)
        mel = torch.rand(B, L, 80, device=dev) * 13.5 - 11.5
        cond = torch.randn(B, L, 256, device=dev)
        spk = torch.randn(B, 256, device=dev) if ms else None
        pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
        ts = timeit(lambda: trainer.step(mel, cond, spk, pad), 2, 5)
        # per step: 2 denoiser fwd + 1 bwd (= 4 fwd-equivalents) + 4 D fwd + 2 D bwd passes
        flop = (4 * 23805952.0 + 8 * 645504.0) * B * L
        out.append({"config": name, "train_step_ms": round(ts * 1e3, 2), "train_steps_per_s": round(1 / ts, 2),
                    "approx_tflops": round(flop / ts / 1e12, 1)})
        print(json.dumps(out[-1]), flush=True)
        del G, D, trainer
        torch.cuda.empty_cache()


def lingops_bench():
    """the four index ops at B=16, ~64 words, ~1000 frames per utterance --
    device kernels vs the reference-style Python loops (oracle/refmath.py) run on the same GPU tensors
    (every int(...) in those loops is a device->host sync, as in the reference)."""
    from oracle import refmath as R
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(17)
    B, Tw, H = 16, 64, 256
    swl = torch.full((B,), Tw, dtype=torch.long)
    wb = torch.randint(1, 5, (B, Tw), generator=gen)
    dur = torch.randint(8, 24, (B, Tw), generator=gen)
    src_len = wb.sum(1)
    Tp = int(src_len.max())
    src = torch.randn(B, Tp, H, generator=gen).to(dev)
    xw = torch.randn(B, Tw, H, generator=gen).to(dev)
    wb, dur, swl, src_len = wb.to(dev), dur.to(dev), swl.to(dev), src_len.to(dev)
    L = mg.lingops
    lr = L.LengthRegulator()
    o, ml = lr(xw, dur, None)
    Lq = o.shape[1]
    mel_mask = torch.arange(Lq, device=dev)[None, :] < ml[:, None]
    q, kv = torch.zeros(B, Lq, 1, device=dev), torch.zeros(B, Tp, 1, device=dev)

    def ours():
        L.word_level_pooling(src, src_len, wb, swl, "mean", max_words=Tw)
        lr(xw, dur, Lq)
        L.get_mapping_mask(q, kv, dur, wb, swl)
        L.get_rel_coef(dur, swl, mel_mask)

    def loops():
        R.word_level_pooling(src, src_len, wb, swl, "mean")
        R.length_regulate(xw, dur, Lq)
        R.mapping_mask(Lq, Tp, dur, wb, swl).to(dev)
        R.rel_coef(dur.cpu(), swl.cpu(), mel_mask.cpu())  # (its index lists are host-built, as in the reference)

    t_ours = timeit(ours, 3, 50)
    t_loops = timeit(loops, 1, 3)
    bytes_moved = (src.numel() + B * Tw * H) * 4 + (xw.numel() + o.numel()) * 4 + B * Lq * Tp + B * Lq * 4
    print(json.dumps({"config": "f1 index ops, B=16, 64 words, %d phonemes, %d frames" % (Tp, Lq),
                      "device_kernels_us": round(t_ours * 1e6, 1), "python_loops_ms": round(t_loops * 1e3, 1),
                      "speedup": round(t_loops / t_ours, 1), "algorithmic_MB": round(bytes_moved / 1e6, 1),
                      "GBps": round(bytes_moved / t_ours / 1e9, 1)}), flush=True)


def hifigan_flops(h, L):
    """MACs*2 of one Generator.forward on L mel frames (hifigan/models.py:112-173)."""
    c = h["upsample_initial_channel"]
    fl = 2 * 80 * c * 7 * L
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        co = c // 2
        L *= u
        fl += 2 * c * co * (k // u) * L         # transposed conv: k/u live taps per output sample
        for kk, ds in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            fl += 2 * len(ds) * 2 * co * co * kk * L
        c = co
    return fl + 2 * c * 7 * L


def vocoder_bench():
    """HiFi-GAN V1 generator, B utterances of 1000 mel frames -> 256000 samples each."""
    import types
    from oracle import refmath as R
    dev = torch.device("cuda", 0)
    G = mg.vocoder.Generator(types.SimpleNamespace(**R.HIFIGAN_V1)).to(dev).eval()
    G.remove_weight_norm()
    for B, L in ((1, 1000), (16, 1000)):
        mel = torch.empty(B, 80, L, device=dev).uniform_(-11.5, 2.0)
        t = timeit(lambda: G(mel), 2, 5)
        fl = hifigan_flops(R.HIFIGAN_V1, L) * B
        print(json.dumps({"config": "f3 hifigan V1, B=%d, L=%d" % (B, L), "ms": round(t * 1e3, 2),
                          "audio_seconds_per_s": round(B * L * 256 / 22050 / t, 1),
                          "useful_TFLOPs": round(fl / t / 1e12, 1), "GFLOP": round(fl / 1e9, 1)}), flush=True)


def stock_eager_bench():
    """Second baseline of SURVEY.md section 8(d): the reference's op sequence as stock PyTorch-ROCm eager
    (MIOpen / rocBLAS; oracle/refmath.py moved to the GPU) on the same card, same shapes as the bench line,
    next to our HIP path.  fp32, TF32-style downcasts off (torch default)."""
    import types
    from oracle import refmath as R
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp()
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80, n_speakers=218)
    B, L = 16, 1000
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    seeded_(gd, 1)
    gd = gd.to(dev).eval()
    W = {k: v.detach() for k, v in gd.state_dict().items()}
    buf = {k: v.detach() for k, v in gd._buf().items()}
    x = torch.randn(B, 1, 80, L, device=dev)
    cond = torch.randn(B, 256, L, device=dev)
    t = torch.full((B,), 3, dtype=torch.long, device=dev)
    nz = torch.randn(B, 1, 80, L, device=dev)
    with torch.no_grad():
        ref = R.p_sample(W, buf, x, t, cond, None, nz)
        gd.noise_fn = lambda shape: nz
        ours = gd.p_sample(x, t, cond, None)
        gd.noise_fn = None
        ts = timeit(lambda: R.p_sample(W, buf, x, t, cond, None, nz), 3, 20)
        to = timeit(lambda: gd.p_sample(x, t, cond, None), 3, 20)
    print(json.dumps({"config": "cfg2 p_sample step B=16 L=1000", "stock_pytorch_rocm_eager_ms": round(ts * 1e3, 3),
                      "ours_ms": round(to * 1e3, 3), "speedup": round(ts / to, 2),
                      "max_abs_diff_same_noise": float((ours - ref).abs().max())}), flush=True)
    G = mg.vocoder.Generator(types.SimpleNamespace(**R.HIFIGAN_V1)).to(dev).eval()
    Wv = {k: v.detach() for k, v in G.state_dict().items()}
    for Bv in (1, 16):
        mel = torch.empty(Bv, 80, L, device=dev).uniform_(-11.5, 2.0)
        with torch.no_grad():
            err = float((G(mel) - R.hifigan_forward(Wv, mel)).abs().max())
            ts = timeit(lambda: R.hifigan_forward(Wv, mel), 1, 3)
        to = timeit(lambda: G(mel), 1, 3)
        print(json.dumps({"config": "hifigan V1 B=%d L=1000" % Bv, "stock_pytorch_rocm_eager_ms": round(ts * 1e3, 2),
                          "ours_ms": round(to * 1e3, 2), "speedup": round(ts / to, 2), "max_abs_diff": err}), flush=True)


def data_bench():
    """groups of 4 x 8 utterances (~800-1000 frames each) from a synthetic
    preprocessed_data tree: the reference's synchronous loop (np.load + collate + to_device on the training
    thread) vs the prefetching loader (loads / pinned staging / copies hidden behind the consumer)."""
    import types
    from mixgan_tts_amd import data as D
    rng = np.random.default_rng(5)
    d = tempfile.mkdtemp(prefix="mg_data_")
    for k in D.Dataset.KINDS:
        os.makedirs(os.path.join(d, k))
    with open(os.path.join(d, "speakers.json"), "w") as f:
        json.dump({"spk": 0}, f)
    lines, ids = [], {}
    for i in range(256):
        nw = int(rng.integers(12, 20))
        ppw = rng.integers(2, 6, nw)
        n_ph = int(ppw.sum())
        dur = rng.integers(8, 22, n_ph)
        Lm = int(dur.sum())
        arrs = {"mel": rng.uniform(-11.5, 2, (Lm, 80)).astype(np.float32), "pitch": rng.standard_normal(Lm),
                "energy": rng.standard_normal(Lm).astype(np.float32), "duration": dur, "phones_per_word": ppw,
                "attn_prior": rng.uniform(0, 1, (n_ph, Lm)).astype(np.float32)}
        for k, a in arrs.items():
            np.save(os.path.join(d, k, "spk-%s-u%03d.npy" % (k, i)), a)
        text = "t%d" % i
        ids[text] = rng.integers(1, 100, n_ph)
        lines.append("u%03d|spk|%s|raw" % (i, text))
    with open(os.path.join(d, "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    pre = {"dataset": "Synth", "path": {"preprocessed_path": d},
           "preprocessing": {"text": {"text_cleaners": []}, "speaker_embedder": "none"}}
    ds = D.Dataset("train.txt", types.SimpleNamespace(model="naive"), pre, {"multi_speaker": False},
                   {"optimizer": {"batch_size": 8}}, sort=True, drop_last=True,
                   text_to_sequence=lambda t, c: ids[t].tolist())
    smp = D.RankShardSampler(len(ds), 32, seed=1)
    dev = torch.device("cuda", 0)
    work = torch.randn(4096, 4096, device=dev)

    def run(loader_kind, sync_each):
        def consume(batchs):   # stand-in for the training step: 4 sub-batches x 8 GEMMs of GPU work per group
            for k in range(4):
                for _ in range(8):
                    (work @ work).sum()
                if batchs:
                    batchs[k][11].sum()
                if sync_each:  # a loop that reads its losses back every step (`.item()`)
                    torch.cuda.synchronize()
        if loader_kind == "none":
            it = ([] for _ in smp)
        elif loader_kind == "sync":
            it = ([D.to_device(b, dev) for b in ds.collate_fn([ds[i] for i in idxs])] for idxs in smp)
        else:
            it = iter(loaders[loader_kind])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for batchs in it:
            consume(batchs)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    loaders = {"prefetch_w1": D.PrefetchLoader(ds, smp, dev, depth=2, workers=1),
               "prefetch_w2": D.PrefetchLoader(ds, smp, dev, depth=3, workers=2)}
    out = {"config": "f2 data path: 8 groups of 4x8 utterances (~900 frames), warm page cache"}
    for sync_each in (False, True):
        tag = "step_syncs" if sync_each else "async_steps"
        for kind in ("none", "sync", "prefetch_w1", "prefetch_w2"):
            run(kind, sync_each)       # warm-up epoch (page cache, allocator, pinned arenas)
            t = min(run(kind, sync_each) for _ in range(2))
            out["%s/%s_ms" % (tag, {"none": "gpu_work_only", "sync": "reference_style_loop"}.get(kind, kind))] = round(t, 1)
    print(json.dumps(out), flush=True)


def aux_bench():
    """the aux-mode acoustic step (Decoder 6 FFT blocks -> mel_linear -> PostNet ->
    diffuse_trace -> losses -> backward -> ScheduledOptim) at B=8, L=1000 on the HIP path, next to the same
    chain as stock PyTorch-ROCm eager (oracle on the GPU, torch autograd)."""
    from oracle import refmath as R, schedule as S
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp()
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
    args, pre, mc, tr = hot_path_configs("aux", 4, stats_dir=stats, max_seq_len=1000)
    tr = dict(tr)
    tr["optimizer_fs2"] = {"betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "warm_up_step": 4000,
                           "anneal_steps": [300000], "anneal_rate": 0.3}
    tr.setdefault("optimizer", {})["grad_clip_thresh"] = 1.0
    B, L = 8, 1000
    net = mg.MixGANTTS(args, pre, mc, tr).to(dev).train()
    params = [p for n_, p in net.named_parameters() if n_.split(".")[0] in ("decoder", "mel_linear", "postnet")
              and p.requires_grad]
    trainer = mg.AuxTrainer(net, tr, mc, params=params)
    cond = torch.randn(B, L, 256, device=dev)
    mel = torch.rand(B, L, 80, device=dev) * 13.5 - 11.5
    pad = torch.zeros(B, L, dtype=torch.bool, device=dev)
    t_ours = timeit(lambda: trainer.step(cond.clone().requires_grad_(), mel, pad), 2, 5)
    if os.environ.get("MG_AUX_OURS_ONLY"):
        print(json.dumps({"ours_ms": round(t_ours * 1e3, 2)}), flush=True)
        return
    # stock eager: same parameters, torch autograd + torch dropout masks + torch Adam
    W = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "position_enc" not in k)
         for k, v in net.state_dict().items() if k.split(".")[0] in ("decoder", "mel_linear", "postnet")}
    buf = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in
           S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = net.diffusion.spec_min, net.diffusion.spec_max
    opt = torch.optim.Adam([w for w in W.values() if w.requires_grad], betas=(0.9, 0.98), eps=1e-9)
    drop = lambda shape, p: (torch.rand(shape, device=dev) >= p).float()  # noqa: E731

    class _Noise:
        def next(self, shape=None):
            return torch.randn(B, 1, 80, L, device=dev)
        __call__ = next

    def stock():
        c = cond.clone().requires_grad_()
        tape = R.NoiseTape([torch.randn(B, 1, 80, L, device=dev) for _ in range(4)])
        ml, pl, _ = R.aux_acoustic_losses(W, buf, c, mel, pad, 1000, 4, tape, drop)
        (ml + pl).backward()
        torch.nn.utils.clip_grad_norm_([w for w in W.values() if w.requires_grad], 1.0)
        opt.step()
        opt.zero_grad()

    t_stock = timeit(stock, 2, 5)
    print(json.dumps({"config": "f4 aux acoustic train step, B=8, L=1000 (decoder 6 FFT blocks + PostNet, fwd+bwd+opt)",
                      "ours_ms": round(t_ours * 1e3, 2), "stock_pytorch_rocm_eager_ms": round(t_stock * 1e3, 2),
                      "speedup": round(t_stock / t_ours, 2)}), flush=True)


def attention_bench():
    """BASELINE configs[4] attention shape: B=16 per GPU, L=4000, 2 heads x 128: exact fp32 MFMA kernel vs the
    fp16-operand one (both streaming softmax, no L x L tensor)."""
    dev = torch.device("cuda", 0)
    for B, L in ((16, 1000), (16, 4000)):
        qkv = torch.randn(B, 768, L, device=dev)
        pad = torch.zeros(B, L, dtype=torch.uint8, device=dev)
        fl = 4.0 * L * L * 128 * 2 * B
        t32 = timeit(lambda: mg.ops.attention(qkv, pad, 2, 128), 2, 5)
        t16 = timeit(lambda: mg.ops.attention(qkv, pad, 2, 128, precision="f16"), 2, 5)
        err = float((mg.ops.attention(qkv, pad, 2, 128) - mg.ops.attention(qkv, pad, 2, 128, precision="f16")).abs().max())
        print(json.dumps({"config": "attention B=%d L=%d" % (B, L), "fp32_ms": round(t32 * 1e3, 3),
                          "fp32_TFLOPs": round(fl / t32 / 1e12, 1), "f16_ms": round(t16 * 1e3, 3),
                          "f16_TFLOPs": round(fl / t16 / 1e12, 1), "max_abs_diff": err}), flush=True)


def e2e_bench():
    """Conditioner -> mel -> waveform on the HIP path (`synthesize.py` minus the out-of-scope text front-end and
    linguistic encoder): naive model, T=4 sampling (hipGraph) + HiFi-GAN + int16 conversion, 1000-frame utterances."""
    import types
    from oracle import refmath as R   # only for the HiFi-GAN V1 config constants
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp()
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    seeded_(gd, 1)
    gd = gd.to(dev).eval()
    voc = mg.vocoder.Generator(types.SimpleNamespace(**R.HIFIGAN_V1)).to(dev).eval()
    voc.remove_weight_norm()
    mc = {"vocoder": {"model": "HiFi-GAN", "speaker": "LJSpeech"}}
    pre = {"preprocessing": {"audio": {"max_wav_value": 32768.0}}}
    for B, L in ((1, 1000), (16, 1000)):
        cond = torch.randn(B, L, 256, device=dev)
        pad = torch.zeros(B, L, dtype=torch.bool, device=dev)

        def run():
            with torch.no_grad():
                mel = gd(None, cond, None, pad, None)[0]          # inference branch: T p_sample steps -> [B, L, 80]
                return mg.vocoder.vocoder_infer(mel.transpose(1, 2).contiguous(), voc, mc, pre)
        t = timeit(run, 2, 5)
        audio = B * L * 256 / 22050.0
        print(json.dumps({"config": "e2e cond -> mel (T=4) -> wav, B=%d, L=%d" % (B, L), "ms": round(t * 1e3, 2),
                          "audio_s": round(audio, 1), "real_time_factor": round(t / audio, 5),
                          "audio_seconds_per_s": round(audio / t, 1)}), flush=True)


def elementwise_bench():
    """Rows a4-a6 (norm/denorm, diffuse_fn / q_sample, q_posterior_sample) are HBM-bound: GB/s against the 8 TB/s
    spec at the bench size (B=16, L=1000: 20-26 MB per launch, launch-latency-limited) and at a size that fills the
    memory system (B=128, L=4000: the global batch of configs[4])."""
    dev = torch.device("cuda", 0)
    d = tempfile.mkdtemp()
    stats = write_stats(d, [-11.5] * 80, [2.0] * 80)
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats)).to(dev)
    buf = gd._buf()
    for B, L in ((16, 1000), (128, 4000)):
        M = 80
        mel = torch.rand(B, L, M, device=dev) * 13.5 - 11.5
        x = torch.randn(B, M, L, device=dev)
        x2 = torch.randn(B, M, L, device=dev)
        nz = torch.randn(B, M, L, device=dev)
        t = torch.randint(0, 4, (B,), device=dev)
        out = torch.empty_like(x)
        res = {"config": "elementwise kernels B=%d L=%d" % (B, L)}
        n = B * L * M * 4
        for name, fn, nbytes in (
                ("diffuse (mel + noise -> x_t)", lambda: mg.ops.diffuse(mel, t, nz, None, buf), 3 * n),
                ("posterior (x0, x_t, noise -> x_t-1)", lambda: mg.ops.posterior_sample(x, x2, t, nz, None, buf, out=out), 4 * n),
                ("transpose + denorm ([B,M,L] -> [B,L,M])", lambda: mg.ops.transpose_bml(x, True, 2, gd.spec_min, gd.spec_max), 2 * n)):
            tt = timeit(fn, 3, 30)
            res[name] = {"us": round(tt * 1e6, 1), "GBps": round(nbytes / tt / 1e9), "frac_of_8TBps": round(nbytes / tt / 8e12, 3)}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "elementwise":
        elementwise_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "e2e":
        e2e_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "attention":
        attention_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "aux":
        aux_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "data":
        data_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "stock":
        stock_eager_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "lingops":
        lingops_bench()
    elif len(sys.argv) > 1 and sys.argv[1] == "vocoder":
        vocoder_bench()
    else:
        main()
