#!/usr/bin/env python3
"""bench.py -- denoiser steps/sec on the MI355X HIP path (BASELINE.json metric).

A "step" is one reverse-diffusion step on a whole batch: Denoiser.forward + clamp +
q_posterior_sample (the reference's p_sample, model/diffusion.py:121-129), at BASELINE
configs[1]: LJSpeech 'naive' model, B=16 per GPU, 80 mel bins, L=1000 frames, fp32.
Inputs are synthetic and already resident in HBM; weights are seeded random (no checkpoints
ship with the reference).  N>1: one process per GPU, each with its own batch (weak scaling);
the path has no data-path collective at inference.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the dominant
kernel, the k=3 gated conv, from HIP events recorded on the launch stream inside the timed
region) and `cpu_baseline` (the CPU oracle timed on this box's host cores, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

B_PER_GPU = 16
L_FRAMES = 1000
MEL = 80
FLOP_PER_FRAME = 23_805_952            # Denoiser.forward per frame
COND_PROJ_FLOP_PER_FRAME = 20 * 2 * 256 * 256   # of which: the 20 conditioner projections (model/blocks.py:1160)
K3_FLOP_PER_FRAME = 2 * 512 * 768      # generic path's dominant kernel: Conv1d(256->512, k=3) of one layer
# fused path's dominant kernel = one whole residual layer (model/blocks.py:1157-1176):
# k=3 conv 256->512 + conditioner 1x1 256->256 + output 1x1 256->512 (algorithmic; halo MFMAs not counted)
LAYER_FLOP_PER_FRAME = 2 * 512 * 768 + 2 * 256 * 256 + 2 * 512 * 256
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# roofline.traffic is NOT measured in this run (PMC counters need their own rocprofv3 passes): it is read from the
# tracked profile summary profiles/bench_traffic.json (written from tools/profile_bench.sh's passes over this same
# command), and the line says so in roofline.traffic_source.  Default shape, fp32, single-launch kernel only.


def tracked_traffic(hoisted=True):
    try:
        with open(os.path.join(ROOT, "profiles", "bench_traffic.json")) as f:
            t = json.load(f)
        if not hoisted:
            t = t["projection_in_every_step"]
        return int(t["traffic_bytes_per_launch"]), t["source"]
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--frames", type=int, default=L_FRAMES)
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32: exact fp32 MFMA (headline). bf16x3: residual-layer GEMMs as 3-term bf16-split MFMA "
                         "products with fp32 accumulate (opt-in, parity 2e-4); reported under 'alt'")
    ap.add_argument("--no-alt", action="store_true", help="skip the second measurement in the other precision")
    ap.add_argument("--project-per-step", action="store_true",
                    help="keep the conditioner projections inside every p_sample launch (no per-loop hoisting)")
    ap.add_argument("--workload", choices=["denoise", "train"], default="denoise",
                    help="denoise: the BASELINE metric (default).  train: BASELINE configs[3] -- multi-speaker naive GAN "
                         "training step (G+D), global batch 8*N sharded over N ranks, gradients all-reduced over RCCL")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)                 # the parent never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)"
                 % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # MG_BENCH_EXCHANGE=1 (train workload): a process group on a world of ONE, collectives forced -- the gradient
    # exchange runs through RCCL as it does on 8 GPUs (a self-copy per collective), so comm_exposed_ms has a meaning
    force_pg = world == 1 and args.workload == "train" and os.environ.get("MG_BENCH_EXCHANGE") == "1"
    if force_pg and "MASTER_PORT" not in os.environ:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        s_.close()
    if world > 1 or force_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MG_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): every rank on cuda:0 over gloo
        share = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
        if not share and torch.cuda.device_count() < world:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) visible (MG_BENCH_SHARE_GPU=1 rehearses all ranks on "
                     "cuda:0 over gloo)" % (world, torch.cuda.device_count()))
        dev_index = 0 if share else local_rank
        torch.cuda.set_device(dev_index)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        world = dist.get_world_size()                  # what the process group reports, not what the env claimed
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    import mixgan_tts_amd as mg
    from mixgan_tts_amd import ops, _lib
    from helpers import hot_path_configs, write_stats

    if args.workload == "train":
        return train_workload(args, mg, dev, dist, rank, world)

    B, L = args.batch, args.frames
    with tempfile.TemporaryDirectory() as d:
        stats = write_stats(d, [-11.5] * MEL, [2.0] * MEL)
        gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    # seeded N(0, 1/fan_in) weights incl. a non-zero output projection (the shipped init zeroes it)
    gen = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for p in gd.denoise_fn.parameters():
            fan = p[0].numel() if p.dim() > 1 else 1
            p.copy_(torch.randn(p.shape, generator=gen) * (fan ** -0.5 if p.dim() > 1 else 0.1))
        # keep the predicted x_0 mostly inside the clamp of model/diffusion.py:126-127 (a trained denoiser's output
        # is a normalised mel in [-1, 1]): a saturated clamp would hide denoiser errors from the parity leg
        gd.denoise_fn.output_projection.conv.weight.mul_(0.25)
    gd = gd.to(dev).eval()
    den = gd.denoise_fn
    rng = np.random.default_rng(1234 + rank)
    cond = torch.from_numpy(rng.standard_normal((B, 256, L)).astype(np.float32)).to(dev)
    x = torch.from_numpy(rng.standard_normal((B, MEL, L)).astype(np.float32)).to(dev)
    buf = gd._buf()
    T = gd.num_timesteps
    ts = [torch.full((B,), i, device=dev, dtype=torch.long) for i in range(T)]
    x0 = torch.empty_like(x)
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    noise = torch.empty_like(x)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def single_launch(precision):
        """Does Denoiser.forward run as the one persistent kernel (csrc/denoiser_persist.h)?  Same rule as the library."""
        return precision == "fp32" and os.environ.get("MG_DENOISER_PERSIST", "1")[:1] != "0" and (L + 31) // 32 <= 128 \
            and os.environ.get("MG_DENOISER_GENERIC") is None

    def hoists(precision):
        """Does the sampling loop project the conditioner once per T steps (GaussianDiffusion._loop_cond_projection)?"""
        return precision == "fp32" and not args.project_per_step and single_launch(precision) \
            and os.environ.get("MG_COND_PREPROJECT", "1") != "0"

    def measure(precision, hoist=None):
        den.precision = precision
        pk = den.packed_weights()
        hoist = hoists(precision) if hoist is None else hoist
        hoist = hoist and den.has_cond_projection(pk)
        cbuf = gd._loop_cond_buffer(cond, pk) if hoist else None
        loop_ts = gd._loop_ts(B, dev)          # [T, B]: row k = t of step k of a loop = T-1-k
        state = {"vecs": None}

        def step_(i, xin, xout):
            # one library call per step: Denoiser.forward + clamp + posterior sample with in-kernel noise
            # (mg_denoiser_psample; a single kernel launch on the fp32 path).  The steps walk t = T-1 .. 0 over and
            # over, as back-to-back sampling loops do (model/diffusion.py:133-147); like GaussianDiffusion.sampling, the
            # first step of every loop (t = T-1) projects the conditioner as the reference's every step does and leaves
            # the 20 layers' projections in a buffer, the T-1 steps behind it read them (bit-identical to projecting in
            # every step).  Likewise the step-dependent vectors (step embedding -> MLP -> per-layer projections): a loop
            # knows its T values of t, so its first step's call computes them for all T steps in one set of launches.
            # Every loop recomputes both: nothing is carried over from one loop to the next.
            k = i % T
            if hoist and k == 0:
                state["vecs"] = den.step_vectors(loop_ts, None, pk)
            gd._p_sample_bml(xin, loop_ts[k], cond, None, None, True, out=xout, packed=pk,
                             cproj=None if k == 0 else cbuf, cproj_out=cbuf if k == 0 else None,
                             step_vectors=(state["vecs"], k, T) if hoist else None)

        cur, nxt = x, 0           # x_{t-1} never aliases x_t: ping-pong between the two buffers
        for i in range(args.warmup):
            step_(i, cur, bufs[nxt])
            cur, nxt = bufs[nxt], nxt ^ 1
        single = single_launch(precision)
        n_layers = 1 if single else len(den.residual_layers)
        Lh = _lib.lib()
        if not os.environ.get("MG_BENCH_NO_EVENTS"):
            # HIP-event brackets (on the launch stream, inside the timed region) around the dominant kernel: every
            # launch of the single-launch forward; every 7th launch of the per-layer kernel (7 is coprime with the
            # 20 layers, so every layer is sampled: bracketing all 800 launches costs ~4 % of the step)
            _lib.check(Lh.mg_profile_begin_sampled(args.steps * n_layers, 1 if single else 7))
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_(i, cur, bufs[nxt])
            cur, nxt = bufs[nxt], nxt ^ 1
        sync()
        dt = time.perf_counter() - t0
        ms = (ctypes.c_float * (args.steps * n_layers))()
        n_ev = Lh.mg_profile_end(ms, args.steps * n_layers)
        assert torch.isfinite(cur).all(), "non-finite output"
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        k_ms = float(np.mean(np.frombuffer(ms, dtype=np.float32)[:n_ev])) if n_ev > 0 else float("nan")
        return float(tmax.item()), k_ms, int(n_ev)

    dt, k_ms, n_ev = measure(args.precision)
    per_step = None
    if hoists(args.precision) and not args.no_alt:
        per_step = measure(args.precision, False)     # the same steps with the projection left inside every launch
    alt = None
    if not args.no_alt:
        other = "bf16x3" if args.precision == "fp32" else "fp32"
        alt = (other,) + measure(other)

    if rank == 0:
        BF16_PEAK = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA

        def roof(precision, k_ms, n_ev, dt, hoisted=False):
            generic = os.environ.get("MG_DENOISER_GENERIC") is not None and precision == "fp32"
            single = single_launch(precision)
            # `achieved` is the contract's figure: the ALGORITHMIC flops of the step a launch delivers (SURVEY.md section
            # 8d: the reference's Denoiser.forward, 23,805,952 per frame) / the launch's duration.  With the conditioner
            # projections hoisted, T-1 of T launches EXECUTE 20 x 2 x 256 x 256 fewer flops per frame than that:
            # `executed` prices the matrix pipes' actual load (what SQ_VALU_MFMA_BUSY sees).
            in_launch = FLOP_PER_FRAME - (COND_PROJ_FLOP_PER_FRAME * (T - 1) / T if hoisted else 0)
            k_flop = (FLOP_PER_FRAME if single
                      else K3_FLOP_PER_FRAME if generic else LAYER_FLOP_PER_FRAME) * B * L
            achieved = k_flop / (k_ms * 1e-3) / 1e12
            whole = FLOP_PER_FRAME * B * L * args.steps / dt / 1e12
            if precision == "fp32":
                name = ("denoiser_persist_kernel (the whole p_sample step in one launch: input projection, 20 residual "
                        "layers, skip / output projections, clamp + posterior sample with Philox noise), "
                        "v_mfma_f32_32x32x2_f32" if single else
                        "conv_mfma_kernel<K=3> 256->512 + GLU gate epilogue" if generic else
                        "resblock_fused_kernel (one residual layer: cond 1x1 + k3 conv + gate + out 1x1 + res/skip), "
                        "v_mfma_f32_32x32x2_f32")
                peak = FP32_MFMA_PEAK_TFLOPS
                extra = {}
            else:
                name = ("resblock_split_kernel (same layer; each product = 3 v_mfma_f32_32x32x16_bf16 on hi/lo "
                        "bf16 pairs, fp32 accumulate)")
                peak = BF16_PEAK
                extra = {"executed_mfma_tflops": round(achieved * 3 * 2432 / 2304, 1),
                         "vs_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 3)}
            r = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                 "frac": round(achieved / peak, 4),
                 "traffic": None, "kernel_ms": round(k_ms, 4),
                 "launches_timed": n_ev, "whole_step_tflops": round(whole, 2),
                 "whole_step_frac": round(whole / peak, 4)}
            if hoisted and single:
                ex = in_launch * B * L / (k_ms * 1e-3) / 1e12
                r["executed"] = {"flop_per_frame": in_launch, "tflops": round(ex, 2), "frac": round(ex / peak, 4)}
                r["note"] = ("achieved = the reference step's %d flop/frame per launch (SURVEY 8d) / mean launch duration; "
                             "the 20 conditioner projections (%d flop/frame, x_t-independent) are computed by the first "
                             "launch of every %d-step loop and read by the other %d, so the launches execute less than "
                             "they deliver: `executed` is the matrix pipes' own load (mean over both kinds of launches)"
                             % (FLOP_PER_FRAME, COND_PROJ_FLOP_PER_FRAME, T, T - 1))
            if single and precision == "fp32" and (B, L) == (B_PER_GPU, L_FRAMES):
                r["traffic"], r["traffic_source"] = tracked_traffic(hoisted)
            r.update(extra)
            return r

        value = world * args.steps / dt
        dtype = {"fp32": "f32", "bf16x3": "bf16x3-split products, f32 accumulate and I/O"}
        line = {
            "metric": "denoiser steps/sec (80-mel, L=1000 frames, B=16 per GPU)",
            "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype[args.precision], "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: LJSpeech naive, p_sample step (Denoiser.forward + clamp + "
                                   "posterior sample), B=%d/GPU, L=%d, 80 mel, T=4 schedule" % (B, L),
                       "parallelism": "replicas x%d (batch-sharded, no collective)" % world},
            "roofline": roof(args.precision, k_ms, n_ev, dt, hoists(args.precision)),
        }
        if hoists(args.precision):
            line["config"]["cond_projection"] = ("by the first step of each %d-step sampling loop, read by the "
                                                 "other %d (as GaussianDiffusion.sampling does)" % (T, T - 1))
        if per_step is not None:
            p_dt, p_kms, p_nev = per_step
            line["projection_in_every_step"] = {
                "value": round(world * args.steps / p_dt, 3), "unit": "steps/s",
                "ms_per_step": round(p_dt / args.steps * 1e3, 4), "roofline": roof(args.precision, p_kms, p_nev, p_dt)}
        if alt is not None:
            a_prec, a_dt, a_kms, a_nev = alt
            line["alt"] = {"dtype": dtype[a_prec], "value": round(world * args.steps / a_dt, 3), "unit": "steps/s",
                           "ms_per_step": round(a_dt / args.steps * 1e3, 4), "parity": "2e-4 vs reference fixtures "
                           "(tests/test_gpu_parity.py::test_denoiser_split_bf16_precision)" if a_prec == "bf16x3" else
                           "2e-5 vs reference fixtures", "roofline": roof(a_prec, a_kms, a_nev, a_dt)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(gd, B, L)
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment) and wait for them.  This process makes no HIP call before or after
    (a process that has initialised the GPU must not be replaced or forked on this pool); rank 0 prints the JSON
    line on the inherited stdout; any rank failing fails the run."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:                      # one rank died: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


def train_workload(args, mg, dev, dist, rank, world):
    """BASELINE configs[3]: multi-speaker naive training step, per-rank batch 8 (global 64 at N=8), L=1000.
    One step = D phase + G phase of train.py:91-184 on the hot path (synthetic conditioner standing in for the
    linguistic encoder), both optimizers, gradient all-reduce (mean) per optimizer before clipping."""
    from helpers import hot_path_configs, write_stats
    B, L = (8 if args.batch == B_PER_GPU else args.batch), args.frames
    with tempfile.TemporaryDirectory() as d:
        stats = write_stats(d, [-11.5] * MEL, [2.0] * MEL, n_speakers=218)
        a, pre, mc, tr = hot_path_configs("naive", 4, multi_speaker=True, stats_dir=stats)
        G = mg.GaussianDiffusion(a, pre, mc, tr)
        D = mg.JCUDiscriminator(pre, mc, tr)
    gen = torch.Generator().manual_seed(1234)          # identical initial weights on every rank
    with torch.no_grad():
        for p in list(G.parameters()) + list(D.parameters()):
            fan = p[0].numel() if p.dim() > 1 else 1
            p.copy_(torch.randn(p.shape, generator=gen) * (fan ** -0.5 if p.dim() > 1 else 0.1))
    G, D = G.to(dev), D.to(dev)
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    if dist is not None and world == 1:                 # MG_BENCH_EXCHANGE=1: run the collectives on the world of one
        trainer.bucketG.always_exchange = trainer.bucketD.always_exchange = True
    rng = np.random.default_rng(1234 + rank)            # rank-distinct data shard
    mel = torch.from_numpy(rng.uniform(-11.5, 2.0, (B, L, MEL)).astype(np.float32)).to(dev)
    cond = torch.from_numpy(rng.standard_normal((B, L, 256)).astype(np.float32)).to(dev)
    spk = torch.from_numpy(rng.standard_normal((B, 256)).astype(np.float32)).to(dev)
    pad = torch.zeros(B, L, dtype=torch.bool, device=dev)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(warm, steps):
        for _ in range(warm):
            trainer.step(mel, cond, spk, pad)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = trainer.step(mel, cond, spk, pad)
        sync()
        tm = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item()), o

    dt, out = timed(args.warmup, args.steps)
    assert all(torch.isfinite(v).all() for v in out.values())
    trainer.check()                                     # a hand-off timeout would have poisoned the gradients
    comm = {}
    if dist is not None:
        # the exposed cost of the gradient exchange: the same steps with the collectives skipped (everything else --
        # gather, side stream, event hand-off, the division by the world size -- kept).  Measured AFTER the headline
        # timing: without the all-reduce the replicas drift apart, which only this leg tolerates.
        trainer.bucketG.stub = trainer.bucketD.stub = True
        dt_stub, _ = timed(2, args.steps)
        trainer.bucketG.stub = trainer.bucketD.stub = False
        comm = {"comm_exposed_ms": round((dt - dt_stub) / args.steps * 1e3, 3),
                "ms_per_step_without_collectives": round(dt_stub / args.steps * 1e3, 3),
                "grad_bytes": {"G": trainer.bucketG.flat.numel() * 4, "D": trainer.bucketD.flat.numel() * 4,
                               "G_early_chunk": (trainer._early[1] - trainer._early[0]) * 4 if trainer._early else 0},
                "backend": dist.get_backend()}
    if rank == 0:
        # per step: 2 denoiser forwards + 1 backward (= 4 forward-equivalents), 4 D forwards + 2 backward passes
        flop = (4 * FLOP_PER_FRAME + 8 * 645504.0) * B * L * world
        print(json.dumps({
            "metric": "GAN train steps/sec (hot path: GaussianDiffusion + JCUDiscriminator, G+D optimizers)",
            "value": round(args.steps / dt, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: multi-speaker naive train step, batch %d/GPU (global %d), L=%d, "
                                   "T=4" % (B, B * world, L),
                       "parallelism": "dp%d, flat-bucket gradient all-reduce per optimizer (G %.0f MB, its k=3 slice "
                                      "behind an event inside the backward; D %.0f MB)"
                                      % (world, trainer.bucketG.flat.numel() * 4e-6, trainer.bucketD.flat.numel() * 4e-6)},
            "samples_per_s": round(B * world * args.steps / dt, 2),
            "approx_tflops": round(flop * args.steps / dt / 1e12, 1), **comm}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(gd, B, L):
    """The CPU oracle (oracle/refmath.py, the reference's op sequence incl. torch.stack of the 20
    skips) on the host cores: the same p_sample step, same B and L, bounded to ~20 s.  Its first T steps are the
    reverse chain x_T -> x_0, which also supplies the metric's second half ("mel L1 vs ref")."""
    from oracle import refmath as R
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # the box's CPU share (cgroup v2 quota), e.g. "1600000 100000" -> 16 cores
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    W = {k: v.detach().cpu() for k, v in gd.state_dict().items()}
    buf = {k: v.detach().cpu() for k, v in gd._buf().items()}
    dev = next(gd.parameters()).device
    T = gd.num_timesteps
    g = torch.Generator().manual_seed(0)
    x_T = torch.randn(B, 1, MEL, L, generator=g)
    cond = torch.randn(B, 256, L, generator=g)
    cond_d = cond.to(dev)
    noises = [torch.randn(B, 1, MEL, L, generator=g) for _ in range(T)]
    times = []
    t_start = time.perf_counter()

    def err_stats(ours, ref):
        """mean |err| (the metric's "mel L1 vs ref"), max-abs err / max-abs ref (what tests/helpers.py asserts), and
        the 99.9th percentile of the ELEMENT-WISE relative error |err| / max(|ref|, 1e-6)."""
        ours, ref = ours.double().flatten(), ref.double().flatten()
        d = (ours - ref).abs()
        rel = d / ref.abs().clamp_min(1e-6)
        kth = lambda q: float(rel.kthvalue(max(1, int(round(q * rel.numel())))).values)  # noqa: E731
        # element-wise figures are dominated by elements whose reference value is itself ~0 (a zero crossing of a
        # [-1, 1] signal with 3e-7 of absolute error): the share of elements above the 1e-3 budget says how many
        return {"mel_l1": float(d.mean()), "max_abs_over_max_ref": float(d.max() / ref.abs().max()),
                "p50_elementwise_rel": kth(0.5), "p99_elementwise_rel": kth(0.99), "p999_elementwise_rel": kth(0.999),
                "frac_elements_over_1e-3_rel": float((rel > 1e-3).double().mean())}

    # The timed CPU steps ARE the reference's T-step reverse chain (model/diffusion.py:155-165): x_T -> ... -> x_0
    # with pre-drawn noises.  Each step's pre-clamp Denoiser.forward output (the predicted x_0, :125) is compared
    # with the HIP Denoiser on the same x_t, and the final denormalised mel with the HIP chain run end to end
    # (errors compound over the T steps there).  The oracle is the checker here, not the product.
    x = x_T
    x0_stats, clamped = [], []
    for i in reversed(range(T)):
        t = torch.full((B,), i, dtype=torch.long)
        t0 = time.perf_counter()
        with torch.no_grad():
            x0_ref = R.denoiser_forward(W, "denoise_fn.", x, t, cond, None)
            x_next = R.q_posterior_sample(buf, x0_ref.clamp(-1.0, 1.0), x, t, noises[T - 1 - i])
        times.append(time.perf_counter() - t0)
        with torch.no_grad():
            x0_hip = gd.denoise_fn(x.to(dev), t.to(dev), cond_d, None).cpu()
        x0_stats.append(err_stats(x0_hip, x0_ref))
        clamped.append(float((x0_ref.abs() > 1.0).float().mean()))
        x = x_next
    mel_ref = R.denorm_spec(x[:, 0].transpose(1, 2), buf["spec_min"], buf["spec_max"])
    tape = iter(noises)
    saved, saved_cond, saved_spk = gd.noise_fn, gd.cond, gd.spk_emb
    gd.noise_fn = lambda shape: next(tape)
    gd.cond, gd.spk_emb = cond_d, None
    try:
        mel_hip = gd.sampling(noise=x_T.to(dev), keep_trace=False)[-1].cpu()
    finally:
        gd.noise_fn, gd.cond, gd.spk_emb = saved, saved_cond, saved_spk
    worst = lambda key: max(s_[key] for s_ in x0_stats)  # noqa: E731
    parity = {"what": "x0_pred = pre-clamp Denoiser.forward output at each of the T=%d steps of the reverse chain "
                      "(worst step; normalised mel units), and final_mel = denormalised mel after the whole chain "
                      "(log-mel units), HIP vs the CPU restatement on identical weights, x_T, cond and noises at "
                      "B=%d, L=%d" % (T, B, L),
              "x0_pred": {k: worst(k) for k in x0_stats[0]}, "final_mel": err_stats(mel_hip, mel_ref),
              "x0_clamped_frac": round(max(clamped), 4), "tolerance": 1e-3}
    assert parity["x0_pred"]["max_abs_over_max_ref"] <= 1e-3 and parity["final_mel"]["max_abs_over_max_ref"] <= 1e-3, parity
    # more timed steps (same shapes, t cycling) until ~20 s of CPU work
    i = 0
    while time.perf_counter() - t_start < 20 and len(times) < 16:
        t = torch.full((B,), T - 1 - (i % T), dtype=torch.long)
        t0 = time.perf_counter()
        R.p_sample(W, buf, x, t, cond, None, noises[i % T])
        times.append(time.perf_counter() - t0)
        i += 1
    med = float(np.median(times[1:])) if len(times) > 1 else times[0]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {"value": round(1.0 / med, 4), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "%d timed p_sample steps (median, 1 warm-up dropped) at the same B=%d, L=%d on %s"
                      % (max(1, len(times) - 1), B, L, model), "parity": parity}


if __name__ == "__main__":
    main()
