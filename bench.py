#!/usr/bin/env python3
"""bench.py — VQA train-step throughput on MI355X (BASELINE.json metric: VQA samples/sec).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + soft-target CE + backward + gradient all-reduce (N > 1) + Adam on one synthetic
batch that is already resident in HBM: BASELINE.json configs[1] (batch 256 per GPU, 224x224 images,
14-token questions, 1000-way answer head, fp32), train mode with all dropout sites active, random-init
weights of the reference architecture (config.yaml:51-74).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide, "Peak BF16/FP16 MFMA": ~2.5 PF dense (the 5 PF figure is 2:1 sparse)
X3_MFMA_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0   # fp32x3: six bf16 MFMAs per fp32-equivalent 32x32x16 block


def reference_cfg(answers: int) -> dict:
    """train: section of the reference's config/config.yaml:51-74 (max_answers per BASELINE.json)."""
    return {
        "text": {"question_features": 1024, "embedding_features": 300, "dropout": 0.3,
                 "num_lstm_layers": 1, "bidirectional": True},
        "image": {"kernel_size": 3, "dropout": 0.3, "num_channels": [3, 64, 128, 256], "stride": 1,
                  "do_skip_connection": False},
        "attention": {"hidden_dim": 1024, "glimpses": 2, "do_option": "+", "dropout": 0.3},
        "classifier": {"hidden_dim": 1024, "dropout": 0.3},
        "max_answers": answers,
    }


def synthetic_batch(B, S, T, V, A, seed, full_len=True, kmax=3):
    """7-tuple in the dataset layout (data_preprocessing.py:74-87): SURVEY.md §8d synthetic inputs."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, 3, S, S, generator=g)
    q_len = torch.full((B,), T, dtype=torch.int64) if full_len else torch.randint(1, T + 1, (B,), generator=g)
    q = torch.randint(1, V, (B, T), generator=g) * (torch.arange(T)[None, :] < q_len[:, None])
    a_len = torch.randint(1, kmax + 1, (B,), generator=g)
    a_idx = torch.zeros(B, kmax, dtype=torch.int64)
    a_val = torch.zeros(B, kmax, dtype=torch.int64)
    col = torch.arange(kmax)[None, :]
    a_idx = (torch.rand(B, A, generator=g).argsort(dim=1)[:, :kmax] + 1) * (col < a_len[:, None])
    a_val = torch.randint(1, 4, (B, kmax), generator=g) * (col < a_len[:, None])
    return v, q, a_idx, a_val, a_len, torch.arange(B), q_len


def conv_shapes(S, channels, stride=1):
    """[(Cin, Cout, Ho, Wo)] per conv block, for the algorithmic FLOP counts."""
    out, H = [], S
    for ci, co in zip(channels[:-1], channels[1:]):
        Ho = (H - 3) // stride + 1
        out.append((ci, co, Ho, Ho))
        H = Ho // 2
    return out, H


def step_flops_per_sample(cfg, S, T):
    """2*MACs of every contraction, forward + backward (wgrad + dgrad, no dgrad for conv0)."""
    ch = cfg["image"]["num_channels"]
    shapes, g = conv_shapes(S, ch, cfg["image"]["stride"])
    fwd = 0.0
    conv0 = 0.0
    for i, (ci, co, Ho, Wo) in enumerate(shapes):
        m = Ho * Wo * co * 9 * ci
        fwd += m
        if i == 0:
            conv0 = m
    P = g * g
    E, H = cfg["text"]["embedding_features"], cfg["text"]["question_features"]
    nd = 2 if cfg["text"]["bidirectional"] else 1
    mid, G, hid, A = cfg["attention"]["hidden_dim"], cfg["attention"]["glimpses"], cfg["classifier"]["hidden_dim"], cfg["max_answers"]
    C, Q = ch[-1], nd * H
    fwd += P * C * mid + P * mid * G + P * G * C                  # v_conv, x_conv, weighted sum
    fwd += nd * T * (E * 4 * H + H * 4 * H)                       # LSTM
    fwd += Q * mid + (G * C + Q) * hid + hid * A                  # q_lin, lin1, lin2
    return 2.0 * (3.0 * fwd - conv0)


def cpu_baseline(cfg, V, A, T, steps=1):
    """The CPU restatement (oracle, kind 'port') timed on this host: forward + loss + backward + Adam,
    batch 16 of 224x224 images (BASELINE.json configs[0]); one warm-up + `steps` timed steps, bounded to ~10-25 s, and run
    BEFORE the GPU leg so that the GPU part of the run is its contiguous tail (VERDICT r2 'weak' 11)."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd import VqaNet
    B, S = 16, 224
    torch.manual_seed(1)
    sd = {k: t.detach().clone() for k, t in VqaNet(cfg, V).state_dict().items()}
    m = {k: torch.zeros_like(t) for k, t in sd.items()}
    vv = {k: torch.zeros_like(t) for k, t in sd.items()}
    batch = synthetic_batch(B, S, T, V, A, seed=1)
    v, q, a_idx, a_val, _, _, q_len = batch
    times = []
    t_all = time.time()
    for it in range(steps + 1):
        t0 = time.time()
        _, loss, grads = O.loss_and_grads(sd, cfg, v, q, q_len, a_idx, a_val)
        lr = O.learning_rate(5e-4, it)
        for k in sd:
            O.adam_step(sd[k], grads[k], m[k], vv[k], it + 1, lr)
        times.append(time.time() - t0)
        if time.time() - t_all > 25:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": round(B / best, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} train steps (fwd+loss+bwd+Adam) of batch {B}, {S}x{S}, T={T}, A={A}; "
                      f"best of {max(1, len(times) - 1)} after 1 warm-up; os.cpu_count()={os.cpu_count()}; "
                      f"run before the GPU leg ({time.time() - t_all:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE.json configs[1])")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--tokens", type=int, default=14)
    ap.add_argument("--answers", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=5000)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp32x3"],
                    help="fp32 = the headline / parity path (BASELINE configs[1]); bf16 = the bf16 MFMA conv/FC path "
                         "(configs[3]: use with --batch 512 --size 448); never the headline; fp32x3 = fp32 tensors and fp32 "
                         "accuracy with the conv blocks' contractions on the bf16 matrix cores (exact 3 x bf16 operand split)")
    ap.add_argument("--stream-steps", type=int, default=3,
                    help="extra steps after the timed region in which the HBM-bound kernel families are bracketed "
                         "with HIP events (the `streaming` table); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-x3", action="store_true",
                    help="skip the extra fp32x3 measurement that a --dtype fp32 run appends as the `fp32x3` object")
    ap.add_argument("--eval-mode", action="store_true", help="diagnostic only: dropout off")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostic: take the RCCL data-parallel path even with one rank (under torchrun)")
    ap.add_argument("--roofline-kernel", default="auto",
                    help="kernel reported as `roofline`: auto = the kernel family with the most device time; or "
                         "conv_fwd|conv_dgrad|conv_wgrad[:layer]")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # rehearsal knobs (not for measurements): VQA_BENCH_DEVICE=0 puts every rank on one GPU and VQA_BENCH_BACKEND=gloo takes the
    # collective through the host, so that the world_size > 1 branches of this script can be exercised on a one-GPU box
    dev_index = int(os.environ.get("VQA_BENCH_DEVICE", local_rank))
    backend = os.environ.get("VQA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from dl_vqa_amd import VqaNet, _lib
    from dl_vqa_amd.distributed import DataParallel
    from dl_vqa_amd.train import FusedAdam, run_batch, update_learning_rate

    lib = _lib.load()
    assert lib.vqa_device_ok() == 1, "no gfx950 device visible"
    cfg = reference_cfg(args.answers)
    B, S, T, V, A = args.batch, args.size, args.tokens, args.vocab, args.answers
    # the CPU baseline runs FIRST (rank 0, N = 1 only): the GPU leg is then the contiguous tail of the run
    cpu_info = cpu_baseline(cfg, V, A, T, steps=2) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None
    torch.manual_seed(1)                                   # config.yaml:9 seed
    model = VqaNet(cfg, V, compute_dtype=args.dtype).to(dev)
    model.train(not args.eval_mode)
    dp = DataParallel(model) if use_dist else None
    batch = tuple(t.to(dev) for t in synthetic_batch(B, S, T, V, A, seed=1 + rank))
    opt = FusedAdam(model, lr=5e-4)
    it = [0]

    def step():
        # no explicit divisor: under DP run_batch takes local batch x world size from the attached synchroniser
        loss, score = run_batch(model, None, batch, A)
        opt.zero_grad()
        update_learning_rate(opt, it[0], 5e-4)
        loss.backward()
        opt.step()
        it[0] += 1
        return loss

    for _ in range(args.warmup):
        step()
    FAMS = {0: "gemm", 1: "conv_fwd", 2: "conv_dgrad", 3: "conv_wgrad"}
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    lib.vqa_prof_arm_mask(0b1111, -1)            # bracket the MFMA kernel families (GEMM, conv fwd / dgrad / wgrad) with HIP events
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    import ctypes
    cap = 64
    g_id, g_tag, g_n = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    g_ms = (ctypes.c_float * cap)()
    n_groups = lib.vqa_prof_read_groups(g_id, g_tag, g_n, g_ms, cap)
    lib.vqa_prof_arm(-1, -1)
    # HBM-bound stages: a few extra steps OUTSIDE the timed region with their families bracketed as well
    stream_rows = []
    if args.stream_steps > 0 and not use_dist:      # single GPU only: under DP every rank would have to step
        lib.vqa_prof_arm_mask(((1 << _lib.K_COUNT) - 1) & ~0b1111, -1)
        for _ in range(args.stream_steps):
            step()
        torch.cuda.synchronize()
        s_id, s_tag, s_n = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
        s_ms = (ctypes.c_float * cap)()
        ns = lib.vqa_prof_read_groups(s_id, s_tag, s_n, s_ms, cap)
        lib.vqa_prof_arm(-1, -1)
        stream_rows = [(s_id[k], s_tag[k], s_n[k], s_ms[k]) for k in range(ns)]
    dp_info = None
    if use_dist:
        mine = torch.tensor([elapsed], device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        elapsed = max(per_rank)
        dp_info = {"ms_per_step_per_rank_min": round(min(per_rank) / args.steps * 1e3, 3),
                   "ms_per_step_per_rank_max": round(max(per_rank) / args.steps * 1e3, 3),
                   "bucketed_direct_path": bool(model._last_backward_direct),
                   "collectives_per_backward": list(dp.issued[-4:]),
                   "streams_bwd": os.environ.get("VQA_STREAMS_BWD", os.environ.get("VQA_STREAMS", "2"))}

        def timed_dp(nsteps):
            """nsteps more steps with every bucket's issue point and the end of backward marked by HIP events."""
            dp.record_events = True
            torch.cuda.synchronize()
            dist.barrier()
            t_ = time.perf_counter()
            for _ in range(nsteps):
                step()
            torch.cuda.synchronize()
            dist.barrier()
            el = time.perf_counter() - t_
            dp.record_events = False
            rows = dp.timings()
            tm = torch.tensor([el], device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            avg = {k: round(sum(r[k] for r in rows) / len(rows), 3) for k in rows[0]} if rows else {}
            return float(tm.item()) / nsteps * 1e3, avg

        # overlap evidence OUTSIDE the timed region: ms from each bucket's issue point to the end of backward's kernels
        # (what its all-reduce can hide under) and ms the stream then still waits for the collectives
        ms_a, ov_a = timed_dp(max(3, args.steps // 2))
        dp_info["overlap_ms_bucket_issue_to_backward_end"] = {k: v for k, v in ov_a.items() if not k.startswith("_")}
        dp_info["exposed_allreduce_ms_after_backward"] = ov_a.get("_exposed_ms")
        # the other backward schedule (VQA_STREAMS_BWD: 1 = the BPTT chain joined before the conv backward, so the 'text'
        # bucket's all-reduce runs under all of the conv kernels; 2 = the chain under the conv kernels): measured beside
        # the headline schedule so that the first multi-GPU run can pick the default
        cur = dp_info["streams_bwd"]
        alt = "1" if cur != "1" else "2"
        os.environ["VQA_STREAMS_BWD"] = alt
        for _ in range(2):
            step()
        ms_b, ov_b = timed_dp(max(3, args.steps // 2))
        os.environ["VQA_STREAMS_BWD"] = cur
        dp_info["schedule_alt"] = {"streams_bwd": alt, "ms_per_step": round(ms_b, 3), "same_run_ms_per_step_of_default": round(ms_a, 3),
                                   "overlap_ms_bucket_issue_to_backward_end": {k: v for k, v in ov_b.items() if not k.startswith("_")},
                                   "exposed_allreduce_ms_after_backward": ov_b.get("_exposed_ms")}
    final_loss = float(loss.detach())

    # The same step in the fp32x3 mode (fp32 tensors and fp32-level accuracy, the conv blocks and v_conv contractions on the
    # bf16 matrix cores through exact 3 x bf16 operand splits), measured the same way and reported BESIDE the headline as
    # the `fp32x3` object: `value` stays the native fp32 MFMA path (BASELINE configs[1]).
    x3_info = None
    if args.dtype == "fp32" and not args.no_x3:
        try:
            torch.manual_seed(1)
            model_x = VqaNet(cfg, V, compute_dtype="fp32x3").to(dev)
            model_x.train(not args.eval_mode)
            if use_dist:
                DataParallel(model_x)
            opt_x = FusedAdam(model_x, lr=5e-4)
            itx = [0]

            def step_x():
                loss_x, _ = run_batch(model_x, None, batch, A)
                opt_x.zero_grad()
                update_learning_rate(opt_x, itx[0], 5e-4)
                loss_x.backward()
                opt_x.step()
                itx[0] += 1
                return loss_x

            for _ in range(args.warmup):
                step_x()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            lib.vqa_prof_arm_mask(0b1111, -1)        # the same event bracketing as the headline's timed loop
            torch.cuda.synchronize()
            tx0 = time.perf_counter()
            for _ in range(args.steps):
                loss_x = step_x()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            el_x = time.perf_counter() - tx0
            lib.vqa_prof_arm(-1, -1)
            if use_dist:
                tmax = torch.tensor([el_x], device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el_x = float(tmax.item())
            x3_info = {"value": round(B * world * args.steps / el_x, 2), "unit": "samples/s",
                       "ms_per_step": round(el_x / args.steps * 1e3, 3), "steps": args.steps, "warmup": args.warmup,
                       "final_loss": round(float(loss_x.detach()), 5),
                       # the step's fp32 FLOPs per second against the fp32 MFMA peak (what the native path is priced against)
                       "step_frac_of_fp32_mfma_peak": round(B * world * args.steps / el_x * step_flops_per_sample(cfg, S, T) / 1e12
                                                            / (FP32_MFMA_PEAK_TFLOPS * world), 4),
                       "dtype": "f32 (3xbf16 split on bf16 MFMA, fp32 accumulate)",
                       "note": "same step, batch and weights; conv blocks 1.. and the v_conv products on the bf16 matrix cores with "
                               "every fp32 operand split exactly into three bf16 terms (six partial products, fp32 accumulate): "
                               "fp32-level error against float64 (tests/test_x3_gpu.py), reference-fixture parity at the fp32 "
                               "tolerances (tests/test_model_gpu.py); opt-in (compute_dtype='fp32x3'), never `value`"}
            del model_x, opt_x
        except Exception as exc:      # the headline above must not depend on the extra measurement
            x3_info = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        shapes, _ = conv_shapes(S, cfg["image"]["num_channels"], cfg["image"]["stride"])
        # live per-kernel table: every convolution kernel of the step, algorithmic FLOPs / measured launch time
        peak = {"bf16": BF16_MFMA_PEAK_TFLOPS, "fp32x3": X3_MFMA_PEAK_TFLOPS}.get(args.dtype, FP32_MFMA_PEAK_TFLOPS)
        # HBM-side bytes per launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes, gfx950
        # correction per MI355X_MICROARCH.md): tools/pmc_conv_run.py launches every conv kernel of one layer in isolation
        # under rocprofv3, tools/pmc_traffic_summary.py writes profiles/r03_conv_traffic_<dtype>_<size>_<batch>.json keyed
        # "<family>:<layer>" with the kernel names it saw
        traffic_db, traffic_file = {}, f"r03_conv_traffic_{args.dtype}_{S}_{B}.json"
        try:
            with open(os.path.join(ROOT, "profiles", traffic_file)) as f:
                traffic_db = json.load(f)
        except (OSError, ValueError):
            traffic_file = None
        esz = 2 if args.dtype == "bf16" else 4           # bytes per activation element between the conv blocks

        def conv_alg_bytes(fam, tag):
            """Compulsory operand bytes of one launch (each tensor once): DESIGN.md 4.1."""
            ci, co, Ho, Wo = shapes[tag]
            Hi, Hp = Ho + 2, Ho // 2
            x_b = B * Hi * Hi * max(ci, 4) * (4 if tag == 0 else esz)
            pooled = B * Hp * Hp * co
            last = tag == len(shapes) - 1
            if fam == "conv_fwd":
                return x_b + pooled * ((4 if last else esz) + 1) + 9 * ci * co * esz
            if fam == "conv_dgrad":
                return pooled * (esz + 1) + x_b + 9 * ci * co * esz
            return x_b + pooled * (esz + 1) + 9 * ci * co * 4
        kernels = []
        for g in range(n_groups):
            fam, tag = FAMS.get(g_id[g]), g_tag[g]
            if fam in (None, "gemm") or not (0 <= tag < len(shapes)) or g_n[g] == 0:
                continue
            ci, co, Ho, Wo = shapes[tag]
            flops_launch = 2.0 * B * Ho * Wo * co * 9 * ci
            avg_ms = g_ms[g] / g_n[g]
            ach = flops_launch / (avg_ms * 1e-3) / 1e12
            kpeak = peak if (args.dtype != "fp32" and tag > 0) else FP32_MFMA_PEAK_TFLOPS
            kernels.append({"kernel": f"{fam}[conv{tag}]", "launches": g_n[g], "avg_launch_ms": round(avg_ms, 4),
                            "algorithmic_gflop_per_launch": round(flops_launch / 1e9, 2),
                            "achieved": round(ach, 2), "frac": round(ach / kpeak, 4), "peak": kpeak,
                            "traffic": (int(traffic_db[f"{fam}:{tag}"]["hbm_bytes_corrected"])
                                        if f"{fam}:{tag}" in traffic_db else None),
                            "algorithmic_bytes_per_launch": int(conv_alg_bytes(fam, tag)),
                            "ms_per_step": round(g_ms[g] / args.steps, 3), "family": fam})
        # dominant kernel = the kernel FAMILY (conv_fwd / conv_dgrad / conv_wgrad, conv1 + conv2 launches together)
        # with the most device time in the step; --roofline-kernel conv_fwd|conv_dgrad|conv_wgrad[:layer] overrides.
        # (rocprofv3 --stats groups by kernel NAME instead, where the two forward launches share one name and rank
        # first; the family choice is the more conservative fraction.)
        roofline = None
        by_fam = {}
        for k in kernels:
            by_fam.setdefault(k["family"], []).append(k)
        pick = args.roofline_kernel
        if pick == "auto" and by_fam:
            pick = max(by_fam, key=lambda f: sum(k["ms_per_step"] for k in by_fam[f] if k["kernel"] != f + "[conv0]"))
        sel = [k for k in kernels if k["family"] == pick.split(":")[0] and k["kernel"] != k["family"] + "[conv0]"
               and (":" not in pick or k["kernel"].endswith(f"[conv{pick.split(':')[1]}]"))]
        if sel:
            n = sum(k["launches"] for k in sel)
            tot_ms = sum(k["avg_launch_ms"] * k["launches"] for k in sel)
            gf = sum(k["algorithmic_gflop_per_launch"] * k["launches"] for k in sel)
            ach = gf / tot_ms                     # GFLOP / ms = TFLOP/s
            tr = [k["traffic"] for k in sel]
            roofline = {"bound": "mfma", "kernel": "+".join(k["kernel"] for k in sel), "achieved": round(ach, 2),
                        "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": (int(sum(t * k["launches"] for t, k in zip(tr, sel)) / n) if all(t is not None for t in tr) else None),
                        "avg_launch_ms": round(tot_ms / n, 4), "launches": n,
                        "algorithmic_gflop_per_launch": round(gf / n, 2),
                        "algorithmic_bytes_per_launch": int(sum(k["algorithmic_bytes_per_launch"] * k["launches"] for k in sel) / n),
                        "traffic_unit": f"bytes/launch (PMC FETCH_SIZE x 2 + WRITE_SIZE, profiles/{traffic_file})" if traffic_file
                                        else "no PMC file for this shape / dtype under profiles/"}
        for k in kernels:
            del k["family"]
        # HBM-bound stages: algorithmic bytes per launch / measured launch time, against the 8 TB/s
        # HBM3E peak (6.3 TB/s is what a plain copy achieves on this chip)
        conv_out, g_ = conv_shapes(S, cfg["image"]["num_channels"], cfg["image"]["stride"])
        Pn, Cc, mid_, G_ = g_ * g_, cfg["image"]["num_channels"][-1], cfg["attention"]["hidden_dim"], cfg["attention"]["glimpses"]
        Mrows = B * Pn
        n_params = sum(p.numel() for p in model.parameters())
        alg = {
            (_lib.K_L2NORM_FWD, -1): ("l2norm_fwd", 2 * Mrows * Cc * 4),
            (_lib.K_L2NORM_BWD, -1): ("l2norm_bwd", 3 * Mrows * Cc * 4),
            (_lib.K_ATT_SCORE_FWD, -1): ("att_score_fwd", Mrows * mid_ * 4 + Mrows * G_ * 4),
            (_lib.K_ATT_SCORE_BWD, -1): ("att_score_bwd", 2 * Mrows * mid_ * 4 + Mrows * G_ * 4),
            (_lib.K_ATT_APPLY_FWD, -1): ("att_apply_fwd", Mrows * Cc * 4 + 2 * Mrows * G_ * 4),
            (_lib.K_ATT_APPLY_BWD, -1): ("att_apply_bwd", 2 * Mrows * Cc * 4 + 3 * Mrows * G_ * 4),
            (_lib.K_DROPOUT, 1): ("dropout[v]", (2 * Mrows * Cc * 4) if args.dtype == "fp32" else None),
            (_lib.K_ADAM, -1): ("adam", 7 * n_params * 4),
            (_lib.K_SOFTCE, -1): ("softce_fwd_bwd", 2 * B * A * 4),
        }
        streaming = []
        for fid, tag, n_l, ms in stream_rows:
            if fid == _lib.K_LSTM_SEQ and n_l:
                streaming.append({"kernel": "lstm_seq_" + ("fwd" if tag == 0 else "bwd"), "launches": n_l,
                                  "avg_call_ms": round(ms / n_l, 4), "note": "whole recurrence, both directions"})
                continue
            name, nbytes = alg.get((fid, tag), (None, None))
            if name is None or not nbytes or not n_l:
                continue
            avg_ms = ms / n_l
            gbs = nbytes / (avg_ms * 1e-3) / 1e9
            streaming.append({"kernel": name, "launches": n_l, "avg_launch_ms": round(avg_ms, 4),
                              "algorithmic_bytes_per_launch": int(nbytes), "achieved_GBps": round(gbs, 1),
                              "frac_of_8000": round(gbs / 8000.0, 4), "frac_of_6300_achievable": round(gbs / 6300.0, 4)})
        gflop_sample = step_flops_per_sample(cfg, S, T) / 1e9
        out = {
            "metric": "VQA samples/sec (train step)", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16", "fp32x3": "f32 (3xbf16 split on bf16 MFMA, fp32 accumulate)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": f"VqaNet train step (fwd+softCE+bwd+{'allreduce+' if use_dist else ''}Adam), "
                                   f"batch {B}/GPU, {S}x{S} images, {T}-token questions, {A}-way head, "
                                   f"{'eval' if args.eval_mode else 'train'} mode"
                                   + {"fp32": "", "bf16": ", bf16 MFMA conv blocks + v_conv (fp32 accumulate), fp32 LSTM / reductions / Adam",
                                      "fp32x3": ", conv blocks 1.. as exact 3 x bf16 splits on the bf16 MFMA (fp32 tensors, fp32 accuracy); "
                                                "peak = 2500 / 6 TFLOP/s fp32-equivalent"}[args.dtype],
                       "global_batch": B * world, "image_size": S, "tokens": T, "answers": A, "vocab": V,
                       "parallelism": f"dp{world}"},
            "step_gflop_per_sample": round(gflop_sample, 3),
            "step_mfma_frac": round(value * gflop_sample / 1e3 / (peak * world), 4),
            "step_mfma_peak_tflops": peak,
            "final_loss": round(final_loss, 5),
            "roofline": roofline,
            "kernels": kernels,
            "streaming": streaming,
        }
        if x3_info is not None:
            out["fp32x3"] = x3_info
        if dp_info is not None:
            out["data_parallel"] = dp_info
        if cpu_info is not None:
            out["cpu_baseline"] = cpu_info
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
