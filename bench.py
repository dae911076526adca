#!/usr/bin/env python3
"""bench.py -- molecules/s of the MM-DTI dual-encoder contrastive fine-tune step on MI355X.

Workload (BASELINE.json north_star / configs[1]): BBBP-like classification + SupCon (CT_Single) + InfoNCE, bf16 MFMA
compute, 256 molecules per GPU, every molecule at the worst case 128 atoms (N = 130 with BOS/EOS) and 256 SMILES
tokens, synthetic data, random-init weights of the reference architecture (Uni-Mol 15L/512/64 heads; ChemBERTa ASSUMED
6L/512/8 heads/FFN 2048/vocab 600 -- SURVEY.md 8d; cross-modal 1L x2/16 heads; InfoNCE 512-512-50; head 512-512-2).
One step = zero-grad + forward + backward (+ gradient all-reduce over RCCL when N > 1) + clip + Adam, dropout ON at
the reference's probabilities.  Weak scaling: 256 molecules per rank; InfoNCE negatives are global.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, timed live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))

import torch


def build_model(task="classification", output_dim=2):
    from mmdti_hip.models import mm_model as mm
    rcfg = SimpleNamespace(layers=6, dim=512, heads=8, ffn=2048, vocab=600, max_pos=514, type_vocab=1, pad_idx=1, ln_eps=1e-12,
                           hidden_dropout=0.1, attn_dropout=0.1)
    torch.manual_seed(1234)
    model = mm.MM_Model.from_configs(output_dim, task, roberta_cfg=rcfg)
    return model, rcfg


def synth(B, atoms, tokens, seed, task="classification", ragged=False):
    from oracle import mmdti_oracle as O          # synthetic-batch generator only (seeded numpy); no oracle compute here
    cfg = O.ModelCfg(task=task, output_dim=2 if task == "classification" else 1)
    batch, label = O.synth_batch(B, atoms, tokens, cfg, seed=seed, ragged=ragged)
    return cfg, batch, label


def cpu_baseline(atoms, tokens, sample_B=8, iters=2):
    """The CPU oracle (fp32 PyTorch restatement, kind='port') fwd+bwd on a bounded sample of the same workload."""
    from oracle import mmdti_oracle as O
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    cfg = O.ModelCfg(task="classification", output_dim=2)
    P = {k: v.requires_grad_() for k, v in O.init_params(cfg, seed=1).items()}
    batch, label = O.synth_batch(sample_B, atoms, tokens, cfg, seed=99, ragged=False)

    def one():
        for p in P.values():
            p.grad = None
        out = O.mm_forward(batch, P, cfg, net_target=label, training=True)
        loss, _ = O.step_loss(out, label, cfg.task)
        loss.backward()

    one()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters
    return {"value": sample_B / dt, "unit": "molecules/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} timed fwd+bwd steps (after 1 warm-up) of the fp32 PyTorch-CPU oracle on {sample_B} molecules x {atoms} atoms x {tokens} "
                      f"tokens, same architecture, dropout on; {dt:.2f} s/step"}


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of `kernel_prefix` from the committed PMC summary (profiles/r01_bench_pmc_hbm_traffic.csv:
    separate FETCH_SIZE / WRITE_SIZE rocprofv3 passes of this same command, gfx950 units already applied there).  PMC
    counters cannot be collected from inside the timed run, so this is the recorded value or null."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_hbm_traffic.csv")
    try:
        import csv
        with open(path) as f:
            for row in csv.DictReader(line for line in f if not line.startswith("#")):
                if kernel_prefix in row["kernel"]:
                    return int(float(row["hbm_bytes_per_launch"]))
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU")
    ap.add_argument("--atoms", type=int, default=128)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--ragged", action="store_true", help="molecules of mixed length padded to the batch maximum (not the headline workload)")
    args = ap.parse_args()

    from mmdti_hip import parallel, ops
    from mmdti_hip.trainer import FineTuner
    rank, local, world = parallel.init_from_env(force=os.environ.get("MMDTI_FORCE_DDP") == "1")
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run for N>1", file=sys.stderr)
        if args.gpus > 1 and world == 1:
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model, rcfg = build_model()
    model = model.to(dev).train()
    if os.environ.get("MMDTI_NO_OVERLAP") == "1":      # A/B switch: run the two towers back to back on one stream
        model.overlap_towers = False
    if os.environ.get("MMDTI_SPLIT_TOWER1"):           # A/B switch: tower 1 as n sub-batches on n streams
        model.split_tower1 = int(os.environ["MMDTI_SPLIT_TOWER1"])
    tuner = FineTuner(model, "classification", total_steps=10_000, distributed=(world > 1 or os.environ.get("MMDTI_FORCE_DDP") == "1"))
    _, batch, label = synth(args.batch, args.atoms, args.tokens, seed=1234 + rank, ragged=args.ragged)
    batch = {k: v.to(dev) for k, v in batch.items()}
    label = label.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = tuner.step(batch, label, epoch=0)
    barrier()
    ops.kernel_timer.enable(("pair_attn_bwd", "pair_attn_fwd", "gbf_features_fwd"))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tuner.step(batch, label, epoch=0)
    barrier()
    dt = time.perf_counter() - t0
    ops.kernel_timer.disable()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    losses = {"loss": float(out.loss), "task": float(out.task_loss), "infonce": float(out.infonce_loss), "ct": float(out.ct_loss)}
    # MFMA side of the roofline: every GEMM launch of two EXTRA steps (outside the timed region: ~600 event pairs per
    # step would perturb `value`) timed with HIP events on its launch stream; algorithmic flops 2*M*N*K per launch.
    # (every rank runs them: the step holds collectives)
    # Twice: towers overlapped as in the timed region (a launch's event time then includes sharing the chip with the
    # other tower's kernels), and towers back to back on one stream (each launch alone on the chip).
    pa_timers = ops.kernel_timer.summary()
    gemm_timers = {}
    from mmdti_hip import functional as Fn
    was = (model.overlap_towers, model.infonce_on_side_stream, model.cross_modal_module.two_streams, Fn.DEFER_WGRAD_LAYERS)
    for mode, ov in (("overlapped", True), ("alone", False)):
        if not ov:      # every stream-level overlap off: each launch has the chip to itself
            model.overlap_towers, model.infonce_on_side_stream, model.cross_modal_module.two_streams, Fn.DEFER_WGRAD_LAYERS = False, False, False, 0
        ops.kernel_timer.enable(("gemm",))
        for _ in range(2):
            tuner.step(batch, label, epoch=0)
        gemm_timers[mode] = ops.kernel_timer.summary().get("gemm")
        ops.kernel_timer.disable()
    model.overlap_towers, model.infonce_on_side_stream, model.cross_modal_module.two_streams, Fn.DEFER_WGRAD_LAYERS = was
    barrier()

    if rank == 0:
        N = int(batch["src_tokens"].shape[1])
        H = 64
        timers = pa_timers
        # dominant HBM-bound kernel: pair attention backward.  Algorithmic bytes per launch (DESIGN.md "roofline"):
        # per atom pair and head: read S (4 B) + read G (4 B) + write G (4 B) = 12 B  -> 768 B per pair over 64 heads,
        # plus q|k|v|dO|dqkv rows (7 x 16 B per (token, head)).
        pairs = args.batch * N * N
        pa_bytes = pairs * H * 12 + args.batch * N * H * 7 * 16
        ms = timers.get("pair_attn_bwd", {}).get("mean_ms")
        roofline = None
        if ms:
            ach = pa_bytes / (ms * 1e-3) / 1e9
            roofline = {"kernel": "pair_attn_bwd_mfma_kernel<9, true, true, 3>", "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(ach / 8000.0, 4), "traffic": pmc_traffic("pair_attn_bwd_mfma_kernel"), "algorithmic_bytes_per_launch": pa_bytes,
                        "mean_launch_ms": round(ms, 4), "launches_timed": timers["pair_attn_bwd"]["n"], "other_kernels_ms": {k: round(v["mean_ms"], 4) for k, v in timers.items()}}
        roofline_mfma = None
        if gemm_timers.get("alone"):
            ga, go = gemm_timers["alone"], gemm_timers["overlapped"]
            tf = ga["work"] / (ga["total_ms"] * 1e-3) / 1e12
            roofline_mfma = {"kernel": "gemm_glds_kernel / gemm_glds_tall_kernel / gemm_bf16_kernel (all GEMM launches of a step)", "bound": "mfma",
                             "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "traffic": None,
                             "launches_per_step": ga["n"] // 2, "gemm_ms_per_step": round(ga["total_ms"] / 2, 3),
                             "algorithmic_tflop_per_step": round(ga["work"] / 2 / 1e12, 3),
                             "achieved_with_towers_overlapped": round(go["work"] / (go["total_ms"] * 1e-3) / 1e12, 1),
                             "note": "HIP events around every GEMM launch of 2 extra steps after the timed region, towers back to back on one stream "
                                     "(each launch alone on the chip); achieved_with_towers_overlapped: same with the two towers sharing the chip"}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.atoms, args.tokens, args.cpu_sample)
        mols = args.batch * world * args.steps
        line = {
            "metric": "molecules/sec fwd+bwd (InfoNCE fine-tune)", "value": round(mols / dt, 2), "unit": "molecules/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BBBP-like classification + SupCon + InfoNCE fine-tune step, {args.batch} molecules/GPU x {args.atoms} atoms x "
                                   f"{args.tokens} SMILES tokens (all at max length), fwd+bwd+allreduce+clip+Adam, dropout on",
                       "global_batch": args.batch * world, "atoms": args.atoms, "tokens": args.tokens, "parallelism": f"dp{world}",
                       "unimol": "15L/512/64h", "chemberta_assumed": "6L/512/8h/ffn2048/vocab600", "infonce_negatives": "global",
                       "grad_buckets_reduced_during_backward": None if tuner.reducer is None or not tuner.reducer.active
                       else f"{tuner.reducer.overlapped}/{len(tuner.reducer.buckets)}"},
            "losses_last_step": losses, "roofline": roofline, "roofline_mfma": roofline_mfma, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
