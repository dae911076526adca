#!/usr/bin/env python3
"""bench.py -- molecules/s of the MM-DTI dual-encoder contrastive fine-tune step on MI355X.

Workload (BASELINE.json north_star / configs[1]): BBBP-like classification + SupCon (CT_Single) + InfoNCE, 16-bit MFMA compute
(fp16 forward operands -- the mode whose embeddings are within the north star's 1e-3 of the fp32 reference --, bf16 backward
operands, fp32 accumulation; MMDTI_FWD_FP16=0: bf16 everywhere), 256 molecules per GPU, every molecule at the worst case 128 atoms (N = 130 with BOS/EOS) and 256 SMILES
tokens, synthetic data, random-init weights of the reference architecture (Uni-Mol 15L/512/64 heads; ChemBERTa ASSUMED
6L/512/8 heads/FFN 2048/vocab 600 -- SURVEY.md 8d; cross-modal 1L x2/16 heads; InfoNCE 512-512-50; head 512-512-2).
One step = zero-grad + forward + backward (+ gradient all-reduce over RCCL when N > 1) + clip + Adam, dropout ON at
the reference's probabilities.  Weak scaling: 256 molecules per rank; InfoNCE negatives are global.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python bench.py --gpus N ...        (no WORLD_SIZE in the environment: starts the N ranks itself, before touching the GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement):
  roofline      the kernel family that takes the most GPU time of the step (the GEMMs: bound "mfma"), timed live with HIP
                events on the launch streams;
  rooflines     the same for every hot kernel family (pair attention fwd/bwd, the fused pair-distance kernel fwd/bwd that
                the north star names for the HBM figure, fused attention fwd/bwd, LayerNorm fwd/bwd, GEMMs), each with its
                algorithmic bytes or flops per launch, mean launch time and fraction of the 8 TB/s / 2.5 PF peak;
  cpu_baseline  the fp32 CPU oracle (kind "port") on a bounded sample of the same workload, median of 3 steps.
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))

import torch

HBM_PEAK_GBS, MFMA_PEAK_TFS = 8000.0, 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md


def build_model(task="classification", output_dim=2):
    from mmdti_hip.models import mm_model as mm
    rcfg = SimpleNamespace(layers=6, dim=512, heads=8, ffn=2048, vocab=600, max_pos=514, type_vocab=1, pad_idx=1, ln_eps=1e-12,
                           hidden_dropout=0.1, attn_dropout=0.1)
    torch.manual_seed(1234)
    model = mm.MM_Model.from_configs(output_dim, task, roberta_cfg=rcfg)
    return model, rcfg


def synth(B, atoms, tokens, seed, task="classification", ragged=False):
    """Seeded synthetic collated batch (mmdti_hip.synth: the reference's layout, host-side numpy)."""
    from mmdti_hip.synth import synth_batch
    batch, label = synth_batch(B, atoms, tokens, task=task, seed=seed, ragged=ragged)
    return None, batch, label


def cpu_baseline(atoms, tokens, sample_B=32, iters=3, scaling_B=8):
    """The CPU oracle (fp32 PyTorch restatement, kind='port') fwd+bwd on a bounded sample of the same workload: median of
    `iters` timed steps after one warm-up at sample_B molecules, plus one timed step at scaling_B to show how the rate
    depends on the sample size (small GEMMs under-use many cores)."""
    from oracle import mmdti_oracle as O
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    cfg = O.ModelCfg(task="classification", output_dim=2)
    P = {k: v.requires_grad_() for k, v in O.init_params(cfg, seed=1).items()}

    def timed(B, n):
        batch, label = O.synth_batch(B, atoms, tokens, cfg, seed=99, ragged=False)

        def one():
            for p in P.values():
                p.grad = None
            out = O.mm_forward(batch, P, cfg, net_target=label, training=True)
            loss, _ = O.step_loss(out, label, cfg.task)
            loss.backward()

        one()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            one()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    small = timed(scaling_B, 1) if scaling_B and scaling_B < sample_B else None
    dt = timed(sample_B, iters)
    out = {"value": sample_B / dt, "unit": "molecules/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"median of {iters} fwd+bwd steps (after 1 warm-up) of the fp32 PyTorch-CPU oracle on {sample_B} molecules x {atoms} atoms x "
                     f"{tokens} tokens, same architecture, dropout on; {dt:.2f} s/step"}
    if small:
        out["scaling"] = {f"B{scaling_B}_molecules_per_s": round(scaling_B / small, 3), f"B{sample_B}_molecules_per_s": round(sample_B / dt, 3)}
    return out


def pmc_traffic_family(kernel_prefix):
    """launch-weighted mean HBM bytes per launch over EVERY kernel whose name starts with `kernel_prefix` (the GEMM family: a dozen
    kernels and shapes), from the same PMC summary as pmc_traffic (whose passes run the headline step only)."""
    import csv
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_hbm_traffic.csv")), reverse=True):
        try:
            n = b = 0.0
            with open(path) as f:
                for row in csv.DictReader(line for line in f if not line.startswith("#")):
                    if row["kernel"].replace("mmdti::", "").startswith(kernel_prefix):
                        n += float(row["launches"]); b += float(row["launches"]) * float(row["hbm_bytes_per_launch"])
            if n:
                return int(b / n), os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of `kernel_prefix` from the NEWEST committed PMC summary (profiles/r*_bench_pmc_hbm_traffic.csv:
    separate FETCH_SIZE / WRITE_SIZE rocprofv3 passes of this command, gfx950 corrections applied by profiles/summarize.py).
    PMC counters cannot be collected from inside the timed run, so this is the recorded value, or null."""
    import csv
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_hbm_traffic.csv")), reverse=True):
        try:
            with open(path) as f:
                for row in csv.DictReader(line for line in f if not line.startswith("#")):
                    if row["kernel"].replace("mmdti::", "").startswith(kernel_prefix):
                        return int(float(row["hbm_bytes_per_launch"])), os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


# kernel families timed live: name -> (device kernel-name prefix for the PMC lookup, bound)
FAMILIES = {
    "pair_attn_fwd": ("pair_attn_fwd_mfma_kernel", "hbm"),
    "pair_attn_bwd": ("pair_attn_bwd_mfma_kernel", "hbm"),
    "gbf_bias_fwd": ("gbf_bias_fwd_kernel", "hbm"),        # "the pair-distance kernel" of the north star
    # (the complete backward kernel recomputes instead of re-reading: its work unit is MFMA flops, see ops.gbf_bias_bwd_full;
    #  with MMDTI_GBF_FULL_BWD=0 the round-1 per-pair kernel runs and the unit is bytes again)
    "gbf_bias_bwd": ("gbf_bias_bwd_full_kernel", "mfma") if os.environ.get("MMDTI_GBF_FULL_BWD", "1") != "0" else ("gbf_bias_bwd_kernel", "hbm"),
    "ln_fwd": ("ln_fwd_kernel", "hbm"),
    "ln_bwd": ("ln_bwd_kernel", "hbm"),
    "attn_fwd": ("attn_fwd_kernel", "mfma"),
    "attn_bwd": ("attn_bwd_q_kernel", "mfma"),
    "gemm": ("gemm_", "mfma"),
}


def family_rooflines(summary, steps):
    rows = []
    for name, (prefix, bound) in FAMILIES.items():
        d = summary.get(name)
        if not d or not d["total_ms"]:
            continue
        per_launch = d["work"] / d["n"]
        rate = d["work"] / (d["total_ms"] * 1e-3)
        peak = HBM_PEAK_GBS if bound == "hbm" else MFMA_PEAK_TFS
        ach = rate / 1e9 if bound == "hbm" else rate / 1e12
        traffic, src = pmc_traffic(prefix) if name != "gemm" else pmc_traffic_family(prefix)   # (one family, many kernels: launch-weighted mean)
        rows.append({"kernel": name, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                     "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": src,
                     ("algorithmic_bytes_per_launch" if bound == "hbm" else "algorithmic_flop_per_launch"): int(per_launch),
                     "mean_launch_ms": round(d["mean_ms"], 4), "launches_per_step": round(d["n"] / steps, 1), "ms_per_step": round(d["total_ms"] / steps, 3)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (this process has not
    touched the GPU -- nothing before this point initialises HIP -- and never does), relay rank 0's JSON line, exit with the
    child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n}-rank run failed (exit code {proc.returncode}, JSON line {'missing' if line is None else 'present'})", file=sys.stderr)
        sys.exit(proc.returncode or 1)
    print(line)
    sys.exit(0)


def parity_record():
    """the newest committed parity measurement of the default precision mode against the reference's own fp32 run (written by
    tests/test_g9_gpu.py on the GPU box, copied to profiles/): embeddings, logits, losses"""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_default_mode.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            d["source"] = os.path.basename(path)
            return d
        except (OSError, ValueError):
            continue
    return None


def pmc_step_bytes():
    """HBM bytes of ONE headline step from the newest committed PMC summary: sum over its kernels of launches x bytes per launch,
    divided by the steps of that trace (header comment `steps_in_trace=`; 2 = the warm-up + the step of the PMC command)."""
    import csv
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_hbm_traffic.csv")), reverse=True):
        try:
            txt = open(path).read()
            m = re.search(r"steps_in_trace=(\d+)", txt)
            steps = int(m.group(1)) if m else 2
            tot = gem = 0.0
            for row in csv.DictReader(line for line in txt.splitlines() if not line.startswith("#")):
                b = float(row["launches"]) * float(row["hbm_bytes_per_launch"])
                tot += b
                if row["kernel"].replace("mmdti::", "").startswith(("gemm_", "grouped_reduce", "splitk_reduce")):
                    gem += b
            return tot / steps, gem / steps, os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


class PipelineBatches(torch.utils.data.Dataset):
    """item i = mixed-length batch i % nb of `B` molecules, built, right-padded and narrowed (collate.device_payload) in a loader WORKER
    process -- collate.HostCollate's job in tasks.Trainer(num_workers=k).  The per-molecule samples are regenerated from the seed in each
    worker (spawned processes: nothing of the parent's GPU state is inherited)."""

    def __init__(self, n, nb, B, atoms, tokens, seed):
        self.n, self.nb, self.B, self.atoms, self.tokens, self.seed = n, nb, B, atoms, tokens, seed
        self.samples = None

    def __len__(self):
        return self.n

    def _make(self):
        import numpy as np
        from mmdti_hip.synth import molecule
        rng = np.random.default_rng(self.seed)
        vocab, elem_p = 31, np.zeros(31)
        elem_p[8], elem_p[4], elem_p[5], elem_p[6] = 0.5, 0.3, 0.075, 0.075
        rest = [i for i in range(4, 30) if i not in (4, 5, 6, 8)]
        elem_p[rest] = 0.05 / len(rest)
        out = []
        for _ in range(self.nb):
            mols = []
            for _ in range(self.B):
                na = int(np.clip(round(rng.normal(0.375 * self.atoms, 0.16 * self.atoms)), max(2, self.atoms // 16), self.atoms))
                nt = int(np.clip(round(0.8 * na), 8, self.tokens))
                t, d, e = molecule(rng, na, vocab, elem_p)
                ids = np.concatenate([[0], rng.integers(4, 600, size=nt - 2), [2]]).astype(np.int64)
                mols.append((torch.from_numpy(t), torch.from_numpy(d), torch.from_numpy(e), torch.from_numpy(ids)))
            out.append((mols, torch.from_numpy((rng.random((self.B, 1)) < 0.2).astype(np.int64))))
        return out

    def __getitem__(self, i):
        from mmdti_hip.collate import device_payload, right_pad
        if self.samples is None:
            self.samples = self._make()
        mols, y = self.samples[i % self.nb]
        t0 = time.perf_counter()
        ids = right_pad([m[3] for m in mols], 1)
        batch = {"src_tokens": right_pad([m[0] for m in mols], 0), "src_distance": right_pad([m[1] for m in mols], 0.0, square=True),
                 "src_edge_type": right_pad([m[2] for m in mols], 0, square=True), "input_ids": ids, "attention_mask": ids.ne(1).long()}
        batch = device_payload(batch, 961, 0)
        return batch, y, time.perf_counter() - t0


def pipeline_workload(tuner, model, args, dev, world, rank, barrier):
    """SURVEY 8d / VERDICT r03: the step fed a FRESH mixed-length batch every iteration, as the reference's loop is
    (tasks/trainer.py:177-283; its loaders collate in the main process, :551-555): per-molecule samples -> right-padded host batch
    (collate.py) -> device_payload (int16 edge types, packing facts), both in loader worker processes -> pinned staging + H2D on the copy
    stream (DevicePrefetcher) -> step.  8 distinct batches cycled; the host-side key-tile / tile-prefix arithmetic and their uploads
    happen INSIDE the timed region every step (the resident-batch workloads reuse them from a cache)."""
    from mmdti_hip.data import DevicePrefetcher
    B, nb = min(args.batch, 256), 8
    t_collate = []

    n_steps = max(3 * nb, args.steps)
    # ONE loader for the warm-up pass and the timed pass (persistent workers: process start-up, imports and the synthetic molecules are
    # paid once, before the clock starts).  pin_memory=True as tasks.Trainer builds its loaders: the loader's pin thread moves a batch out
    # of the worker's shared-memory segment -- 26 MB of first-touch page faults, 30 ms when the step loop does it itself (measured:
    # 49.9 ms per step with pin_memory=False against 29.8 ms, the GPU time of the step) -- and DevicePrefetcher copies from the pinned
    # tensors where they lie.
    dl = torch.utils.data.DataLoader(PipelineBatches(n_steps, nb, B, args.atoms, args.tokens, 777 + rank), batch_size=None, shuffle=False, num_workers=4,
                                     pin_memory=True, prefetch_factor=2, multiprocessing_context="spawn", persistent_workers=True)

    def collated(_n):
        for batch, y, dt_c in dl:
            t_collate.append(float(dt_c))
            yield batch, y

    def run(n_steps, timed):
        h2d, opt = [], []
        pf = DevicePrefetcher(collated(n_steps), dev, narrow=False)       # (already narrowed by the loader's workers)
        real_enqueue, real_opt = pf._enqueue, tuner.optimizer_step

        def launch(slot, staged, label):
            if not timed:
                return real_enqueue(slot, staged, label)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(pf.stream)
            out = real_enqueue(slot, staged, label)
            e1.record(pf.stream)
            h2d.append((e0, e1))
            return out

        def opt_step():
            if not timed:
                return real_opt()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            real_opt()
            e1.record()
            opt.append((e0, e1))

        pf._enqueue, tuner.optimizer_step = launch, opt_step
        try:
            t0 = None
            n = 0
            for net_input, y in pf:
                if t0 is None:              # (the clock starts at the first batch: worker start-up -- process spawn, imports -- is not the steady state)
                    barrier()
                    t0 = time.perf_counter()
                tuner.step(net_input, y, epoch=0)
                n += 1
            barrier()
            dt = time.perf_counter() - t0
        finally:
            tuner.optimizer_step = real_opt
        return h2d, opt, dt, n

    run(n_steps, False)                  # warm-up pass: every batch shape (allocator, pinned buffers), the workers up and running
    t_collate.clear()
    h2d, opt, dt, n = run(n_steps, True)
    del dl
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    ms = lambda evs: round(sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs)), 3)
    return {"workload": f"same step fed a fresh batch every iteration: {nb} distinct mixed-length batches of {B} molecules/GPU cycled through per-molecule samples -> "
                        "right-padded host batch -> device_payload (int16 edge types, packing facts) in 4 spawned DataLoader worker processes -> the loader's pin "
                        "thread -> H2D on the copy stream (DevicePrefetcher) -> step; host-side key-tile / tile-prefix arithmetic and their uploads inside the timed region",
            "unit": "molecules/s", "steps": n, "value": round(B * world * n / dt, 2), "ms_per_step": round(dt / n * 1e3, 3),
            "collate_ms": round(sum(t_collate) / max(1, len(t_collate)) * 1e3, 3), "h2d_ms": ms(h2d), "optimizer_ms": ms(opt),
            "note": "collate_ms: time one worker process spends on one batch (right-pad + device_payload; 4 workers run ahead of the step); h2d_ms: the copies "
                    "of one batch, event-timed on the copy stream (in flight under the previous step); optimizer_ms: grad-norm + clip + Adam + both weight "
                    "shadows, event-timed; the clock starts at the first delivered batch (worker start-up excluded)",
            "layout": model.last_layout}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU")
    ap.add_argument("--atoms", type=int, default=128)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--no-rooflines", action="store_true", help="skip the extra (untimed) steps that time every kernel family")
    ap.add_argument("--graph", action="store_true", help="replay the step from one captured HIP graph (FineTuner.graphed_step; single GPU)")
    ap.add_argument("--ragged", action="store_true", help="molecules of mixed length padded to the batch maximum (not the headline workload)")
    ap.add_argument("--no-ragged-workload", action="store_true", help="skip the second (mixed-length) workload record of the default run")
    ap.add_argument("--no-pipeline-workload", action="store_true", help="skip the fresh-batch-every-step workload record")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

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
    tuner = FineTuner(model, "classification", total_steps=10_000, distributed=(world > 1 or os.environ.get("MMDTI_FORCE_DDP") == "1"))
    from mmdti_hip.collate import packing_fields, atom_counts

    def resident(seed, ragged):
        """a synthetic collated batch resident in HBM in the reference's format (int64 edge types), plus -- for a ragged batch --
        the host-side lengths collate.device_payload attaches (the kernels then skip padding: packed token rows)"""
        _, b, y = synth(args.batch, args.atoms, args.tokens, seed=seed, ragged=ragged)
        host = dict(packing_fields(b), atom_counts=atom_counts(b["src_tokens"], 0)) if ragged else {}
        b = {k: v.to(dev) for k, v in b.items()}
        b.update(host)
        return b, y.to(dev)

    batch, label = resident(1234 + rank, args.ragged)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step = tuner.graphed_step if (args.graph and world == 1) else tuner.step
    for _ in range(args.warmup):
        out = step(batch, label, epoch=0)
    barrier()
    # inside the timed region only the two pair-attention kernels are event-timed (30 launches per step); everything else is
    # timed in extra steps afterwards so that ~1 000 event pairs per step do not perturb `value`
    ops.kernel_timer.enable(("pair_attn_bwd", "pair_attn_fwd"))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(batch, label, epoch=0)
    barrier()
    dt = time.perf_counter() - t0
    in_step = ops.kernel_timer.summary()
    ops.kernel_timer.disable()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    losses = {"loss": float(out.loss), "task": float(out.task_loss), "infonce": float(out.infonce_loss), "ct": float(out.ct_loss)}
    headline_layout = model.last_layout

    # Second workload, timed in the same process (not the headline): SURVEY 8d's C-main length distribution -- atoms ~ N(48, 20^2)
    # clamped to [8, 128], SMILES 0.8 x atoms, padded to the batch maximum as the reference collates.  Timed on the layout the DEFAULT
    # picks for a training step with dropout on -- the reference's padded rows (all-padding key tiles skipped: exact), because a padded
    # row of the reference draws its own dropout mask -- and on the packed token rows (strict_reference=False: opt-in for training, the
    # default wherever no dropout is live).
    workloads = {}
    sr_default = model.strict_reference
    if not args.ragged and not args.no_ragged_workload:
        rb, rl = resident(4321 + rank, True)
        rec = {}
        for tag, strict in (("default", sr_default), ("packed", False)):
            model.strict_reference = strict
            for _ in range(max(2, args.warmup)):
                tuner.step(rb, rl, epoch=0)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ro = tuner.step(rb, rl, epoch=0)
            barrier()
            d1 = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([d1], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                d1 = float(t)
            rec[tag] = {"ms_per_step": round(d1 / args.steps * 1e3, 3), "value": round(args.batch * world * args.steps / d1, 2), "layout_ran": model.last_layout,
                        "loss_last_step": float(ro.loss)}
        pk = model._pack_cache[3] if model._pack_cache is not None else None
        Nr, Lr = int(rb["src_tokens"].shape[1]), int(rb["input_ids"].shape[1])
        workloads["ragged"] = {
            "workload": f"same step, {args.batch} molecules/GPU of mixed length (atoms ~ N(48, 20^2) clamped to [8, {args.atoms}], SMILES 0.8 x atoms) "
                        f"right-padded to the batch maximum N = {Nr}, L = {Lr}",
            "unit": "molecules/s", "steps": args.steps, "value": rec["default"]["value"], "ms_per_step": rec["default"]["ms_per_step"],
            "layout": rec["default"]["layout_ran"], "strict_reference": sr_default,
            "token_rows": None if pk is None else {"tower1_packed": pk[0].M, "tower1_padded": args.batch * Nr, "tower2_packed": pk[1].M, "tower2_padded": args.batch * Lr},
            "packed_rows_opt_in": {"ms_per_step": rec["packed"]["ms_per_step"], "value": rec["packed"]["value"], "layout": rec["packed"]["layout_ran"],
                                   "strict_reference": False},
            "note": "default (strict_reference=None): a training step with dropout on computes the reference's padded rows (each draws its own dropout mask; only the "
                    "all-padding key tiles are skipped, which is exact); packed token rows -- every sequence's real tokens + ONE representative pad row weighted "
                    "by the padded positions it stands for in the unmasked InfoNCE mean -- are the reference's computation at dropout 0 (the default for eval / "
                    "predict) and an opt-in for training (equal in expectation only; DESIGN.md section 3)"}
        model.strict_reference = sr_default
        # Third workload: the reference's REAL batch size (finetune.py:14 batch_size=32, config/default.yaml:18 16) on the same length
        # distribution -- a chain of ~480 small kernels per step, where launch structure, not arithmetic, sets the time
        if args.batch > 32:
            small = 32
            _, sb, sy = synth(small, args.atoms, args.tokens, seed=8765 + rank, ragged=True)
            shost = dict(packing_fields(sb), atom_counts=atom_counts(sb["src_tokens"], 0))
            sb = dict({k: v.to(dev) for k, v in sb.items()}, **shost)
            sy = sy.to(dev)
            for first_small, strict in ((True, sr_default), (False, False)):
                model.strict_reference = strict
                for _ in range(max(10, args.warmup)):
                    tuner.step(sb, sy, epoch=0)
                barrier()
                t1 = time.perf_counter()
                n_small = max(args.steps, 50)        # (8 ms steps paced by launch issue: short runs scatter by a millisecond)
                for _ in range(n_small):
                    tuner.step(sb, sy, epoch=0)
                barrier()
                d1 = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([d1], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    d1 = float(t)
                sm = {"unit": "molecules/s", "steps": n_small, "value": round(small * world * n_small / d1, 2), "ms_per_step": round(d1 / n_small * 1e3, 3),
                      "layout": model.last_layout}
                if first_small:
                    workloads["small_batch"] = dict(sm, workload=f"same step, {small} molecules/GPU (the reference's default batch size), mixed lengths, default layout "
                                                                  "(padded rows under dropout)", padded_N=int(sb["src_tokens"].shape[1]), strict_reference=sr_default)
                else:
                    workloads["small_batch"]["packed_rows_opt_in"] = sm
            model.strict_reference = sr_default

    # Fourth record: the headline batch with bf16 operands everywhere (MMDTI_FWD_FP16=0, the round-1..3 default) -- faster by the
    # in-kernel conversions of the fp16 mode, and outside the north star's 1e-3 on embeddings (4.6e-3): the A/B of the precision choice
    if not args.ragged and not args.no_ragged_workload:
        was16 = ops.FWD_F16
        ops.set_forward_fp16(not was16)
        try:
            for _ in range(max(2, args.warmup)):
                tuner.step(batch, label, epoch=0)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                fo = tuner.step(batch, label, epoch=0)
            barrier()
            d1 = time.perf_counter() - t1
        finally:
            ops.set_forward_fp16(was16)
        if world > 1:
            t = torch.tensor([d1], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            d1 = float(t)
        other = "bf16_operands" if was16 else "fp16_forward_operands"
        workloads[other] = {
            "workload": ("the headline batch with bf16 operands in every GEMM, forward and backward (MMDTI_FWD_FP16=0)" if was16 else
                         "the headline batch with fp16 forward operands (the default mode; this run was started with MMDTI_FWD_FP16=0)"),
            "unit": "molecules/s", "steps": args.steps, "value": round(args.batch * world * args.steps / d1, 2), "ms_per_step": round(d1 / args.steps * 1e3, 3),
            "loss_last_step": float(fo.loss),
            "parity": "bf16 operands vs the reference's own fp32 run (B = 32 fixture): encoder_rep 4.6e-3, out_bert 2.1e-3, logits 2.4-2.8e-3, InfoNCE 1.3e-4 relative "
                      "-- embeddings outside the north star's 1e-3; fp16 forward operands (the headline): config.parity"}
        for _ in range(2):      # back in the headline mode before the per-family steps
            tuner.step(batch, label, epoch=0)
    # (N = 1 only: the loader's worker processes are per rank -- four more processes on every GPU of a node say nothing the one-GPU
    #  record does not, and the scaling runs stay what they measure)
    if not args.ragged and not args.no_ragged_workload and not args.no_pipeline_workload and not args.graph and world == 1:
        workloads["pipeline"] = pipeline_workload(tuner, model, args, dev, world, rank, barrier)

    # every kernel family in two extra modes (every rank runs them: the step holds collectives):
    #   "overlapped": streams as in the timed region (a launch's event time includes sharing the chip with the other tower);
    #   "alone"     : towers back to back on one stream, every stream-level overlap off -- each launch has the chip to itself.
    fam = {}
    extra = 0 if args.no_rooflines else 2
    if extra:
        from mmdti_hip import functional as Fn
        was = (model.overlap_towers, model.infonce_on_side_stream, model.cross_modal_module.two_streams, Fn.DEFER_WGRAD_LAYERS)
        for mode, ov in (("overlapped", True), ("alone", False)):
            if not ov:
                model.overlap_towers, model.infonce_on_side_stream, model.cross_modal_module.two_streams, Fn.DEFER_WGRAD_LAYERS = False, False, False, 0
            ops.kernel_timer.enable(tuple(FAMILIES))
            for _ in range(extra):
                tuner.step(batch, label, epoch=0)
            fam[mode] = ops.kernel_timer.summary()
            ops.kernel_timer.disable()
        model.overlap_towers, model.infonce_on_side_stream, model.cross_modal_module.two_streams, Fn.DEFER_WGRAD_LAYERS = was
    barrier()

    if rank == 0:
        N = int(batch["src_tokens"].shape[1])
        rooflines, roofline = [], None
        if fam:
            rooflines = family_rooflines(fam["alone"], extra)
            over = {r["kernel"]: r for r in family_rooflines(fam["overlapped"], extra)}
            for r in rooflines:
                o = over.get(r["kernel"])
                if o:
                    r["achieved_in_overlapped_step"] = o["achieved"]
                    r["ms_per_step_overlapped"] = o["ms_per_step"]
            # the two pair-attention kernels were also timed INSIDE the timed region: report those launch times too
            for r in rooflines:
                d = in_step.get(r["kernel"])
                if d:
                    r["mean_launch_ms_in_timed_region"] = round(d["mean_ms"], 4)
                    r["launches_timed_in_region"] = d["n"]
            roofline = dict(rooflines[0])               # the family with the most GPU time per step
            roofline["kernel"] = {"gemm": "gemm_glds_kernel / gemm_glds_tall_kernel / gemm_big_kernel / gemm_bf16_kernel (all GEMM launches of a step)"}.get(
                roofline["kernel"], roofline["kernel"])
            roofline["note"] = ("dominant kernel family by GPU time; HIP events on the launch stream around every launch of 2 extra steps after the timed "
                                "region with every stream-level overlap off (each launch alone on the chip); achieved = algorithmic work / summed launch time")
            # both fractions of the family: of the MFMA peak (flops) and of the HBM peak (PMC bytes of the family per step / its time)
            step_bytes, gemm_bytes, src = pmc_step_bytes()
            for r in rooflines + [roofline]:
                if r["bound"] == "mfma" and str(r["kernel"]).startswith("gemm"):
                    r["frac_mfma"] = r["frac"]
                    if gemm_bytes:
                        r["hbm_bytes_per_step"] = int(gemm_bytes)
                        r["frac_hbm"] = round(gemm_bytes / (r["ms_per_step"] * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
            if step_bytes:
                roofline["step_hbm_bytes"] = int(step_bytes)
                roofline["step_hbm_frac_of_peak"] = round(step_bytes / (dt / args.steps) / (HBM_PEAK_GBS * 1e9), 4)
                roofline["step_hbm_source"] = src
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # (rank 0 at N = 1 only: the other ranks of a multi-GPU run would sit in a collective for a minute)
            cpu = cpu_baseline(args.atoms, args.tokens, args.cpu_sample)
        mols = args.batch * world * args.steps
        shape = "mixed lengths padded to the batch maximum" if args.ragged else "all at max length"
        line = {
            "metric": "molecules/sec fwd+bwd (InfoNCE fine-tune)", "value": round(mols / dt, 2), "unit": "molecules/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16 forward / bf16 backward GEMM operands, fp32 accumulation" if ops.FWD_F16 else "bf16 GEMM operands, fp32 accumulation", "data": "synthetic",
            "config": {"workload": f"BBBP-like classification + SupCon + InfoNCE fine-tune step, {args.batch} molecules/GPU x {args.atoms} atoms x "
                                   f"{args.tokens} SMILES tokens ({shape}), fwd+bwd+allreduce+clip+Adam, dropout on",
                       "global_batch": args.batch * world, "atoms": args.atoms, "tokens": args.tokens, "padded_N": N, "parallelism": f"dp{world}",
                       "launch": "one HIP graph per step" if (args.graph and world == 1) else "eager (one launch per kernel)",
                       "precision_mode": "fp16 forward operands (default)" if ops.FWD_F16 else "bf16 operands (MMDTI_FWD_FP16=0)",
                       "parity": parity_record() if ops.FWD_F16 else None,
                       "token_layout": headline_layout, "strict_reference": sr_default, "unimol": "15L/512/64h", "chemberta_assumed": "6L/512/8h/ffn2048/vocab600", "infonce_negatives": "global",
                       "grad_buckets_reduced_during_backward": None if tuner.reducer is None or not tuner.reducer.active
                       else f"{tuner.reducer.overlapped}/{len(tuner.reducer.buckets)}"},
            "losses_last_step": losses, "roofline": roofline, "rooflines": rooflines, "cpu_baseline": cpu, "workloads": workloads,
        }
        print(json.dumps(line))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
