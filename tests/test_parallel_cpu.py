"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel host logic in mmdti_hip/parallel.py:
global InfoNCE negatives (all-gather + its adjoint), bucketed arena all-reduce, batch sharding.

No GPU and no HIP compute here: the per-rank arithmetic is the CPU oracle, which is exactly what lets these tests pin
the SEMANTICS chosen for DDP: two ranks on a global batch reproduce the single-process loss and
gradients."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, initfile):
    for p in (ROOT, os.path.join(ROOT, "mm-dti_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"file://{initfile}")


class _AllGatherFn(torch.autograd.Function):
    """autograd wrapper used only by this test: forward = GlobalNegatives.gather, backward = its reduce_scatter."""

    @staticmethod
    def forward(ctx, x, negs):
        ctx.negs = negs
        return negs.gather(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.negs.reduce_scatter(g.clone()), None


def _worker_global_negatives(rank, world, initfile, out):
    _init(rank, world, initfile)
    from mmdti_hip.parallel import GlobalNegatives, shard_batch
    from oracle import mmdti_oracle as O
    torch.manual_seed(0)
    B, d = 8, 50
    q_all = torch.randn(B, d)
    k_all = torch.randn(B, d)
    label = torch.arange(B).view(B, 1)
    (sh, lab) = shard_batch({"q": q_all, "k": k_all}, label, rank, world)
    assert lab[0, 0] == rank * (B // world)
    negs = GlobalNegatives()
    q = sh["q"].clone().requires_grad_()
    k = sh["k"].clone().requires_grad_()
    both = _AllGatherFn.apply(torch.cat((q, k), 1), negs)               # ONE fused message for both towers
    qg, kg = both[:, :d], both[:, d:]
    qh, kh = torch.nn.functional.normalize(qg, dim=-1), torch.nn.functional.normalize(kg, dim=-1)
    logits = qh @ kh.T / 0.1
    b = B // world
    rows = slice(negs.row0(b), negs.row0(b) + b)
    lse_a = torch.logsumexp(logits[rows], 1) - logits[rows].diagonal(offset=rows.start)
    lse_b = torch.logsumexp(logits.T[rows], 1) - logits.T[rows].diagonal(offset=rows.start)
    share = (lse_a.sum() + lse_b.sum()) / (2 * B)                        # this rank's share of the global loss
    (share * world).backward()                                           # trainer.py: x world, then gradients are AVERAGED
    g = torch.cat((q.grad, k.grad), 1)
    dist.all_reduce(share)                                               # sum of shares == global loss
    # reference: single process on the global batch
    qr, kr = q_all.clone().requires_grad_(), k_all.clone().requires_grad_()
    ref = O.info_nce(qr, kr, temperature=0.1)
    ref.backward()
    gref = torch.cat((qr.grad, kr.grad), 1)[rows]
    torch.testing.assert_close(share, ref.detach(), rtol=1e-5, atol=1e-6)
    # after the rank-mean of gradients, a parameter sees (1/world) * sum_r grad_r ; the embedding gradient of OWN rows is
    # only produced on this rank, so grad_r / world must equal the single-process gradient
    torch.testing.assert_close(g / world, gref, rtol=1e-4, atol=1e-6)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


class _FakeArena:
    def __init__(self, n):
        self.numel = n
        self.grad = torch.zeros(n)


def _worker_reducer(rank, world, initfile, out):
    _init(rank, world, initfile)
    from mmdti_hip.parallel import ArenaReducer
    n = 1000
    arena = _FakeArena(n)
    arena.grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
    red = ArenaReducer(arena, bucket_bytes=4 * 256)                      # 4 buckets of 256 floats
    assert len(red.buckets) == 4 and red.buckets[-1] == (768, 1000)
    red.reduce_range(512, 1000)                                          # "backward finished the tail of the arena first"
    assert sorted(red._pending) == [(512, 768), (768, 1000)]
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * (1 + 2) / 2          # mean over the two ranks
    torch.testing.assert_close(arena.grad, expect)
    assert red._pending == []
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


def _worker_reducer_direct(rank, world, initfile, out):
    """MMDTI_REDUCE=direct: reduce-scatter as one all-to-all + all-gather (the xGMI-shaped alternative to a ring all-reduce) gives
    the same mean as the all-reduce, for bucket lengths that do and do not divide by the rank count."""
    os.environ["MMDTI_REDUCE"] = "direct"
    _init(rank, world, initfile)
    from mmdti_hip.parallel import ArenaReducer
    n = 1001                                                             # last bucket: 233 floats, odd
    arena = _FakeArena(n)
    g = torch.Generator().manual_seed(7 + rank)
    mine = torch.randn(n, generator=g)
    arena.grad.copy_(mine)
    red = ArenaReducer(arena, bucket_bytes=4 * 256)
    assert red.algo == "direct" and red.buckets[-1] == (768, 1001)
    red.finish()
    both = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(both, mine)
    torch.testing.assert_close(arena.grad, (both[0] + both[1]) / 2, rtol=1e-6, atol=1e-7)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


class _ParamArena(_FakeArena):
    """arena stand-in that knows its parameters (sizes 300, 212, 256, 232 -> offsets 0, 300, 512, 768)."""

    def __init__(self):
        super().__init__(1000)
        self.params = [torch.nn.Parameter(torch.zeros(k)) for k in (300, 212, 256, 232)]
        self.offsets, off = {}, 0
        for p in self.params:
            self.offsets[id(p)] = off
            off += p.numel()


def _worker_reducer_hook(rank, world, initfile, out):
    """on_grads_ready: a bucket leaves as soon as every parameter overlapping it has reported, never before."""
    _init(rank, world, initfile)
    from mmdti_hip.parallel import ArenaReducer
    arena = _ParamArena()
    base = torch.arange(1000, dtype=torch.float32)
    red = ArenaReducer(arena, bucket_bytes=4 * 256)                      # buckets [0,256) [256,512) [512,768) [768,1000)
    p0, p1, p2, p3 = arena.params
    for step in range(2):                                                # second step: begin_step() forgets the first
        arena.grad.copy_(base * (rank + 1))
        red.begin_step()
        red.on_grads_ready([p3])
        assert red._pending == [(768, 1000)]
        red.on_grads_ready([p1])                                         # bucket 1 also needs p0 (rows 256..299)
        assert red._pending == [(768, 1000)]
        red.on_grads_ready([p2])
        assert red._pending == [(768, 1000), (512, 768)]
        torch.testing.assert_close(arena.grad[512:], base[512:] * 1.5)  # reduced (mean of x1, x2) ...
        torch.testing.assert_close(arena.grad[:512], base[:512] * (rank + 1))   # ... and nothing else touched yet
        red.on_grads_ready([p0])
        assert sorted(red._pending) == [(0, 256), (256, 512), (512, 768), (768, 1000)] and red.overlapped == 4
        red.finish()                                                     # nothing left: no bucket is reduced twice
        torch.testing.assert_close(arena.grad, base * 1.5)
    # a parameter that never reports leaves its buckets to finish()
    arena.grad.copy_(base * (rank + 1))
    red.begin_step()
    red.on_grads_ready([p0, p2, p3])
    assert sorted(red._pending) == [(0, 256), (512, 768), (768, 1000)] and red.unreported() == [set(), {id(p1)}, set(), set()]
    red.finish()
    torch.testing.assert_close(arena.grad, base * 1.5)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


def _worker_step_equivalence(rank, world, initfile, out):
    """Full tiny model: rank-local oracle steps with global negatives + averaged gradients == one global step with
    (global InfoNCE, rank-mean of local task/CT losses)."""
    _init(rank, world, initfile)
    from mmdti_hip.parallel import GlobalNegatives, shard_batch
    from oracle import mmdti_oracle as O
    cfg = O.ModelCfg(unimol=O.UniMolCfg(layers=1, dim=32, ffn=64, heads=4, K=8, vocab=31),
                     roberta=O.RobertaCfg(layers=1, dim=32, heads=2, ffn=64, vocab=40, max_pos=40),
                     cross=O.CrossCfg(dim=32, heads=2, ffn=64), task="classification", output_dim=2)
    P = {k: v.requires_grad_() for k, v in O.init_params(cfg, seed=3, std=0.1).items()}
    batch, label = O.synth_batch(4, 6, 9, cfg, seed=11, ragged=True)      # GLOBAL batch, padded to global max lengths
    negs = GlobalNegatives()
    sh, lab = shard_batch(batch, label, rank, world)
    enc, bert, pooled = O.mm_features(sh, P, cfg)
    a, b = O.infonce_embed(enc, bert, P, p=0.0)
    both = _AllGatherFn.apply(torch.cat((a, b), 1), negs)
    d = a.shape[1]
    Bg, bl = both.shape[0], a.shape[0]
    qh, kh = torch.nn.functional.normalize(both[:, :d], dim=-1), torch.nn.functional.normalize(both[:, d:], dim=-1)
    logits = qh @ kh.T / 0.1
    r0 = negs.row0(bl)
    rows = slice(r0, r0 + bl)
    share = ((torch.logsumexp(logits[rows], 1) - logits[rows].diagonal(offset=r0)).sum()
             + (torch.logsumexp(logits.T[rows], 1) - logits.T[rows].diagonal(offset=r0)).sum()) / (2 * Bg)
    logits_head = O.classification_head(pooled, P)
    ct = O.ct_single(pooled, lab, logits_head)
    loss = O.task_loss(logits_head, lab, cfg.task) + 0.1 * share * world + 0.1 * ct
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel()) for p in P.values()])
    dist.all_reduce(flat)
    flat /= world
    # single-process reference of the same semantics
    Pr = {k: v.detach().clone().requires_grad_() for k, v in P.items()}
    enc, bert, pooled = O.mm_features(batch, Pr, cfg)
    infonce = O.infonce_forward(enc, bert, Pr, p=0.0)
    lh = O.classification_head(pooled, Pr)
    half = label.shape[0] // world
    tl = sum(O.task_loss(lh[i * half:(i + 1) * half], label[i * half:(i + 1) * half], cfg.task) for i in range(world)) / world
    ctr = sum(O.ct_single(pooled[i * half:(i + 1) * half], label[i * half:(i + 1) * half], None) for i in range(world)) / world
    (tl + 0.1 * infonce + 0.1 * ctr).backward()
    ref = torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel()) for p in Pr.values()])
    torch.testing.assert_close(flat, ref, rtol=2e-3, atol=2e-6)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


def _worker_fds_stats_identical(rank, world, initfile, out):
    """FDS epoch pass under DDP: each rank holds the pooled features of ITS shard; after parallel.gather_features every rank
    updates its FDS buffers from the whole epoch -> buffers identical on all ranks and equal to the single-process run."""
    _init(rank, world, initfile)
    from mmdti_hip.parallel import GlobalNegatives, gather_features
    from oracle import mmdti_oracle as O
    g = torch.Generator().manual_seed(5)
    feats_all, labels_all = torch.randn(24, 16, generator=g), torch.randn(24, 1, generator=g)
    b = 24 // world
    negs = GlobalNegatives()
    f, y = gather_features(negs, feats_all[rank * b:(rank + 1) * b].clone(), labels_all[rank * b:(rank + 1) * b].clone())
    assert f.shape == (24, 16) and torch.equal(f, feats_all) and torch.equal(y, labels_all)
    def run(ff, yy):
        fo = O.FDSOracle(16, -2.5, 0.5, bucket_num=10, kernel="gaussian", ks=5, sigma=1)
        fo.update_last_epoch_stats(0); fo.update_running_stats(ff, yy, 0); fo.update_last_epoch_stats(1)
        return fo.state()
    mine, ref = run(f, y), run(feats_all, labels_all)
    for k in mine:
        assert torch.equal(mine[k], ref[k]), k
        other = mine[k].clone()
        dist.broadcast(other, src=0)
        assert torch.equal(other, mine[k]), k                                   # identical across ranks, bit for bit
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


def _worker_bucket_sampler_global_infonce(rank, world, initfile, out):
    """LengthBucketBatchSampler(rank, world) deals equal-size batches; every rank collates its own molecules, pads to the
    GLOBAL lengths (parallel.pad_to_global_lengths) and joins the global InfoNCE: the sum of the ranks' shares equals the
    single-process InfoNCE on the union of the two batches collated together."""
    _init(rank, world, initfile)
    from mmdti_hip.parallel import GlobalNegatives, pad_to_global_lengths
    from mmdti_hip.data import LengthBucketBatchSampler
    from mmdti_hip.collate import right_pad
    from oracle import mmdti_oracle as O
    import numpy as np
    cfg = O.ModelCfg(unimol=O.UniMolCfg(layers=1, dim=32, ffn=64, heads=4, K=8, vocab=31),
                     roberta=O.RobertaCfg(layers=1, dim=32, heads=2, ffn=64, vocab=40, max_pos=40),
                     cross=O.CrossCfg(dim=32, heads=2, ffn=64), task="classification", output_dim=2)
    P = O.init_params(cfg, seed=3, std=0.1)
    rng = np.random.default_rng(2)
    mols = []
    for _ in range(37):                                                    # 37 molecules, batch 4, 2 ranks: a ragged remainder
        na, nl = int(rng.integers(2, 9)), int(rng.integers(4, 12))
        d = O.coords2unimol(rng.integers(4, 30, size=na), rng.normal(0, 3, size=(na, 3)), 31)
        ids = np.concatenate([[0], rng.integers(4, 40, size=nl - 2), [2]]).astype(np.int64)
        mols.append((d, ids))
    atoms, toks = [len(m[0]["src_tokens"]) - 2 for m in mols], [len(m[1]) for m in mols]
    def collate(idx):
        ids = right_pad([torch.from_numpy(mols[i][1]) for i in idx], 1)
        return {"src_tokens": right_pad([torch.from_numpy(mols[i][0]["src_tokens"]) for i in idx], 0),
                "src_distance": right_pad([torch.from_numpy(mols[i][0]["src_distance"]) for i in idx], 0.0, square=True),
                "src_edge_type": right_pad([torch.from_numpy(mols[i][0]["src_edge_type"]) for i in idx], 0, square=True),
                "input_ids": ids, "attention_mask": ids.ne(1).long()}
    mine = list(LengthBucketBatchSampler(atoms, toks, 4, shuffle=True, seed=7, rank=rank, world=world))
    both = [list(LengthBucketBatchSampler(atoms, toks, 4, shuffle=True, seed=7, rank=r, world=world)) for r in range(world)]
    assert len(mine) == len(both[1 - rank]) and all(len(b) == 4 for b in mine)          # same step count, same B_loc at every step
    negs = GlobalNegatives()
    for step in range(2):
        batch = pad_to_global_lengths(collate(mine[step]))
        enc, bert, _ = O.mm_features(batch, P, cfg)
        a, b = O.infonce_embed(enc, bert, P, p=0.0)
        allv = negs.gather(torch.cat((a, b), 1))
        d = a.shape[1]
        qh, kh = torch.nn.functional.normalize(allv[:, :d], dim=-1), torch.nn.functional.normalize(allv[:, d:], dim=-1)
        logits = qh @ kh.T / 0.1
        r0 = negs.row0(4)
        rows = slice(r0, r0 + 4)
        share = ((torch.logsumexp(logits[rows], 1) - logits[rows].diagonal(offset=r0)).sum()
                 + (torch.logsumexp(logits.T[rows], 1) - logits.T[rows].diagonal(offset=r0)).sum()) / (2 * allv.shape[0])
        dist.all_reduce(share)
        union = collate(both[0][step] + both[1][step])                     # single process: the union batch collated together
        enc, bert, _ = O.mm_features(union, P, cfg)
        ref = O.infonce_forward(enc, bert, P, p=0.0)
        torch.testing.assert_close(share, ref, rtol=1e-5, atol=1e-6)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


def _worker_unequal_local_batches_raise(rank, world, initfile, out):
    """GlobalNegatives.gather refuses ranks that hold different B_loc at a step (it would otherwise hang or mis-slice in RCCL)."""
    _init(rank, world, initfile)
    from mmdti_hip.parallel import GlobalNegatives
    negs = GlobalNegatives()
    with pytest.raises(RuntimeError, match="different local batch sizes"):
        negs.gather(torch.zeros(4 + rank, 6))
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


def _worker_trainer_ddp_plumbing(rank, world, initfile, out):
    """The device-independent parts of ``tasks.Trainer(distributed=True)`` on 2 gloo ranks: the
    sampler deals disjoint shards with EQUAL step counts; ``decorate_torch_batch`` pads every rank's batch to the global lengths
    with host-side integers; the checkpoint is written by rank 0 only, atomically, and visible to every rank after the barrier;
    the early-stop / improved decision of every rank is rank 0's (a near-tie must not split the ranks)."""
    _init(rank, world, initfile)
    import numpy as np
    from mmdti_hip.tasks import Trainer
    from mmdti_hip.collate import collate_batch
    from oracle import mmdti_oracle as O
    td = os.path.dirname(out)
    trainer = Trainer(save_path=td, task="regression", metrics="none", batch_size=3, epochs=2, use_cuda=False, distributed=True, seed=11,
                      narrow_inputs=True)
    assert trainer._ddp() and trainer.rank == rank

    class _Tok:
        def __call__(self, smiles, padding=True, truncation=True, return_tensors="pt"):
            L = max(len(s) for s in smiles) + 2
            ids = torch.ones(len(smiles), L, dtype=torch.long)
            att = torch.zeros(len(smiles), L, dtype=torch.long)
            for r, s in enumerate(smiles):
                ids[r, :len(s) + 2] = torch.tensor([0] + [5 + (ord(c) % 7) for c in s] + [2])
                att[r, :len(s) + 2] = 1
            return {"input_ids": ids, "attention_mask": att}

    class _Model(torch.nn.Module):          # what the trainer touches outside the step: collate, padding index, dictionary, state dict
        padding_idx, dictionary, tokenizer = 0, list(range(31)), _Tok()

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.full((3,), float(rank)))      # ranks hold DIFFERENT values: only rank 0's may be saved

        def batch_collate_fn(self, samples):
            return collate_batch(samples, 0, self.tokenizer)

    rng = np.random.default_rng(1)
    data = []
    for i in range(17):                                                    # 17 samples, 2 ranks, batch 3: shards of 8 -> 2 steps each
        na = int(rng.integers(2, 12))
        d = O.coords2unimol(rng.integers(4, 30, size=na), rng.normal(0, 3, size=(na, 3)), 31)
        d["smile"] = "C" * int(rng.integers(2, 15))
        data.append((d, np.array([float(i)], dtype=np.float32)))
    model = _Model()
    loader, sampler = trainer.train_loader(model, data)
    assert sampler is not None
    sampler.set_epoch(0)
    seen, shapes = [], []
    for batch in loader:
        net_input, target = trainer.decorate_torch_batch(batch)
        seen += [int(v) for v in target.flatten().tolist()]
        shapes.append((net_input["src_tokens"].shape[1], net_input["input_ids"].shape[1]))
        assert net_input["src_edge_type"].dtype == torch.int16 and net_input["packable"] and len(net_input["atom_counts"]) == 3
        assert net_input["src_distance"].shape[1:] == (shapes[-1][0],) * 2
    steps = torch.tensor([len(shapes)])
    all_steps = [torch.zeros_like(steps) for _ in range(world)]
    dist.all_gather(all_steps, steps)
    assert all(int(t) == len(shapes) == 2 for t in all_steps)              # equal step counts
    mine = torch.tensor(seen + [-1] * (8 - len(seen)))
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    a, b = (set(int(v) for v in t.tolist() if v >= 0) for t in both)
    assert not (a & b) and len(a) == len(b) == 6                           # disjoint shards
    sh = torch.tensor(shapes)
    both_sh = [torch.zeros_like(sh) for _ in range(world)]
    dist.all_gather(both_sh, sh)
    assert torch.equal(both_sh[0], both_sh[1])                              # every step: ONE padded length on all ranks
    # decisions: rank 1 sees a (slightly) worse value than its best, rank 0 a better one -> both follow rank 0 and rank 0 saves
    value = 1.0 - 1e-7 if rank == 0 else 1.0 + 1e-7
    stop, best, wait, _ = trainer._early_stop_choice(0, value, 1.0, {"mse": value}, float("-inf"), model, td, 0, patience=1, epoch=0)
    assert not stop and wait == 0 and best == value
    trainer._checkpoint_barrier()
    path = os.path.join(td, "model_0.pth")
    ck = torch.load(path, weights_only=True)["model_state_dict"]
    assert torch.equal(ck["w"], torch.zeros(3)) and not [f for f in os.listdir(td) if ".tmp" in f]
    # ... and the other way round: rank 0 sees no improvement, rank 1 would have -> nobody saves, both count a wait and stop together
    trainer._checkpoint_barrier()               # (every rank has read the file)
    os.remove(path) if rank == 0 else None
    trainer._checkpoint_barrier()
    value = 2.0 if rank == 0 else 0.5
    stop, best, wait, _ = trainer._early_stop_choice(0, value, 1.0, {"mse": value}, float("-inf"), model, td, 0, patience=1, epoch=0)
    trainer._checkpoint_barrier()
    assert stop and wait == 1 and best == 1.0 and not os.path.exists(path)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("worker", [_worker_trainer_ddp_plumbing, _worker_global_negatives, _worker_reducer, _worker_reducer_direct, _worker_reducer_hook, _worker_step_equivalence,
                                    _worker_fds_stats_identical, _worker_bucket_sampler_global_infonce, _worker_unequal_local_batches_raise])
def test_two_ranks_gloo(worker):
    with tempfile.TemporaryDirectory() as td:
        initfile, out = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(worker, args=(2, initfile, out), nprocs=2, join=True)
        assert torch.load(out)["ok"]


def test_shard_batch_requires_even_split():
    sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
    from mmdti_hip.parallel import shard_batch
    with pytest.raises(AssertionError):
        shard_batch({"x": torch.zeros(5, 2)}, torch.zeros(5, 1), 0, 2)
