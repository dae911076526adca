"""Host logic of the input side (mmdti_hip/data.py): length-bucketed batching."""
import sys, os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mm-dti_amd"))


def _lengths(n=5000, seed=0):
    rng = np.random.default_rng(seed)
    atoms = np.clip(rng.lognormal(3.2, 0.45, n).astype(int), 4, 128)          # drug-like: median ~25 heavy atoms, long tail
    tokens = np.clip((atoms * rng.uniform(1.6, 2.4, n)).astype(int), 8, 256)
    return atoms, tokens


def test_bucket_sampler_covers_every_index_once_and_cuts_padding():
    from mmdti_hip.data import LengthBucketBatchSampler, epoch_cost
    atoms, tokens = _lengths()
    s = LengthBucketBatchSampler(atoms, tokens, batch_size=64, shuffle=True, seed=3)
    batches = list(s)
    flat = sorted(i for b in batches for i in b)
    assert flat == list(range(len(atoms))) and len(batches) == len(s) and max(len(b) for b in batches) == 64
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(atoms))
    naive = [perm[i:i + 64].tolist() for i in range(0, len(atoms), 64)]
    assert epoch_cost(batches, atoms, tokens) < 0.45 * epoch_cost(naive, atoms, tokens)     # > 2x less padded work
    # batches differ between epochs (contrastive partners change), deterministically
    s.set_epoch(1)
    b1 = list(s)
    s.set_epoch(1)
    assert list(s) == b1 and b1 != batches
    assert sorted(i for b in b1 for i in b) == list(range(len(atoms)))


def test_bucket_sampler_shards_whole_batches_evenly():
    from mmdti_hip.data import LengthBucketBatchSampler
    atoms, tokens = _lengths(1000, seed=2)
    shards = [list(LengthBucketBatchSampler(atoms, tokens, 32, seed=5, drop_last=True, rank=r, world=4)) for r in range(4)]
    assert len({len(s) for s in shards}) == 1 and len(shards[0]) == (1000 // 32) // 4
    seen = [i for s in shards for b in s for i in b]
    assert len(seen) == len(set(seen))                                                       # no sample on two ranks
    assert all(len(b) == 32 for s in shards for b in s)


def test_device_prefetcher_passes_batches_through_unchanged_on_cpu():
    import torch
    from mmdti_hip.data import DevicePrefetcher
    batches = [({"a": torch.arange(6).view(2, 3) + i, "m": torch.ones(2, 2, dtype=torch.bool)}, torch.tensor([[i], [0]])) for i in range(4)]
    out = list(DevicePrefetcher(batches, "cpu"))
    assert len(out) == 4
    for (bi, li), (bo, lo) in zip(batches, out):
        assert set(bi) == set(bo) and all(torch.equal(bi[k], bo[k]) and bi[k].dtype == bo[k].dtype for k in bi) and torch.equal(li, lo)
    assert list(DevicePrefetcher([], "cpu")) == []


def test_bucket_sampler_equal_batches_on_every_rank_without_drop_last():
    """with world > 1 and drop_last=False a short last batch used to be dealt to ONE rank, so ranks met a step with
    different B_loc (the InfoNCE all-gather then hangs or mis-slices).  Every dealt batch is full now, at every step."""
    from mmdti_hip.data import LengthBucketBatchSampler
    atoms, tokens = _lengths(100, seed=4)
    for epoch in range(20):
        shards = []
        for r in range(2):
            s = LengthBucketBatchSampler(atoms, tokens, 8, shuffle=True, seed=1, drop_last=False, rank=r, world=2)
            s.set_epoch(epoch)
            shards.append(list(s))
            assert len(shards[-1]) == len(s)
        assert len(shards[0]) == len(shards[1])
        assert all(len(a) == len(b) == 8 for a, b in zip(*shards))
        seen = [i for sh in shards for b in sh for i in b]
        assert len(seen) == len(set(seen))


def test_bucket_sampler_drop_last_is_not_size_biased():
    """drop_last used to drop the tail of the size-sorted order -- always the largest molecules.  The remainder is
    drawn at random each epoch (as the reference's shuffled drop_last DataLoader does, tasks/trainer.py:143-150)."""
    from mmdti_hip.data import LengthBucketBatchSampler
    atoms, tokens = _lengths(203, seed=6)
    dropped = []
    for epoch in range(40):
        s = LengthBucketBatchSampler(atoms, tokens, 8, shuffle=True, seed=2, drop_last=True)
        s.set_epoch(epoch)
        kept = {i for b in s for i in b}
        assert len(kept) == 200
        dropped += [atoms[i] for i in range(203) if i not in kept]
    # dropped molecules look like the population, not like its largest members
    assert np.mean(dropped) < np.mean(atoms) + 1.0 * np.std(atoms) and min(dropped) <= np.median(atoms)
