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


# ------------------------------------------------------------------------------------------------ 8f-3: what crosses PCIe
def _mols(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        a = int(rng.integers(3, 12))
        tok = rng.integers(4, 30, size=a + 2)
        out.append(({"src_tokens": tok, "src_distance": rng.random((a + 2, a + 2)).astype("float32"),
                     "src_coord": rng.random((a + 2, 3)).astype("float32"), "src_edge_type": tok[:, None] * 31 + tok[None, :],
                     "smile": "C" * int(rng.integers(2, 9))}, [float(i)]))
    return out


class _Tok:
    """stand-in for the HF tokenizer call signature (padding=True, truncation=True, return_tensors='pt')"""
    def __call__(self, smiles, padding=True, truncation=True, return_tensors="pt"):
        import torch
        L = max(len(s) for s in smiles) + 2
        ids = torch.ones(len(smiles), L, dtype=torch.long)
        att = torch.zeros(len(smiles), L, dtype=torch.long)
        for r, s in enumerate(smiles):
            ids[r, :len(s) + 2] = torch.tensor([0] + [5 + (ord(c) % 7) for c in s] + [2])
            att[r, :len(s) + 2] = 1
        return {"input_ids": ids, "attention_mask": att}


def test_device_payload_drops_unread_fields_and_narrows_edge_types_without_changing_values():
    import torch
    from mmdti_hip.collate import collate_batch, device_payload
    batch, label = collate_batch(_mols(5, 1), 0, _Tok())
    assert batch["src_edge_type"].dtype == torch.int64 and "src_coord" in batch          # the reference's collate, untouched
    out = device_payload(batch, n_edge_types=31 * 31)
    assert "src_coord" not in out and out["src_edge_type"].dtype == torch.int16
    assert torch.equal(out["src_edge_type"].long(), batch["src_edge_type"])
    from mmdti_hip.collate import HOST_FIELDS
    assert all(out[k] is batch[k] for k in out if k not in ("src_edge_type",) + HOST_FIELDS)
    # host-side descriptors of the packed token layout: SMILES lengths, the id of the masked slots, right-padded on both sides
    assert out["packable"] is True and out["token_pad_id"] == 1 and torch.equal(out["token_counts"], batch["attention_mask"].sum(1).to(torch.int32))
    # host-side lengths for the ragged pair kernels: last non-pad position + 1 of every molecule
    st = batch["src_tokens"]
    want = torch.tensor([max(j + 1 for j in range(st.shape[1]) if int(st[b, j]) != 0) for b in range(st.shape[0])], dtype=torch.int32)
    assert torch.equal(out["atom_counts"], want) and out["atom_counts"].device.type == "cpu"
    # without the dictionary size the tensor's own range decides; an index that does not fit keeps int64
    assert device_payload(batch)["src_edge_type"].dtype == torch.int16
    big = dict(batch, src_edge_type=batch["src_edge_type"] + 40000)
    assert device_payload(big)["src_edge_type"].dtype == torch.int64
    assert device_payload(big, n_edge_types=50000)["src_edge_type"].dtype == torch.int64
    assert device_payload(device_payload(batch))["src_edge_type"].dtype == torch.int16   # idempotent


def test_worker_side_collate_yields_the_batches_of_the_in_process_collate():
    """HostCollate in DataLoader worker processes == the model-side collate in the main process (same order, same tensors,
    edge types narrowed); the collate object carries no model."""
    import pickle
    import torch
    from torch.utils.data import DataLoader
    from mmdti_hip.collate import HostCollate, collate_batch
    data = _mols(23, 3)
    hc = HostCollate(0, _Tok(), n_edge_types=961)
    assert len(pickle.dumps(hc)) < 2000
    ref = list(DataLoader(data, batch_size=4, shuffle=False, collate_fn=lambda s: collate_batch(s, 0, _Tok())))
    got = list(DataLoader(data, batch_size=4, shuffle=False, collate_fn=hc, num_workers=2))
    assert len(ref) == len(got) == 6
    for (rb, rl), (gb, gl) in zip(ref, got):
        from mmdti_hip.collate import HOST_FIELDS
        assert list(gb) == [k for k in rb if k != "src_coord"] + list(HOST_FIELDS)
        assert torch.equal(gl, rl) and gb["src_edge_type"].dtype == torch.int16
        assert all(torch.equal(gb[k].long() if k == "src_edge_type" else gb[k], rb[k]) for k in gb if k not in HOST_FIELDS)
        assert gb["packable"] is True and torch.equal(gb["token_counts"].long(), rb["attention_mask"].sum(1))


def test_ragged_key_tile_counts_and_pair_bias_prefixes_host_logic():
    """Host arithmetic of the ragged path (no GPU): the key-tile count a molecule's sweeps cover is the next SUPPORTED count (every
    count up to 9 tiles, every 2nd up to 13, every 4th beyond, and the full count), and the pair-bias kernels' tile prefixes follow
    from it -- the 4 x 4 pair blocks of the covered key tiles that hold a real pair, forward and backward alike."""
    import torch
    from mmdti_hip import ops
    for nt in range(1, 18):
        step = 1 if nt <= 9 else (2 if nt <= 13 else 4)
        got = [ops.pair_key_tiles_effective(k, nt) for k in range(0, nt + 3)]
        assert got[0] == got[1] and got[-1] == got[-2] == nt                       # clamped below and above
        for k in range(1, nt + 1):
            ke = ops.pair_key_tiles_effective(k, nt)
            assert k <= ke <= nt and (ke == nt or ke % step == 0) and ke - k < step  # the smallest supported count >= k
        assert sorted(set(got)) == sorted(set([k for k in range(step, nt, step)] + [nt]))
    N = 130                                                                        # nt = 9, nb = 33
    kt = torch.tensor([9, 3, 4, 1])
    f, b = ops.gbf_tile_prefixes(kt, N, "cpu")
    # (the tiled planes store nothing past N: forward and backward enumerate the same blocks -- those that hold a real pair)
    assert f.dtype == b.dtype == torch.int32 and f.tolist() == [0] + torch.cumsum(33 * torch.clamp(4 * kt, max=33), 0).tolist()
    assert b.tolist() == f.tolist()
    f2, _ = ops.gbf_tile_prefixes(torch.tensor([1, 11, 15]), 258, "cpu")           # nt = 17, nb = 65: counts round up to 4, 12, 16
    assert (f2[1:] - f2[:-1]).tolist() == [4 * 4 * 65, 4 * 12 * 65, 4 * 16 * 65]
