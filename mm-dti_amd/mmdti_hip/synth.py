"""Seeded synthetic batches in the reference's collated layout (SURVEY.md 8d "C-main"), for benchmarks and smoke runs.

There are no datasets in the build or benchmark environment (the reference's CSVs are "on request", RDKit is absent), so
the workload is generated: atom tokens with a drug-like element skew, Gaussian coordinates, and from them exactly what
``data/conformer.py:coords2unimol`` (:182-219) and ``MM_Model.batch_collate_fn`` (models/mm_model.py:645-682) would hand
the model -- ``[CLS] atoms [SEP]`` token rows, Euclidean distances with the two specials at the origin, edge types
``tok_i * V + tok_j``, everything right-padded with the reference's pad values -- plus random SMILES token ids
``<s> ... </s>``.  Host-side numpy only.
"""
import numpy as np
import torch

from .collate import right_pad


def molecule(rng, n_atoms, vocab, elem_p, bos=1, eos=2):
    atoms = rng.choice(vocab, size=n_atoms, p=elem_p)
    coords = rng.normal(0, 3.0, size=(n_atoms, 3)).astype(np.float32)
    tokens = np.concatenate([[bos], atoms, [eos]]).astype(np.int64)
    c = coords - coords.mean(axis=0)
    # the specials sit at the origin; (float64 zeros promote the distances to float64 before the final cast, as in conformer.py:207-210)
    c = np.concatenate([np.zeros((1, 3)), c, np.zeros((1, 3))], axis=0)
    dist = np.sqrt(((c[:, None, :] - c[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    edge = tokens.reshape(-1, 1) * vocab + tokens.reshape(1, -1)
    return tokens, dist, edge.astype(np.int64)


def synth_batch(B, max_atoms, max_tokens, task="classification", seed=1234, ragged=False, vocab=31, smiles_vocab=600, pad_idx=0,
                smiles_pad=1, n_labels=1):
    """-> (batch dict of CPU tensors, labels).  ragged: atom counts ~ clamp(N(0.375, 0.16) * max_atoms) (N(48, 20^2)
    clamped to [8, 128] at max_atoms = 128), SMILES length 0.8 x atoms; otherwise every molecule at the maximum."""
    rng = np.random.default_rng(seed)
    elem_p = np.zeros(vocab)
    elem_p[8], elem_p[4], elem_p[5], elem_p[6] = 0.5, 0.3, 0.075, 0.075                    # H, C, N, O
    rest = [i for i in range(4, vocab - 1) if i not in (4, 5, 6, 8)]
    elem_p[rest] = 0.05 / len(rest)
    toks, dists, edges, ids = [], [], [], []
    for _ in range(B):
        if ragged:
            na = int(np.clip(round(rng.normal(0.375 * max_atoms, 0.16 * max_atoms)), max(2, max_atoms // 16), max_atoms))
            nt = int(np.clip(round(0.8 * na), min(8, max(4, max_tokens // 4)), max_tokens))
        else:
            na, nt = max_atoms, max_tokens
        t, d, e = molecule(rng, na, vocab, elem_p)
        toks.append(torch.from_numpy(t)); dists.append(torch.from_numpy(d)); edges.append(torch.from_numpy(e))
        body = rng.integers(4, smiles_vocab, size=nt - 2)
        ids.append(torch.from_numpy(np.concatenate([[0], body, [2]]).astype(np.int64)))
    input_ids = right_pad(ids, smiles_pad)
    batch = {"src_tokens": right_pad(toks, pad_idx), "src_distance": right_pad(dists, 0.0, square=True),
             "src_edge_type": right_pad(edges, pad_idx, square=True), "input_ids": input_ids,
             "attention_mask": input_ids.ne(smiles_pad).long()}
    if task == "regression":
        label = torch.from_numpy(rng.normal(0, 1, size=(B, 1)).astype(np.float32))
    elif task == "multilabel_classification":
        label = torch.from_numpy((rng.random((B, n_labels)) < 0.2).astype(np.int64))
    else:
        label = torch.from_numpy((rng.random((B, 1)) < 0.2).astype(np.int64))
    return batch, label
