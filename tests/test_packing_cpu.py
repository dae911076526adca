"""CPU tests of the packed token layout's host logic (mmdti_hip/packing.py, collate.packing_fields) and of the claim it rests
on, checked on the oracle: at dropout 0 every padded row of a molecule is the same row at the output of both towers, so ONE
representative row weighted by the number of padded positions reproduces the reference's unmasked InfoNCE mean
(models/infonce.py:32-33; SURVEY.md section 7 "Ragged batches").  The oracle's towers are pinned to the reference's own files by
the G6 / G9 fixtures (tests/test_oracle_golden.py, tests/test_g9_cpu.py)."""
import numpy as np
import pytest
import torch

from oracle import mmdti_oracle as O
from mmdti_hip.packing import PackedRows, right_padded_lengths
from mmdti_hip.collate import device_payload, packing_fields, HOST_FIELDS, to_device


def test_packed_rows_index_arithmetic():
    pk = PackedRows(torch.tensor([5, 9, 1, 9]), 9)
    assert (pk.B, pk.S, pk.M, pk.max_rows, pk.n_pad_rows) == (4, 9, 26, 9, 2)
    assert pk.off_host.tolist() == [0, 6, 15, 17, 26] and pk.rows_host.tolist() == [6, 9, 2, 9]
    # the representative pad row is the first padded slot of its sequence
    assert pk.gather_host.tolist() == [0, 1, 2, 3, 4, 5] + list(range(9, 18)) + [18, 19] + list(range(27, 36))
    assert pk.pad_weights().tolist() == [1] * 5 + [4] + [1] * 9 + [1, 8] + [1] * 9
    assert float(pk.pad_weights().sum()) == 4 * 9                      # every padded position is accounted for exactly once
    x = torch.arange(36).view(4, 9)
    pk.gather = pk.gather_host
    assert torch.equal(pk.unpack(pk.pack(x))[1], x[1]) and pk.unpack(pk.pack(x))[0].tolist() == [0, 1, 2, 3, 4, 5, 5, 5, 5]
    with pytest.raises(ValueError):
        PackedRows(torch.tensor([0, 3]), 5)
    with pytest.raises(ValueError):
        PackedRows(torch.tensor([6]), 5)


def test_right_padded_lengths_and_packing_fields():
    m = torch.tensor([[1, 1, 1, 0], [1, 1, 1, 1], [1, 0, 0, 0]])
    assert right_padded_lengths(m).tolist() == [3, 4, 1]
    assert right_padded_lengths(torch.tensor([[1, 0, 1, 0]])) is None            # a hole
    assert right_padded_lengths(torch.tensor([[0, 0, 0, 0]])) is None            # empty
    cfg = O.ModelCfg()
    batch, _ = O.synth_batch(5, 12, 16, cfg, seed=1, ragged=True)
    f = packing_fields(batch)
    assert f["packable"] and f["token_pad_id"] == 1 and f["token_counts"].tolist() == batch["attention_mask"].sum(1).tolist()
    pay = device_payload(batch)
    assert all(k in pay for k in HOST_FIELDS) and pay["atom_counts"].tolist() == batch["src_tokens"].ne(0).sum(1).tolist()
    # host fields stay on the host through to_device
    moved = to_device(pay, "cpu")
    assert moved["packable"] is True and moved["token_pad_id"] == 1
    # a pad hole among the atoms / a mask that is not a prefix / masked slots with different ids: not packable
    holes = dict(batch, src_tokens=batch["src_tokens"].clone())
    holes["src_tokens"][0, 1] = 0
    assert packing_fields(holes) == {"packable": False}
    bad = dict(batch, attention_mask=batch["attention_mask"].clone())
    row = int(batch["attention_mask"].sum(1).argmin())
    bad["attention_mask"][row, -1] = 1
    assert packing_fields(bad) == {"packable": False}
    ids = dict(batch, input_ids=batch["input_ids"].clone())
    ids["input_ids"][row, -1] = 5
    assert packing_fields(ids) == {"packable": False}
    full, _ = O.synth_batch(3, 6, 8, cfg, seed=2, ragged=False)
    assert packing_fields(full)["token_pad_id"] == -1 and packing_fields(full)["packable"]
    assert packing_fields({k: v for k, v in batch.items() if k != "input_ids"}) == {}


@pytest.mark.parametrize("task", ["classification"])
def test_oracle_pad_rows_are_one_row_and_the_weighted_mean_is_the_reference_mean(task):
    """The oracle (fp32 restatement of the reference's step) on a ragged batch at dropout 0: padded rows of encoder_rep / out_bert
    are identical within a molecule, and (sum_real + n_pad * x_pad) / S equals the unmasked mean over the padded tensor."""
    cfg = O.ModelCfg(unimol=O.UniMolCfg(layers=3, dim=64, ffn=128, heads=8, K=16, vocab=31, emb_dropout=0.0, dropout=0.0, attn_dropout=0.0, pooler_dropout=0.0),
                     roberta=O.RobertaCfg(layers=2, dim=64, heads=4, ffn=128, vocab=40, max_pos=40, hidden_dropout=0.0, attn_dropout=0.0),
                     cross=O.CrossCfg(dim=64, heads=4, ffn=128, hidden_dropout=0.0, attn_dropout=0.0), task=task, output_dim=2, infonce_dropout=0.0)
    P = O.init_params(cfg, seed=3, std=0.2)
    batch, label = O.synth_batch(6, 14, 18, cfg, seed=5, ragged=True)
    with torch.no_grad():
        out = O.mm_forward(batch, P, cfg, net_target=label, training=True)
    enc, bert = out["enc"], out["bert"]
    na, nt = batch["src_tokens"].ne(0).sum(1), batch["attention_mask"].sum(1)
    assert int(na.min()) < enc.shape[1] and int(nt.min()) < bert.shape[1]
    for x, n in ((enc, na), (bert, nt)):
        S = x.shape[1]
        for b in range(x.shape[0]):
            k = int(n[b])
            if k < S:
                assert float((x[b, k:] - x[b, k:k + 1]).abs().max()) <= 1e-6 * float(x[b, k].abs().max()), "padded rows differ"
        pk = PackedRows(n, S)
        pk.gather = pk.gather_host
        xp = pk.pack(x)
        w = pk.pad_weights().view(-1, 1)
        mean_packed = torch.zeros(x.shape[0], x.shape[2]).index_add_(0, pk.seq_host, xp * w) / S
        assert torch.allclose(mean_packed, x.mean(1), rtol=1e-5, atol=1e-6)
