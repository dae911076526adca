"""MI355X-native mirror of the reference's ``models/mm_model.py`` plugin boundary.

``MM_Model`` keeps the reference's constructor contract (``MM_Model(output_dim=..., **params)``), forward signature and
tuple-return protocol (/root/reference/models/mm_model.py:408-618), ``batch_collate_fn`` (:645-682), the attributes the
trainer reads (``fds_cfg.start_update``, ``FDS.*``, ``output_dim``) and the reference's ``state_dict()`` key names
(SURVEY.md Appendix A), so checkpoints move in both directions.  Every arithmetic step of ``forward`` runs in the
gfx950 kernels of libmmdti_hip.so through ``mmdti_hip.functional``:

    embed_tokens -> [gbf -> gbf_proj -> permute] -> 15-layer pair-bias encoder          (tower 1)
    RoBERTa embeddings + L post-LN layers                                                (tower 2)
    InfoNCE head; two co-attention blocks; masked pooling; FDS smoothing; head; ConR/SupCon

Differences by design, all opt-in or inert for the reference's own usage:
  * construction from explicit configs (``from_configs``) so synthetic / random-init models can be built without the
    pretrained ``weights/`` directory (no network in the build environment);
  * pretrained Uni-Mol weights load with a HARD failure on missing keys (the reference's ``strict=False`` at :514
    silently trains from random init on a key mismatch);
  * no ``CUDA_LAUNCH_BLOCKING`` and no hard-coded ``.cuda()``.
"""
from __future__ import annotations

import argparse
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..unicore_compat import Dictionary, init_bert_params, get_activation_fn
from ..functional import PairBiasFn, PairCompactFn, EmbeddingFn, DropoutFn, MaskedPoolFn, LinearF32Fn
from .. import ops
from ..collate import right_pad, collate_batch
from ..packing import PackedRows

PAIR_RAGGED = os.environ.get("MMDTI_PAIR_RAGGED", "1") != "0"
from .transformers import TransformerEncoderWithPair
from .bert_layers import BertCrossEncoder, RobertaTower
from .infonce import InfoNCE
from .fds import FDS

BACKBONE = {'transformer': TransformerEncoderWithPair}


def pad_1d_tokens(values, pad_idx):
    """utils/util.py:7-38."""
    return right_pad(values, pad_idx)


def pad_2d(values, pad_idx):
    """utils/util.py:41-72."""
    return right_pad(values, pad_idx, square=True)


def pad_coords(values, pad_idx):
    """utils/util.py:75-105."""
    return right_pad(values, pad_idx, tail=(3,))


class ClassificationHead(nn.Module):
    """mm_model.py:44-84: dropout -> dense -> tanh -> dropout -> out_proj."""

    def __init__(self, input_dim, inner_dim, num_classes, activation_fn, pooler_dropout):
        super().__init__()
        if activation_fn != "tanh":
            raise NotImplementedError("only pooler_activation_fn='tanh' is on the MM-DTI path")
        self.dense = nn.Linear(input_dim, inner_dim)
        self.activation_fn = get_activation_fn(activation_fn)
        self.dropout = nn.Dropout(p=pooler_dropout)
        self.out_proj = nn.Linear(inner_dim, num_classes)

    def forward(self, features, **kwargs):
        x = DropoutFn.apply(features, self.dropout.p, self.training)
        x = LinearF32Fn.apply(x, self.dense.weight, self.dense.bias, ops.ACT_TANH)
        x = DropoutFn.apply(x, self.dropout.p, self.training)
        return LinearF32Fn.apply(x, self.out_proj.weight, self.out_proj.bias, ops.ACT_NONE)


class NonLinearHead(nn.Module):
    """mm_model.py:86-128 (parameter container; executed fused inside PairBiasFn)."""

    def __init__(self, input_dim, out_dim, activation_fn, hidden=None):
        super().__init__()
        if activation_fn != "gelu":
            raise NotImplementedError("only activation_fn='gelu' is on the MM-DTI path")
        hidden = input_dim if not hidden else hidden
        self.linear1 = nn.Linear(input_dim, hidden)
        self.linear2 = nn.Linear(hidden, out_dim)


class GaussianLayer(nn.Module):
    """mm_model.py:226-269 (parameter container; executed inside PairBiasFn)."""

    def __init__(self, K=128, edge_types=1024):
        super().__init__()
        self.K = K
        self.means = nn.Embedding(1, K)
        self.stds = nn.Embedding(1, K)
        self.mul = nn.Embedding(edge_types, 1)
        self.bias = nn.Embedding(edge_types, 1)
        nn.init.uniform_(self.means.weight, 0, 3)
        nn.init.uniform_(self.stds.weight, 0, 3)
        nn.init.constant_(self.bias.weight, 0)
        nn.init.constant_(self.mul.weight, 1)


def molecule_architecture():
    """mm_model.py:325-343."""
    return SimpleNamespace(encoder_layers=15, encoder_embed_dim=512, encoder_ffn_embed_dim=2048, encoder_attention_heads=64,
                           dropout=0.1, emb_dropout=0.1, attention_dropout=0.1, activation_dropout=0.0, pooler_dropout=0.2,
                           max_seq_len=512, activation_fn="gelu", pooler_activation_fn="tanh", post_ln=False,
                           backbone="transformer", kernel="gaussian", delta_pair_repr_norm_loss=-1.0)


def fds_config():
    """mm_model.py:345-360."""
    return SimpleNamespace(feature_dim=512, bucket_num=20, bucket_start=0, start_update=0, start_smooth=1, kernel='gaussian',
                           ks=5, sigma=1, momentum=0.9, col_data="expt", raw_data="")


def crossmodal_config():
    """mm_model.py:362-377."""
    return SimpleNamespace(attention_probs_dropout_prob=0.2, gradient_checkpointing=False, hidden_act="gelu",
                           hidden_dropout_prob=0.3, hidden_size=512, initializer_range=0.02, intermediate_size=2048,
                           layer_norm_eps=1e-12, max_position_embeddings=512, num_attention_heads=16, num_hidden_layers=12,
                           position_embedding_type="absolute")


class CrossAttentionModel(nn.Module):
    """mm_model.py:379-406."""

    def __init__(self, cross_cfg, num_layers=1):
        super().__init__()
        self.text_attention = BertCrossEncoder(cross_cfg, num_layers)
        self.graph_attention = BertCrossEncoder(cross_cfg, num_layers)
        self.dropout = nn.Dropout(cross_cfg.hidden_dropout_prob)
        self.two_streams = os.environ.get("MMDTI_CROSS_TWO_STREAMS", "1") != "0"
        self._stream = None

    def forward(self, text_embeddings, graph_embeddings, text_mask, graph_mask, packs=None):
        """packs = (PackedRows of text_embeddings, PackedRows of graph_embeddings): both inputs (and both outputs) are packed rows
        [M, D]; the additive -10000 key masks are then implied -- a sequence's keys are its real rows (packing.py)."""
        text_embeddings = DropoutFn.apply(text_embeddings, self.dropout.p, self.training)
        graph_embeddings = DropoutFn.apply(graph_embeddings, self.dropout.p, self.training)
        if packs is not None:
            rev = (packs[1], packs[0])
            if self.two_streams and text_embeddings.is_cuda:
                main = torch.cuda.current_stream()
                if self._stream is None:
                    self._stream = torch.cuda.Stream()
                st = self._stream
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    text_to_graph = self.text_attention(text_embeddings, graph_embeddings, None, packs=packs)[-1]
                for t in (text_embeddings, graph_embeddings):
                    t.record_stream(st)
                graph_to_text = self.graph_attention(graph_embeddings, text_embeddings, None, packs=rev)[-1]
                main.wait_stream(st)
                text_to_graph.record_stream(main)
                return text_to_graph, graph_to_text
            graph_to_text = self.graph_attention(graph_embeddings, text_embeddings, None, packs=rev)[-1]
            text_to_graph = self.text_attention(text_embeddings, graph_embeddings, None, packs=packs)[-1]
            return text_to_graph, graph_to_text
        extended_txt_mask = (1.0 - text_mask.unsqueeze(1).unsqueeze(2).to(dtype=torch.float32)) * -10000.0
        extended_img_mask = (1.0 - graph_mask.unsqueeze(1).unsqueeze(2).to(dtype=torch.float32)) * -10000.0
        if self.two_streams and text_embeddings.is_cuda:
            # the two directions are independent one-layer blocks: the second runs on its own stream beside the first
            main = torch.cuda.current_stream()
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            st = self._stream
            st.wait_stream(main)
            with torch.cuda.stream(st):
                text_to_graph = self.text_attention(text_embeddings, graph_embeddings, extended_img_mask)[-1]
            for t in (text_embeddings, graph_embeddings, extended_img_mask):
                t.record_stream(st)
            graph_to_text = self.graph_attention(graph_embeddings, text_embeddings, extended_txt_mask)[-1]
            main.wait_stream(st)
            text_to_graph.record_stream(main)
            return text_to_graph, graph_to_text
        graph_to_text = self.graph_attention(graph_embeddings, text_embeddings, extended_txt_mask)[-1]
        text_to_graph = self.text_attention(text_embeddings, graph_embeddings, extended_img_mask)[-1]
        return text_to_graph, graph_to_text


class MM_Model(nn.Module):
    def __init__(self, output_dim=2, **params):
        super().__init__()
        self.cross_cfg = params.get('_cross_cfg') or crossmodal_config()
        self.fds_cfg = fds_config()
        self.args = params.get('_mol_args') or molecule_architecture()
        self.output_dim = output_dim
        self.data_type = 'molecule'
        self.remove_hs = params.get('remove_hs', False)
        self.use_fds = params.get('fds', False)
        self.using_scale = params.get('use_scaler', True)
        self.fds_num = params.get('fds_num', 30)
        self.fds_raw_path = params.get('fds_raw_path', '')
        self.fds_col_data = params.get('fds_col_data', '')
        self.ct_w = params.get('ct_w', 0.2)
        self.task = params.get('task', False)
        self.chemberta_dir = params.get('chemberta_dir', '')
        self.unimol_dir = params.get('unimol_dir', '')

        dictionary = params.get('_dictionary')
        if dictionary is None:
            dictionary = Dictionary.load(os.path.join(os.path.dirname(self.unimol_dir), 'mol.dict.txt'))
        self.dictionary = dictionary
        self.mask_idx = self.dictionary.add_symbol("[MASK]", is_special=True)
        self.padding_idx = self.dictionary.pad()
        a = self.args
        self.embed_tokens = nn.Embedding(len(self.dictionary), a.encoder_embed_dim, self.padding_idx)
        self.encoder = BACKBONE[a.backbone](
            encoder_layers=a.encoder_layers, embed_dim=a.encoder_embed_dim, ffn_embed_dim=a.encoder_ffn_embed_dim,
            attention_heads=a.encoder_attention_heads, emb_dropout=a.emb_dropout, dropout=a.dropout,
            attention_dropout=a.attention_dropout, activation_dropout=a.activation_dropout, max_seq_len=a.max_seq_len,
            activation_fn=a.activation_fn, no_final_head_layer_norm=a.delta_pair_repr_norm_loss < 0)
        K = params.get('_gbf_K', 128)
        n_edge_type = len(self.dictionary) * len(self.dictionary)
        self.gbf_proj = NonLinearHead(K, a.encoder_attention_heads, a.activation_fn)
        if a.kernel != 'gaussian':
            raise NotImplementedError("only kernel='gaussian' is on the MM-DTI path (mm_model.py:339)")
        self.gbf = GaussianLayer(K, n_edge_type)
        self.classification_head = ClassificationHead(input_dim=self.cross_cfg.hidden_size, inner_dim=a.encoder_embed_dim,
                                                      num_classes=self.output_dim, activation_fn=a.pooler_activation_fn,
                                                      pooler_dropout=a.pooler_dropout)
        self.apply(init_bert_params)
        if self.unimol_dir:
            self.load_pretrained_weights(path=self.unimol_dir)

        roberta_cfg = params.get('_roberta_cfg')
        if roberta_cfg is not None:
            self.bert = RobertaTower(roberta_cfg)
            self.bert.apply(init_bert_params)
            self.tokenizer = params.get('_tokenizer')
        else:
            self.bert = RobertaTower.from_pretrained(self.chemberta_dir)
            from transformers import AutoTokenizer          # preprocessing only (Rust tokenizer, unchanged by this build)
            self.tokenizer = AutoTokenizer.from_pretrained(self.chemberta_dir)
        if self.bert.cfg.dim != self.cross_cfg.hidden_size:
            raise ValueError(f"ChemBERTa hidden size {self.bert.cfg.dim} != cross-modal hidden size {self.cross_cfg.hidden_size}")

        self.cross_modal_module = CrossAttentionModel(self.cross_cfg, num_layers=1)
        if roberta_cfg is not None:
            self.cross_modal_module.apply(init_bert_params)

        # Select the contrastive loss (mm_model.py:481-491).  The reference leaves `CT` unbound for other tasks and
        # dies with UnboundLocalError; here that is an explicit error.
        if self.task == 'classification':
            from .contrastive import CT_Single as CT
        elif self.task == 'multilabel_classification':
            from .contrastive import CT_Multi as CT
        elif self.task == 'regression':
            from .contrastive import CT_Regress as CT
        else:
            raise ValueError(f"task={self.task!r}: only classification, multilabel_classification and regression are "
                             "constructible in the reference (mm_model.py:481-491)")
        self.CT = CT
        self.infonce = InfoNCE(self.cross_cfg.hidden_size, self.cross_cfg.hidden_size)
        if self.use_fds and self.task == 'regression':
            raw = params.get('_fds_raw_values', self.fds_raw_path)
            self.FDS = FDS(raw_data=raw, using_scale=self.using_scale, col_data=self.fds_col_data,
                           feature_dim=self.fds_cfg.feature_dim if '_mol_args' not in params else self.cross_cfg.hidden_size,
                           bucket_num=self.fds_num, bucket_start=self.fds_cfg.bucket_start, start_update=self.fds_cfg.start_update,
                           start_smooth=self.fds_cfg.start_smooth, kernel=self.fds_cfg.kernel, ks=self.fds_cfg.ks,
                           sigma=self.fds_cfg.sigma, momentum=self.fds_cfg.momentum)
        self.overlap_towers = bool(params.get('overlap_towers', True))
        self.infonce_on_side_stream = bool(params.get('infonce_on_side_stream', os.environ.get("MMDTI_INFONCE_SIDE", "1") != "0"))
        # Ragged batches can run on PACKED token rows -- real tokens plus ONE representative pad row per sequence, weighted by the padded
        # positions it stands for in the unmasked InfoNCE mean (packing.py).  That is the reference's computation exactly when the
        # padded rows of a sequence ARE one row, i.e. when no dropout is live; with dropout ON the reference draws an independent mask
        # per padded row, and one weighted row has n_pad times the variance in the pooled embedding (and, through F.normalize, not even
        # the same expectation) -- ADVICE r03.  strict_reference:
        #   None (default)  packed rows iff no dropout is live for this forward (eval / predict, or every p = 0), else the padded rows;
        #   True            always the padded rows (MMDTI_STRICT_REFERENCE=1);
        #   False           always packed rows when the batch allows (MMDTI_STRICT_REFERENCE=0): an opt-in trade of the padded rows'
        #                   dropout statistics for ~1.7 x on drug-like length distributions.
        env = os.environ.get("MMDTI_STRICT_REFERENCE")
        sr = params.get('strict_reference', None if env is None else env == "1")
        self.strict_reference = None if sr is None else bool(sr)
        self.last_layout = "padded"          # what the last forward ran on ("padded" | "packed"): read by bench.py / tests
        self._side = None
        self._pack_cache = None

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_configs(cls, output_dim, task, mol_args=None, roberta_cfg=None, cross_cfg=None, dictionary=None, gbf_K=128, **params):
        """Random-init model of the reference architecture from explicit configs (synthetic benchmarks / tests)."""
        return cls(output_dim=output_dim, task=task, _mol_args=mol_args or molecule_architecture(), _roberta_cfg=roberta_cfg,
                   _cross_cfg=cross_cfg or crossmodal_config(), _dictionary=dictionary or Dictionary.default_molecule(),
                   _gbf_K=gbf_K, **params)

    def load_pretrained_weights(self, path):
        """mm_model.py:499-514, but strict about the tower-1 keys (SURVEY.md section 7 'silent weight-name mismatch')."""
        if path is None:
            return
        state_dict = torch.load(path, map_location="cpu", weights_only=True)
        sd = state_dict['model'] if 'model' in state_dict else state_dict
        own = self.state_dict()
        tower1 = [k for k in own if k.startswith(("embed_tokens.", "encoder.", "gbf.", "gbf_proj."))]
        missing = [k for k in tower1 if k not in sd]
        if missing:
            raise RuntimeError(f"Uni-Mol checkpoint {path} lacks {len(missing)} tower-1 parameters, e.g. {missing[:5]}")
        self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)

    def pad_row_dropout_live(self):
        """True iff a padded row meets a live dropout site on its way to the only thing that reads it, the unmasked InfoNCE mean
        (infonce.py:32-33): tower 1 (embedding / residual / attention dropout), tower 2 (hidden / attention dropout), the InfoNCE
        head's query dropout.  (The fusion block and the pool never read padded rows.)"""
        if not self.training:
            return False
        ps = [self.encoder.emb_dropout, self.encoder.dropout, self.encoder.attention_dropout, self.bert.cfg.hidden_dropout, self.bert.cfg.attn_dropout,
              getattr(self.infonce, "embed_dropout", 0.0)]
        return max(float(p) for p in ps) > 0.0

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    # ------------------------------------------------------------------ forward
    def pair_bias(self, src_distance, src_edge_type, key_tiles_host=None, rows_host=None):
        """mm_model.py:553-556 fused: -> the pair bias ([B,H,N,ld] fp32, or the tiled pair layout on the hot path).  key_tiles_host
        (ragged batches): [B] real key tiles per molecule, on the HOST; rows_host (packed token rows): [B] query rows per molecule."""
        N = src_distance.shape[-1]
        return PairBiasFn.apply(self.gbf.means.weight, src_distance.float(), src_edge_type, self.gbf, self.gbf_proj, ops.pair_ld(N), key_tiles_host,
                                rows_host)

    def _packings(self, src_tokens, input_ids, atom_counts, token_counts, token_pad_id, packable):
        """-> (PackedRows of tower 1, PackedRows of tower 2) when this batch can and should run on packed token rows, else None.
        Decided on the HOST (the counts come from collate.device_payload): no device sync."""
        if self.strict_reference or not packable or atom_counts is None or token_counts is None or not src_tokens.is_cuda or not PAIR_RAGGED:
            return None
        if self.strict_reference is None and self.pad_row_dropout_live():
            return None                                   # (auto: the padded rows keep the reference's per-row dropout masks)
        if token_pad_id is not None and int(token_pad_id) not in (-1, int(self.bert.cfg.pad_idx)):
            return None                                   # masked SMILES slots do not hold the pad id: their rows are not one row
        (B, N), L = src_tokens.shape, input_ids.shape[1]
        if not (ops.PAIR_COMPACT and ops.pair_tiled_ok(N)):
            return None                                   # (the packed pair kernels exist for the compact tiled planes)
        c = self._pack_cache
        if c is not None and c[0] is atom_counts and c[1] is token_counts and c[2] == (B, N, L, src_tokens.device):
            return c[3]
        ac, tc = torch.as_tensor(atom_counts, device="cpu"), torch.as_tensor(token_counts, device="cpu")
        if ac.numel() != B or tc.numel() != B or int(ac.max()) > N or int(tc.max()) > L or int(ac.min()) < 1 or int(tc.min()) < 1:
            return None
        packs = None
        if int(ac.min()) < N or int(tc.min()) < L:         # (nothing padded: the padded layout IS the packed one)
            pk1, pk2 = PackedRows(ac, N), PackedRows(tc, L)
            D2, D1 = self.bert.cfg.dim, self.args.encoder_embed_dim
            hd_ok = all(ops.attn_eligible(pk1.max_rows, pk2.max_rows, D // h, D) for D, h in
                        ((D2, self.bert.cfg.heads), (self.cross_cfg.hidden_size, self.cross_cfg.num_attention_heads)))
            if hd_ok and max(pk1.max_rows, pk2.max_rows) <= 256 and D1 == self.cross_cfg.hidden_size:
                packs = (pk1.to(src_tokens.device), pk2.to(src_tokens.device))
        self._pack_cache = (atom_counts, token_counts, (B, N, L, src_tokens.device), packs)
        return packs

    def forward(self, src_tokens, src_distance, src_edge_type, input_ids, attention_mask, weights=None,
                return_infonce_loss=False, return_ct_loss=False, return_feature=False, net_target=None, use_weight=None,
                epoch=0, atom_counts=None, token_counts=None, token_pad_id=None, packable=False, **kwargs):
        padding_mask = src_tokens.eq(self.padding_idx)
        # Ragged batches.  atom_counts ([B] ints ON THE HOST: position of each molecule's last real token + 1, attached by
        # collate.device_payload) tells, without a device sync, whether some molecule is shorter than the padded length; then the
        # pair-attention kernels skip the all-padding key tiles (a third to a half of the pair traffic on a drug-like batch).
        # With token_counts / packable as well (both sides verified right-padded on the host) the whole step runs on PACKED token
        # rows: padded query rows are not computed either (packing.py).
        packs = self._packings(src_tokens, input_ids, atom_counts, token_counts, token_pad_id, packable)
        self.last_layout = "padded" if packs is None else "packed"
        key_tiles = kt_host = None
        if atom_counts is not None and PAIR_RAGGED and src_tokens.is_cuda:
            kt = (torch.as_tensor(atom_counts, device="cpu").to(torch.int64) + 15) // 16
            nt = (src_tokens.shape[1] + 15) // 16
            if int(kt.min()) < nt or packs is not None:
                kt = kt.clamp_(min=1, max=nt)
                eff = [ops.pair_key_tiles_effective(int(k), nt) for k in kt.tolist()]     # (what the kernels cover)
                if packs is None:
                    ops.set_pair_kept(sum(eff) / (kt.numel() * nt))
                else:      # packed rows: only the query blocks up to the representative pad row are walked
                    qb = ((packs[0].rows_host + 15) // 16).tolist()
                    ops.set_pair_kept(sum(q * e for q, e in zip(qb, eff)) / (kt.numel() * nt * nt))
                kt_host = kt
                key_tiles = ops.upload(kt.to(torch.int32), src_tokens.device)
        img_mask = ~padding_mask
        attention_mask = attention_mask.bool().to(src_tokens.device)
        # NOTE: the reference sets padding_mask=None when nothing is padded (:548-549), which costs a host sync
        # (`.any()`); the kernels treat an all-false mask identically, so no branch is needed here.
        pk1, pk2 = packs if packs is not None else (None, None)

        # The two towers are independent until InfoNCE: tower 2 runs on a side HIP stream so its kernels fill the CUs that
        # tower 1's tile tails and latency-bound pair kernels leave idle (autograd replays each backward on the stream of
        # its forward, so the overlap holds in the backward pass too).
        side = self._side_stream() if (self.overlap_towers and src_tokens.is_cuda) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out_bert = self.bert(input_ids, attention_mask, return_dict=True, pack=pk2)[0]

        if pk1 is None:
            xs = EmbeddingFn.apply(self.embed_tokens.weight, src_tokens, self.padding_idx)
            bias_s = self.pair_bias(src_distance, src_edge_type, kt_host)
            encoder_rep = self.encoder.encode(xs, bias_s, padding_mask, key_tiles)[0]
        else:
            # packed rows of tower 1: [M1] token ids / pad flags (integer plumbing; the representative pad row is the first padded
            # slot of its molecule, so it gathers the pad id), the pair bias stays positional
            ids1, pad1 = src_tokens.reshape(-1)[pk1.gather], padding_mask.reshape(-1)[pk1.gather]
            xs = EmbeddingFn.apply(self.embed_tokens.weight, ids1, self.padding_idx)
            bias_s = self.pair_bias(src_distance, src_edge_type, kt_host, pk1.rows_host)
            if bias_s.dtype != torch.float16:         # (head / basis counts the fused pair-bias kernel is not built for: re-lay out)
                bias_s = PairCompactFn.apply(bias_s, src_tokens.shape[1])
            encoder_rep = self.encoder.encode(xs, bias_s, pad1, key_tiles, pack=pk1)[0]

        if side is not None:
            main.wait_stream(side)
            out_bert.record_stream(main)
        else:
            out_bert = self.bert(input_ids, attention_mask, return_dict=True, pack=pk2)[0]

        infonce_side = False
        if return_infonce_loss:
            # The InfoNCE head and the cross-modal block both start from (encoder_rep, out_bert) and meet only in the loss: the
            # head -- two token-level GEMMs and a dozen latency-bound B x B kernels -- runs on the side stream under the
            # block's GEMMs (and its backward under the block's backward: autograd replays on the forward's stream).
            if side is not None and self.infonce_on_side_stream:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ct_loss = self.infonce(encoder_rep, out_bert, packs=packs)
                encoder_rep.record_stream(side)
                out_bert.record_stream(side)
                infonce_side = True
            else:
                ct_loss = self.infonce(encoder_rep, out_bert, packs=packs)

        cross_txt_output_layer, cross_output_layer = self.cross_modal_module(encoder_rep, out_bert, img_mask, attention_mask, packs=packs)
        # mm_model.py:572-576 (zero padded rows, concat, masked mean) in one kernel
        classification_feats_pooled = MaskedPoolFn.apply(cross_txt_output_layer, cross_output_layer, img_mask, attention_mask, packs)

        smoothed_features = classification_feats_pooled
        if self.training and epoch >= self.fds_cfg.start_smooth and self.use_fds and self.task == 'regression':
            smoothed_features = self.FDS.smooth(smoothed_features, net_target, epoch)
            classification_feats_pooled = smoothed_features      # the reference smooths IN PLACE (:579-581): CT sees it too

        logits = self.classification_head(smoothed_features)

        rnc_loss = None
        if return_ct_loss and net_target is not None:
            if use_weight:
                rnc_loss = self.CT(classification_feats_pooled, net_target, logits, weights=weights, w=self.ct_w)
            else:
                rnc_loss = self.CT(classification_feats_pooled, net_target, logits, w=self.ct_w)
        if infonce_side:
            main.wait_stream(side)
            ct_loss.record_stream(main)
        out = [logits]
        if return_feature:
            out.append(classification_feats_pooled)
        if return_infonce_loss:
            out.append(ct_loss)
        if rnc_loss is not None:
            out.append(rnc_loss)
        return out[0] if len(out) == 1 else tuple(out)

    # ------------------------------------------------------------------ collate (semantics of :645-682, pinned by G9)
    def batch_collate_fn(self, samples):
        """samples: list of (feature dict, label) as DataHub / TorchDataset yield them -> (batch dict, label tensor|None).
        Layout rules: ``mmdti_hip.collate``.  A key without a layout rule re-uses the previous field's value, as the
        reference's loop variable does (and raises if it comes first)."""
        return collate_batch(samples, self.padding_idx, self.tokenizer)
