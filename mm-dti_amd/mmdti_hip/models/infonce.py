"""Drop-in for the reference's ``models/infonce.py`` (InfoNCE module + info_nce function), MI355X-native.

Same class/function names, ctor arguments, child names (``info_proj_query.{0,2}``, ``info_proj_positive.{0,2}``) and
``ValueError`` behaviour as /root/reference/models/infonce.py:6-105.  Reference quirks kept: the per-token projections
are averaged over ALL sequence positions including padding (:32-33); dropout (p=0.1) is applied to the query side only
(:24); the explicit ``negative_keys`` branches can never return (the symmetric cross-entropy at :98 raises for
non-square logits) -- here they raise ``ValueError`` up front.

Extension for data parallelism (absent from the reference): ``set_global_negatives(gather, reduce_scatter, rank_row0)``
makes the B x B logits use every rank's keys.
"""
import torch
from torch import nn

from ..functional import InfoNCEFn, InfoNCELossFn


class InfoNCE(nn.Module):
    def __init__(self, bert_output_size, graph_ouput_size, temperature=0.1, reduction='mean', negative_mode='unpaired'):
        super().__init__()
        self.temperature = temperature
        self.reduction = reduction
        self.negative_mode = negative_mode
        self.orig_d_l = bert_output_size
        self.orig_d_av = graph_ouput_size
        self.d_l, self.d_av = 50, 50
        self.embed_dropout = 0.1
        self.training = True
        self.info_proj_query = nn.Sequential(nn.Linear(self.orig_d_l, self.orig_d_l), nn.GELU(), nn.Linear(self.orig_d_l, self.d_l))
        self.info_proj_positive = nn.Sequential(nn.Linear(self.orig_d_av, self.orig_d_av), nn.GELU(), nn.Linear(self.orig_d_av, self.d_av))
        self._gather = None
        self._reduce_scatter = None
        self._row0 = 0

    def set_global_negatives(self, gather, reduce_scatter, row0):
        self._gather, self._reduce_scatter, self._row0 = gather, reduce_scatter, int(row0)

    def forward(self, query, positive_key, negative_keys=None, packs=None):
        """packs = (PackedRows of query, PackedRows of positive_key): the inputs are packed rows [M, D] (packing.py); the
        unmasked mean of infonce.py:32-33 weights each representative pad row by the padded positions it stands for."""
        if packs is not None:
            if negative_keys is not None or self.reduction != 'mean' or query.dim() != 2 or positive_key.dim() != 2 or packs[0].B != packs[1].B:
                raise ValueError('packed rows: <query> / <positive_key> must be [rows, dim] with one packing each, implicit negatives.')
            return InfoNCEFn.apply(query.float(), positive_key.float(), self, self.training, self._gather, self._reduce_scatter, self._row0, packs)
        if negative_keys is not None:
            raise ValueError("explicit negative_keys are unreachable in the reference (infonce.py:98 raises); not supported")
        if self.reduction != 'mean':
            raise ValueError("only reduction='mean' is on the MM-DTI path")
        if self.orig_d_l == self.d_l or self.orig_d_av == self.d_av:
            raise ValueError("identity projection (input width == 50) is not on the MM-DTI path")
        if query.dim() != 3 or positive_key.dim() != 3:
            raise ValueError('<query> and <positive_key> must be [batch, seq, dim] token representations.')
        if len(query) != len(positive_key):
            raise ValueError('<query> and <positive_key> must must have the same number of samples.')
        return InfoNCEFn.apply(query.float(), positive_key.float(), self, self.training, self._gather, self._reduce_scatter, self._row0)


def info_nce(query, positive_key, negative_keys=None, temperature=0.1, reduction='mean', negative_mode='unpaired'):
    # Check input dimensionality (models/infonce.py:45-67).
    if query.dim() != 2:
        raise ValueError('<query> must have 2 dimensions.')
    if positive_key.dim() != 2:
        raise ValueError('<positive_key> must have 2 dimensions.')
    if negative_keys is not None:
        if negative_mode == 'unpaired' and negative_keys.dim() != 2:
            raise ValueError("<negative_keys> must have 2 dimensions if <negative_mode> == 'unpaired'.")
        if negative_mode == 'paired' and negative_keys.dim() != 3:
            raise ValueError("<negative_keys> must have 3 dimensions if <negative_mode> == 'paired'.")
    if len(query) != len(positive_key):
        raise ValueError('<query> and <positive_key> must must have the same number of samples.')
    if negative_keys is not None:
        if negative_mode == 'paired' and len(query) != len(negative_keys):
            raise ValueError("If negative_mode == 'paired', then <negative_keys> must have the same number of samples as <query>.")
    if query.shape[-1] != positive_key.shape[-1]:
        raise ValueError('Vectors of <query> and <positive_key> should have the same number of components.')
    if negative_keys is not None:
        if query.shape[-1] != negative_keys.shape[-1]:
            raise ValueError('Vectors of <query> and <negative_keys> should have the same number of components.')
        raise ValueError("explicit negative_keys: the reference's symmetric cross-entropy (infonce.py:98) raises for "
                         "non-square logits; this path is unreachable there and unsupported here")
    if reduction != 'mean':
        raise ValueError("only reduction='mean' is on the MM-DTI path")
    return InfoNCELossFn.apply(query.float(), positive_key.float(), float(temperature))
