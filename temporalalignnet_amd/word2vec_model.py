"""Parameter surface of the reference's model/word2vec_model.py:Word2VecModel (frozen 66250x300 word embedding ->
fc1 300->2048, ReLU, max-pool over words -> fc2 2048->512), i.e. the `bert.*` / `lang_model.*` state_dict keys.

SURVEY.md section 8(f) row f1 ("next"): the sentence embedder is the input provider of the hot path -- BASELINE
configs feed random 512-d sentence embeddings -- so this module currently only carries the parameters for
checkpoint compatibility.  Calling it raises until row f1 is built on the HIP GEMM (gather + GEMM + masked max-pool).
"""
from __future__ import annotations

import torch
from torch import nn

from .tfm_model import _LinearParams


class Word2VecModel(nn.Module):
    def __init__(self, num_embeddings=66250, word_embedding_dim=300, embd_dim=512, hidden=2048):
        super().__init__()
        self.word_embd = nn.Embedding(num_embeddings, word_embedding_dim)
        self.word_embd.weight.requires_grad = False          # frozen in the reference (word2vec_model.py:84-85)
        self.fc1 = _LinearParams(word_embedding_dim, hidden)
        self.fc2 = _LinearParams(hidden, embd_dim)

    def forward(self, input_ids, attention_mask=None, **kw):
        raise NotImplementedError("Word2VecModel forward is SURVEY.md row f1 (next); feed sentence embeddings directly")
