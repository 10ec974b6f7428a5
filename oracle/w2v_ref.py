"""CPU restatement of model/word2vec_model.py (TEST INFRASTRUCTURE -- see oracle/__init__.py)."""
from __future__ import annotations

import re

import torch


def tokenize(sentences, word_to_token, max_words=32):
    """Word2VecTokenizer.__call__ (word2vec_model.py:33-73): lower-case, regex words, dictionary ids (0 = unknown),
    pad / cut to max_words.  Returns (ids [M, max_words] int64, mask [M, max_words] uint8)."""
    rows = []
    for s in sentences:
        words = re.findall(r"[\w']+", str(s).lower())[:max_words]
        ids = [word_to_token.get(w, 0) for w in words]
        rows.append(ids + [0] * (max_words - len(ids)))
    ids = torch.tensor(rows, dtype=torch.long)
    return ids, (ids != 0).to(torch.uint8)


def forward(p, input_ids, attention_mask=None):
    """Word2VecModel.forward (word2vec_model.py:83-102).  p: {'word_embd.weight', 'fc1.weight', 'fc1.bias', 'fc2.weight',
    'fc2.bias'}.  Returns {'last_hidden_state' [M,W,512], 'pooler_output' [M,512]}."""
    with torch.no_grad():
        x = p["word_embd.weight"][input_ids]
    h = torch.relu(x @ p["fc1.weight"].t() + p["fc1.bias"])
    if attention_mask is not None:
        keep = attention_mask.bool()
        keep = keep | (keep.sum(-1, keepdim=True) == 0)          # all-stop-word sentence: keep every position (:92-93)
        pooled = h.masked_fill(~keep[:, :, None], -6e4).max(dim=1).values
    else:
        pooled = h.max(dim=1).values
    return {"last_hidden_state": h @ p["fc2.weight"].t() + p["fc2.bias"],
            "pooler_output": pooled @ p["fc2.weight"].t() + p["fc2.bias"]}
