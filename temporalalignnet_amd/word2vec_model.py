"""MI355X-native sentence embedder: the reference's model/word2vec_model.py (`Word2VecTokenizer`, `Word2VecModel`),
SURVEY.md section 8(f) row f1 -- the provider of the `[M, 512]` sentence embeddings the alignment hot path consumes
(train/main.py:58-60,172-175).

    frozen word2vec table [66250, 300]  ->  fc1 300->2048 + ReLU  ->  masked max over the <=32 words  ->  fc2 2048->512

State-dict keys are the reference's (`word_embd.weight`, `fc1.{weight,bias}`, `fc2.{weight,bias}`, i.e. `bert.*` /
`lang_model.*` inside the aligner).  Arithmetic runs in libtan_hip.so: `tan_embed_gather` (gather + pad 300->320 so the
contraction is a multiple of the MFMA K-step), `tan_gemm` with the ReLU epilogue, `tan_wordpool_fwd/bwd`, `tan_gemm` again;
backward returns gradients for fc1/fc2 (the table is frozen, word2vec_model.py:84-85).  No CPU fallback.

Differences from the reference, documented: `last_hidden_state` (= fc2 of every word, word2vec_model.py:100, never used by
train/main.py or the evaluation) is computed lazily on first access instead of on every call.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .tfm_model import _LinearParams

D_WORD, D_PAD, D_HID, D_OUT = 300, 320, 2048, 512


class Word2VecTokenizer:
    """Regex word split -> S3D dictionary ids, 0 for unknown words, padded / cut to `max_words` (word2vec_model.py:26-73).
    `vocab`: the `token_to_word` array of s3d_dict.npy (word i gets id i+1, s3dg.py:203-205) or a path to that file."""

    def __init__(self, max_words=32, vocab=None):
        if vocab is None:
            vocab = os.path.join(os.path.dirname(__file__), "s3d_dict.npy")
        if isinstance(vocab, (str, os.PathLike)):
            vocab = np.load(vocab)           # FileNotFoundError when the MIL-NCE assets are absent, like the reference
        self.word_to_token = {str(w): i + 1 for i, w in enumerate(vocab)}
        self.token_to_word = {v: k for k, v in self.word_to_token.items()}
        self.max_words = max_words

    def _split_sentence(self, sentence):
        return re.findall(r"[\w']+", str(sentence).lower())

    def _words_to_token(self, words):
        ids = [self.word_to_token.get(w, 0) for w in words[:self.max_words]]
        return ids + [0] * (self.max_words - len(ids))

    def tokenize(self, inputs):
        if isinstance(inputs, str):
            return self._split_sentence(inputs)
        return [self._split_sentence(i) for i in inputs]

    def __call__(self, inputs, padding=True, return_tensors=None, **kwargs):
        assert padding, f"padding = {padding} is not supported"
        if isinstance(inputs, str):
            tokens = self._words_to_token(self._split_sentence(inputs))
        else:
            tokens = [self._words_to_token(self._split_sentence(s)) for s in inputs]
        mask = (np.array(tokens) != 0).astype(np.uint8)
        if return_tensors == "pt":
            return {"input_ids": torch.from_numpy(np.array(tokens)), "attention_mask": torch.from_numpy(mask)}
        return {"input_ids": tokens, "attention_mask": mask.tolist()}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _SentenceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, mask_u8, w1, b1, w2, b2):
        cd, dev = model.compute_dtype, ids.device
        M, W = ids.shape
        R = M * W
        L = _lib.lib()
        x = torch.empty(R, D_PAD, dtype=cd, device=dev)
        _lib.check(L.tan_embed_gather(_p(ids), _p(model.word_embd.weight), _p(x), C.c_long(R), D_WORD, D_PAD,
                                      C.c_long(model.word_embd.weight.shape[0]), ops._dt(x), ops._stream()), "tan_embed_gather")
        w1p = torch.empty(D_HID, D_PAD, dtype=cd, device=dev)            # fc1 weight, K padded with zeros
        _lib.check(L.tan_embed_gather(None, _p(w1), _p(w1p), C.c_long(D_HID), D_WORD, D_PAD, C.c_long(D_HID), ops._dt(w1p),
                                      ops._stream()), "tan_embed_gather")
        w2c = w2 if cd == torch.float32 else ops.cast(w2.detach().contiguous(), torch.empty(D_OUT, D_HID, dtype=cd, device=dev))
        h = torch.empty(R, D_HID, dtype=cd, device=dev)
        ops.gemm(x, w1p, h, M=R, N=D_HID, K=D_PAD, bias=b1, act=_lib.ACT_RELU)
        pooled = torch.empty(M, D_HID, dtype=cd, device=dev)
        argmax = torch.empty(M, D_HID, dtype=torch.int32, device=dev)
        _lib.check(L.tan_wordpool_fwd(_p(h), _p(mask_u8), _p(pooled), _p(argmax), C.c_long(M), W, D_HID, ops._dt(h), ops._stream()),
                   "tan_wordpool_fwd")
        out = torch.empty(M, D_OUT, dtype=cd, device=dev)
        ops.gemm(pooled, w2c, out, M=M, N=D_OUT, K=D_HID, bias=b2)
        ctx.saved = (x, w2c, pooled, argmax, M, W)
        ctx.lazy = (h, w2c, b2)
        ctx.set_materialize_grads(False)
        return out.float() if cd != torch.float32 else out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        x, w2c, pooled, argmax, M, W = ctx.saved
        cd, dev = x.dtype, x.device
        R = M * W
        L = _lib.lib()
        g = g.contiguous().to(cd)
        g_b2 = torch.zeros(D_OUT, device=dev)
        ops.colsum_acc(g, g_b2, M, D_OUT)
        g_w2 = torch.zeros(D_OUT, D_HID, device=dev)
        ops.gemm(g, pooled, g_w2, M=D_OUT, N=D_HID, K=M, a_kc=False, b_kc=False, lda=D_OUT, ldb=D_HID, accumulate=True,
                 split_k=max(1, min(8, M // 256)))
        d_pooled = torch.empty(M, D_HID, dtype=cd, device=dev)
        ops.gemm(g, w2c, d_pooled, M=M, N=D_HID, K=D_OUT, a_kc=True, b_kc=False, ldb=D_HID)
        dh = torch.empty(R, D_HID, dtype=cd, device=dev)
        g_b1 = torch.zeros(D_HID, device=dev)
        _lib.check(L.tan_wordpool_bwd(_p(d_pooled), _p(pooled), _p(argmax), _p(dh), _p(g_b1), C.c_long(M), W, D_HID, ops._dt(dh),
                                      ops._stream()), "tan_wordpool_bwd")
        g_w1p = torch.zeros(D_HID, D_PAD, device=dev)
        ops.gemm(dh, x, g_w1p, M=D_HID, N=D_PAD, K=R, a_kc=False, b_kc=False, lda=D_HID, ldb=D_PAD, accumulate=True,
                 split_k=max(1, min(32, R // 512)))
        g_w1 = torch.zeros(D_HID, D_WORD, device=dev)
        _lib.check(L.tan_unpad_add(_p(g_w1p), _p(g_w1), C.c_long(D_HID), D_WORD, D_PAD, ops._stream()), "tan_unpad_add")
        return None, None, None, g_w1, g_b1, g_w2, g_b2


class _LazyOutputs(dict):
    """{'pooler_output': ...} plus 'last_hidden_state' = fc2(relu(fc1(x))) per word, built on first access."""

    def __init__(self, pooler, make_last):
        super().__init__(pooler_output=pooler)
        self._make_last = make_last

    def __missing__(self, key):
        if key == "last_hidden_state":
            self[key] = self._make_last()
            return self[key]
        raise KeyError(key)


class Word2VecModel(nn.Module):
    def __init__(self, num_embeddings=66250, compute_dtype="fp32"):
        super().__init__()
        self.word_embd = nn.Embedding(num_embeddings, D_WORD)
        self.word_embd.weight.requires_grad = False          # used under no_grad in the reference (word2vec_model.py:84-85)
        self.fc1 = _LinearParams(D_WORD, D_HID)
        self.fc2 = _LinearParams(D_HID, D_OUT)
        self.compute_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, torch.float32: torch.float32,
                              torch.bfloat16: torch.bfloat16}[compute_dtype]

    def forward(self, input_ids, attention_mask=None, *args, **kwargs):
        if not input_ids.is_cuda:
            raise _lib.TanHipError("Word2VecModel needs device tensors: the HIP path has no CPU fallback")
        ids = input_ids.long().contiguous()
        mask = None
        if attention_mask is not None:                        # 1 = keep; all-stop-word sentences keep everything (:92-93)
            mask = attention_mask.bool()
            mask = mask | (mask.sum(-1, keepdim=True) == 0)
            mask = mask.to(torch.uint8).contiguous()
        fn_ctx = {}
        pooler = _SentenceFn.apply(self, ids, mask, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)

        def make_last():
            with torch.no_grad():
                cd, dev = self.compute_dtype, ids.device
                M, W = ids.shape
                R = M * W
                L = _lib.lib()
                x = torch.empty(R, D_PAD, dtype=cd, device=dev)
                _lib.check(L.tan_embed_gather(_p(ids), _p(self.word_embd.weight), _p(x), C.c_long(R), D_WORD, D_PAD,
                                              C.c_long(self.word_embd.weight.shape[0]), ops._dt(x), ops._stream()), "tan_embed_gather")
                w1p = torch.empty(D_HID, D_PAD, dtype=cd, device=dev)
                _lib.check(L.tan_embed_gather(None, _p(self.fc1.weight), _p(w1p), C.c_long(D_HID), D_WORD, D_PAD, C.c_long(D_HID),
                                              ops._dt(w1p), ops._stream()), "tan_embed_gather")
                h = torch.empty(R, D_HID, dtype=cd, device=dev)
                ops.gemm(x, w1p, h, M=R, N=D_HID, K=D_PAD, bias=self.fc1.bias, act=_lib.ACT_RELU)
                w2c = self.fc2.weight if cd == torch.float32 else self.fc2.weight.detach().to(cd).contiguous()
                out = torch.empty(R, D_OUT, dtype=cd, device=dev)
                ops.gemm(h, w2c, out, M=R, N=D_OUT, K=D_HID, bias=self.fc2.bias)
                return out.float().view(M, W, D_OUT)

        del fn_ctx
        return _LazyOutputs(pooler, make_last)
