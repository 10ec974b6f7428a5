"""Activation workspaces of the encoder stacks: the saved tensors of one stack (`_EncRun`, ~1 GB at B=128) and the backward scratch
come from a per-shape pool -- allocating them afresh every step costs 13 ms of hipMalloc per GB on the host and made the step
host-bound.  `_WorkspaceMixin` is the pool (an LRU over shapes: real batches vary in N, hence in the joint length T + N)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

WIDTH, HEADS = 512, 8
# stacks of at most this many 64-row panels run the MLP branch through the split-hidden launches (the library reads the same variable)
SPLIT_PANELS = int(os.environ.get("TAN_SPLIT_PANELS", "48"))


class _Blocks:
    """Contiguous activation blocks carved out of one allocation."""

    def __init__(self, dtype, device, sizes: dict):
        self.off, total = {}, 0
        for k, n in sizes.items():
            self.off[k] = (total, n)
            total += (n + 63) // 64 * 64
        self.buf = torch.empty(total, dtype=dtype, device=device)

    def __getitem__(self, k):
        o, n = self.off[k]
        return self.buf[o:o + n]


class _EncRun:
    """Saved activations of one encoder stack + its ctypes descriptor."""

    def __init__(self, model, prefix, layers, B, L, cd, dev):
        R, Cw = B * L, WIDTH
        self.prefix, self.layers, self.B, self.L, self.R = prefix, layers, B, L, R
        per = {"xn1": R * Cw, "qkv": R * 3 * Cw, "attn_o": R * Cw, "x_mid": R * Cw, "xn2": R * Cw, "h_pre": R * 4 * Cw,
               "h_act": R * 4 * Cw, "x_out": R * Cw}
        self.act = _Blocks(cd, dev, {f"{i}.{k}": n for i in range(layers) for k, n in per.items()} | {"post": R * Cw})
        st = {"mean1": R, "rstd1": R, "mean2": R, "rstd2": R, "lse": B * HEADS * L}
        self.stat = _Blocks(torch.float32, dev, {f"{i}.{k}": n for i in range(layers) for k, n in st.items()}
                            | {"post_mean": R, "post_rstd": R})
        self.bufs = (_lib.LayerBufs * layers)()
        for i in range(layers):
            for k in per:
                setattr(self.bufs[i], k, self.act[f"{i}.{k}"].data_ptr())
            for k in st:
                setattr(self.bufs[i], k, self.stat[f"{i}.{k}"].data_ptr())
        # small stacks: scratch of the split-hidden MLP launches (tan_encoder_desc.split_part: eight f32 planes of partial sums)
        self.split_part = None
        if cd == torch.bfloat16 and R % 64 == 0 and R // 64 <= SPLIT_PANELS:
            self.split_part = torch.empty(8 * R * Cw, dtype=torch.float32, device=dev)

    def stage(self, s):
        """[R, C] deep-supervision output s (tfm_model.py:48-55)."""
        if s < self.layers - 1:
            return self.act[f"{s + 1}.xn1"].view(self.R, WIDTH)
        return self.act["post"].view(self.R, WIDTH)


class _EmbRun:
    """Buffers of the fused input-embedding launch (tan_embed_fwd) for one (B, T, N): what the stacks read and what the embeddings'
    backward needs, pooled like the stacks' workspaces (no per-step allocation)."""

    def __init__(self, B, T, N, Dv, Dt, cd, dev):
        R, Mp, L = B * T, B * N, T + N
        self.act = _Blocks(cd, dev, {"video_c": R * Dv, "lang_c": Mp * Dt, "proj_v": R * WIDTH, "proj_t": Mp * WIDTH,
                                     "x0": R * WIDTH, "xj": B * L * WIDTH, "lang_raw": Mp * WIDTH,
                                     "dproj_v": R * WIDTH, "dproj_t": Mp * WIDTH})       # (backward: LayerNorm-backward outputs)
        # backward: position-row gradient sums (dual, joint, text), one partial plane per group of videos
        self.nparts = -(-B // _lib.lib().tan_embed_bwd_group())
        self.dpos = torch.empty(3, self.nparts * max(T, N), WIDTH, device=dev)
        self.stat = _Blocks(torch.float32, dev, {"mean_v": R, "rstd_v": R, "mean_t": Mp, "rstd_t": Mp})
        self.keypad = torch.zeros(B, L, dtype=torch.uint8, device=dev)
        self._shape = {"video_c": (R, Dv), "lang_c": (Mp, Dt), "proj_v": (R, WIDTH), "proj_t": (Mp, WIDTH), "x0": (R, WIDTH),
                       "xj": (B * L, WIDTH), "lang_raw": (Mp, WIDTH), "dproj_v": (R, WIDTH), "dproj_t": (Mp, WIDTH)}

    def __getitem__(self, k):
        return self.act[k].view(self._shape[k]) if k in self._shape else self.stat[k]


class _WorkspaceMixin:
    """Pool state lives on the model: _ws_pool / _ws_lru / _ws_tick / _ws_lock (created in TemporalAligner.__init__)."""

    # Activation workspaces (~1 GB per stack at B=128) are pooled per shape: allocating them afresh every step costs
    # tens of ms of hipMalloc/hipFree on the host.  A workspace is taken at forward and handed back after backward
    # (or right after a no-grad forward, whose outputs never alias it).
    # Workspaces (saved activations of a stack, backward scratch) are pooled per shape: allocating ~1 GB afresh each step costs
    # 13 ms/GB of host time.  Real batches vary in N (hence in the joint length L = T + N), so the pool is an LRU over shapes
    # bounded to _WS_POOL_KEYS entries -- at most ~1.3 GB each at B=128.
    _WS_POOL_KEYS = 10

    def _pool_touch(self, key):
        with self._ws_lock:
            self._ws_tick += 1
            self._ws_lru[key] = self._ws_tick
            if len(self._ws_lru) > self._WS_POOL_KEYS:
                for old in sorted(self._ws_lru, key=self._ws_lru.get)[:len(self._ws_lru) - self._WS_POOL_KEYS]:
                    self._ws_lru.pop(old)
                    self._ws_pool.pop(old, None)          # tensors return to the caching allocator (stream-ordered reuse)

    def _take_ws(self, prefix, layers, B, L, cd, dev, alternate=False):
        """alternate: two workspaces per shape, handed out in turn (a pipelined training step: the previous step's last weight-gradient
        launches still read its activations while this step's first kernels write theirs)"""
        key = (prefix, layers, B, L, cd, dev)
        self._pool_touch(key)
        with self._ws_lock:
            pool = self._ws_pool.setdefault(key, [])
            if alternate:
                er = pool.pop(0) if len(pool) >= 2 else None
            else:
                er = pool.pop() if pool else None
        if er is None:
            er = _EncRun(self, prefix, layers, B, L, cd, dev)
        er.pool_key = key
        return er

    def _take_emb(self, B, T, N, Dv, Dt, cd, dev):
        key = ("emb", B, T, N, Dv, Dt, cd, dev)
        self._pool_touch(key)
        with self._ws_lock:
            pool = self._ws_pool.setdefault(key, [])
            em = pool.pop() if pool else None
        if em is None:
            em = _EmbRun(B, T, N, Dv, Dt, cd, dev)
        em.pool_key = key
        return em

    def _release_ws(self, er):
        if er is not None and getattr(er, "pool_key", None) is not None:
            with self._ws_lock:
                if er.pool_key in self._ws_lru:           # its shape may have been evicted meanwhile: then just drop it
                    pool = self._ws_pool.setdefault(er.pool_key, [])
                    if len(pool) < 2:
                        pool.append(er)
            er.pool_key = None

    def _take_scratch(self, R, cd, dev):
        key = ("scr", R, cd, dev)
        self._pool_touch(key)
        scr = self._ws_pool.get(key)
        if scr is None:
            scr = _Blocks(cd, dev, {"dx": R * WIDTH, "dx2": R * WIDTH, "do": R * WIDTH, "dxn": R * WIDTH,
                                    "dh": R * 4 * WIDTH, "dqkv": R * 3 * WIDTH,
                                    # second set of what a block's weight-gradient launch reads (tan_encoder_desc.dw_tail > 1)
                                    "dx_b": R * WIDTH, "dx2_b": R * WIDTH, "dh_b": R * 4 * WIDTH, "dqkv_b": R * 3 * WIDTH})
            n_ws = _lib.lib().tan_layernorm_bwd_ws_floats(C.c_int(WIDTH))
            scr.ln_ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
            scr.dw_ws = torch.empty(32 * 4 * WIDTH * WIDTH, dtype=torch.float32, device=dev)     # split-K partial tiles
            self._ws_pool[key] = scr
        return scr
