"""Graph-captured training step for TransformerLM -- the performance path.

One step = ref: src/train.py:143-151 (get_batch -> forward -> zero_grad -> backward -> AdamW.step)
run as hand-sequenced HIP kernels with no autograd, captured once into a hipGraph and replayed:
Python issues the step's 100 kernel launches (L = 6) once, at capture time, and one graph launch per step afterwards.

MI355X-first layout (288 GB HBM: keep everything resident, nothing is re-packed per step):
  * ONE flat fp32 master buffer holds every parameter; the model's nn.Parameters (reference
    state_dict layout, per-head key/query/value) are re-pointed to be VIEWS into it, with each
    layer's heads laid out as one packed [3C, C] QKV operand.  Regions:
        A  GEMM weights      (gradients written by ONE grouped dW launch; fp32 mode: S split-K slabs from the TN GEMMs)
        B  biases, LayerNorm (gradients arrive as G row-chunk partials)
        E  embeddings        (gradients written directly)
        Z  ln_f              (exists for the checkpoint; no gradient, no optimizer update)
  * flat gradient / Adam m / Adam v buffers with the same layout; one fused AdamW launch, which
    also refreshes the bf16 shadow copy of the weights and moves the step counter on; the W^T shadows (dX
    operands, one flat buffer) are refreshed by one batched transpose launch;
  * step counter, dropout seed, learning rate and the staged block of window offsets live in device
    memory, so a replay sees fresh dropout masks, the right bias correction, the scheduler's current lr
    and its own row of offsets (get_batch runs inside the embedding launch);
  * data parallel: the flat gradient is all-reduced (RCCL via torch.distributed) between the
    backward graph and the optimizer graph -- or, bucketed, range by range while the next layer group's
    backward graph runs (_dp_plan); ln_f is outside the reduced range on every rank.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from . import sublayers as S
from .model import TransformerLM

Tensor = torch.Tensor
ALIGN = 64          # floats: every tensor starts on a 256-byte boundary of the flat buffer


def _round(n: int, a: int = ALIGN) -> int:
    return (n + a - 1) // a * a


class _Layout:
    def __init__(self):
        self.entries: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.size = 0

    def add(self, key: str, shape) -> int:
        off = self.size
        n = 1
        for d in shape:
            n *= d
        self.entries[key] = (off, tuple(shape))
        self.size = off + _round(n)
        return off


def _os_env(name: str, default: str) -> str:
    import os
    return os.environ.get(name, default)


class ShadowWeights:
    """persistent GEMM operands: bf16 (or the fp32 master itself) for forward, W^T for dX"""

    def __init__(self):
        self.fwd_map: Dict[int, Tensor] = {}
        self.bwd_map: Dict[int, Tensor] = {}
        self.fwd8_map: Dict[int, Tuple[Tensor, Tensor]] = {}      # fp8 mode: (e4m3 copy, dequantisation scale [1])
        self.bwd8_map: Dict[int, Tuple[Tensor, Tensor]] = {}
        self.pack_map: Dict[int, Tensor] = {}                      # proj / second FFN Linear in dg_block_chain_fwd's streaming order
        self.packT_map: Dict[int, Tensor] = {}                     # the W^T operands in dg_block_chain_bwd's streaming order
        self.bwd_stale: set = set()                                # precision fp8: matrices whose bf16 W^T is no longer refreshed (only the e4m3 W^T is read)

    def fwd(self, W: Tensor) -> Tensor:
        return self.fwd_map[W.data_ptr()]

    def bwd(self, W: Tensor) -> Tensor:
        if W.data_ptr() in self.bwd_stale:
            raise RuntimeError("drakegpt_amd: the bf16 W^T of this matrix is not maintained in precision fp8 (its e4m3 W^T is)")
        return self.bwd_map[W.data_ptr()]

    def fwd8(self, W: Tensor):
        return self.fwd8_map[W.data_ptr()]

    def pack(self, W: Tensor) -> Optional[Tensor]:
        return self.pack_map.get(W.data_ptr())

    def packT(self, W: Tensor) -> Optional[Tensor]:
        return self.packT_map.get(W.data_ptr())

    def bwd8(self, W: Tensor):
        return self.bwd8_map[W.data_ptr()]


class FlatSink:
    def __init__(self, eng: "TrainEngine"):
        self.e = eng
        self.deferred = []           # (dY, X, dW view, P, Q): operands stay referenced until the grouped launch
        self.small_pending = []      # (slabs, n, splits, out): split-K partials of the small problems, reduced at the next flush

    def defer(self, key, dy, x, P, Q) -> bool:
        if not self.e.grouped_dw:
            return False
        if key in self.e.small_dw:
            return False                              # bucketed data-parallel step, tiny problem: own split-K launch (matrix())
        off, shape = self.e._region(key)              # GEMM weights, or the token table (one-hot dY)
        assert shape == (P, Q), (key, shape, P, Q)
        self.deferred.append((dy, x, self.e.gflat[off:off + P * Q], P, Q))
        return True

    def flush(self, group: int = 0):
        """one grouped launch for everything deferred so far (the whole backward pass, or one layer group of a bucketed
        data-parallel step: every group has its own problem set, hence its own workspace).  precision "fp8": the problems whose
        two operands exist as fp8 copies by now -- every block Linear: the e5m2 dY its dX GEMM consumed, the e4m3 X its forward
        GEMM consumed -- go into a second grouped launch on those copies (dg_gemm_tn_grouped, DG_FP8_E5M2)."""
        if self.deferred:
            bf, f8 = [], []
            for dy, x, view, P, Q in self.deferred:
                d8, x8 = getattr(dy, "dg_fp8", None), getattr(x, "dg_fp8x", None)
                if (self.e.fp8_dw and d8 is not None and x8 is not None and dy.shape[0] % 128 == 0 and d8[0].shape == (dy.shape[0], P)
                        and x8[0].shape == (x.shape[0], Q) and ops._ld(d8[0]) % 16 == 0 and ops._ld(x8[0]) % 16 == 0):
                    f8.append((d8[0], x8[0], view, P, Q, d8[1], x8[1]))
                else:
                    S._refuse_unwritten(dy)
                    S._refuse_unwritten(x)
                    bf.append((dy, x, view, P, Q))
            for kind, probs in ((0, bf), (1, f8)):
                if not probs:
                    continue
                ws = self.e.tn_workspaces.get((group, kind))
                if ws is None:                         # first step: sized for this problem set, zero-filled once
                    ws = self.e.tn_workspaces[(group, kind)] = ops.gemm_tn_grouped_workspace(probs, self.e.dev)
                ops.gemm_tn_grouped(probs, ws)
            self.deferred = []
        for slab, n, splits, out in self.small_pending:
            ops.reduce_partials(slab, n, splits, out, n)
        self.small_pending = []

    def matrix(self, key, P, Q):
        if key in self.e.small_dw:
            # a few-tile problem (lm_head / token table at a char-level vocabulary) inside a bucketed step: as one more problem
            # of a 126-tile group it would push the grouped launch past one round of half tiles, so it gets its own split-K
            # launch into a small slab buffer, summed into the flat gradient right away (deterministic order)
            slab, n = self.e.small_dw[key]
            goff, shape = self.e._region(key)
            assert shape == (P, Q), (key, shape, P, Q)
            self.small_pending.append((slab, P * Q, n, self.e.gflat[goff:goff + P * Q]))
            return slab, P * Q, n
        off, shape = self.e.layA.entries[key]
        assert shape == (P, Q), (key, shape, P, Q)
        # slabs beyond this matrix's own split count are never written: they stay zero from allocation
        n = S.splits_for_matrix(P, Q, self.e.M, self.e.S, self.e.dev)
        return self.e.slabs[0, off:off + P * Q], self.e.layA.size, n

    def vector(self, key, N):
        off, shape = self.e.layB.entries[key]
        assert shape == (N,), (key, shape, N)
        return self.e.vparts[0, off:off + N], self.e.layB.size, self.e.G

    def vector_rows(self, key, N, rows):
        off, shape = self.e.layB.entries[key]
        assert shape == (N,), (key, shape, N)
        if rows > self.e.vparts.shape[0]:
            return None
        # partial rows beyond `rows` are never written for this key: they stay zero from allocation
        return self.e.vparts[:rows, off:off + N]

    def direct(self, key, shape=None):
        return self.e.grad_view(key)


class TrainEngine:
    OFFSET_ROWS = 512          # rows of the staged window-offset block allocated up front (stage_offsets grows it on demand)

    def __init__(self, model: TransformerLM, batch_size: int, context_length: Optional[int] = None, *,
                 lr: float = 1e-3, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-2,
                 seed: int = 42, rank: int = 0, world_size: int = 1, process_group=None, use_graph: bool = True,
                 dp_buckets: Optional[int] = None, logits: str = "auto", grad_stream: str = "auto", fp8_dw: Optional[bool] = None):
        if not isinstance(model, TransformerLM):
            raise TypeError("TrainEngine drives TransformerLM (the other five models train through the autograd path)")
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise RuntimeError("TrainEngine needs the model on the GPU (model.to('cuda')); there is no CPU path")
        self.model = model
        self.dev = p0.device
        self.act = model.act_dtype
        self.fp8 = bool(getattr(model, "fp8", False))
        self.fp8_sites: Dict[str, Tensor] = {}
        self._fp8_seeded = False
        self.B = int(batch_size)
        self.T = int(context_length or model.context_length)
        if self.T > model.context_length:
            raise ValueError("context_length exceeds the model's position table")
        self.V, self.C = model.token_embedding_table.weight.shape
        self.L = len(model.blocks)
        blk0 = model.blocks[0]
        self.NH = len(blk0.sa_head.heads)
        self.H = blk0.sa_head.heads[0].head_size
        self.p_drop = float(model._p)
        self.M = self.B * self.T
        self.rank, self.world, self.pg = rank, world_size, process_group
        self.use_graph = use_graph
        self._dp_buckets_arg = dp_buckets
        import os as _os
        if logits == "auto" and _os.environ.get("DG_LOGITS") in ("fp32", "bf16"):      # A/B runs
            logits = _os.environ["DG_LOGITS"]
        if grad_stream == "auto" and _os.environ.get("DG_GRAD_STREAM") in ("fp32", "bf16"):
            grad_stream = _os.environ["DG_GRAD_STREAM"]
        if grad_stream not in ("auto", "fp32", "bf16"):
            raise ValueError("grad_stream must be 'auto', 'fp32' or 'bf16'")
        # The gradient that flows down the residual branch (dresid -> dx of every LayerNorm backward).  bf16 / fp8 modes keep it
        # in bf16 ("auto"): it is rounded once per sub-layer like every other activation gradient of those modes, and every
        # LayerNorm backward moves 75 MB instead of 100 MB.  The forward residual stream stays fp32 in every mode.
        _C = model.token_embedding_table.weight.shape[1]
        _can = self.act == torch.bfloat16 and ops.layernorm_bwd_fused_supported(_C) and self.M % 64 == 0
        if grad_stream == "bf16" and not _can:
            raise ValueError("a bf16 gradient stream needs the bf16 / fp8 precision and the fused LayerNorm backward")
        self.stream_dtype = torch.bfloat16 if (_can and grad_stream != "fp32") else torch.float32
        if logits not in ("auto", "fp32", "bf16"):
            raise ValueError("logits must be 'auto', 'fp32' or 'bf16'")
        # logits as bf16 (in-place gradient): by default only where they are big enough to matter -- the GPT-2 vocabulary --
        # and never in the fp32 parity mode; "fp32" keeps what the module path returns (tests compare the two)
        V_ = model.token_embedding_table.weight.shape[0]
        self.bf16_logits = self.act == torch.bfloat16 and 4096 < V_ <= 53248 and logits != "fp32" if logits != "bf16" else True
        if self.bf16_logits and (self.act != torch.bfloat16 or not (4096 < V_ <= 53248)):
            raise ValueError("bf16 logits need the bf16 / fp8 precision and a vocabulary of 4097 .. 53248 (the whole-row kernel)")
        g = S.granule(self.act)
        if self.C % g or (self.NH * self.H) % g:
            raise ValueError(f"embedding_dim must be a multiple of {g} for {self.act}")
        self.S = S.n_splits_for(self.M)
        self.G = S.n_partials_for(self.M)
        # bf16: every dW of the step comes from ONE grouped GEMM at the end of backward, written straight into the
        # flat gradient (no split-K slabs); fp32 parity mode keeps the per-matrix split-K path
        self.grouped_dw = self.act == torch.bfloat16 and self.M % 64 == 0
        # LayerNorm inside the epilogue of the GEMM that produces its input (dg_block_chain_fwd modes 3 / 4: proj + residual + LN2,
        # FFN2 + residual + the next block's LN1): bf16 mode at the width the kernel is built for.  OFF by default (DG_CHAIN_LN=1 turns
        # it on): measured inside the captured step (round 3, same box) proj + LN2 24.7 us against 17.4 + 8.8, FFN2 + LN1' 44.2 us
        # against 33.5 + 8.8, plus 7.8 us for the packed-weight refresh: 2.558 vs 2.528 ms per step
        # precision "fp8": the weight gradients of the block Linears on the fp8 copies of their operands (DG_FP8_DW=0: bf16 dW, A/B)
        self.last_block_act = _os.environ.get("DG_LAST_BLOCK_ACT", "1") != "0"      # 0: fp32 output + cast launch (A/B runs)
        self.fp8_head = False       # (set in _alloc_and_adopt: precision fp8 at a large vocabulary)
        self.fp8_dw = self.fp8 and (_os.environ.get("DG_FP8_DW", "1") != "0" if fp8_dw is None else bool(fp8_dw))
        # Everything between two attention calls as ONE launch per layer (dg_block_chain_fwd modes 2 / 0 / 1; forward launches per block
        # 7 -> 2): the default where the kernel exists (bf16, C = 384, M % 64 == 0); DG_CHAIN=0 keeps the separate launches (A/B).
        # Same box, headline configuration: 2.507 -> 2.403 ms per step.
        self.chain_full = (_os.environ.get("DG_CHAIN", "1") != "0" and not self.fp8 and self.NH * self.H == self.C and self.last_block_act
                           and ops.block_chain_supported(self.M, self.C, self.act))
        self.chain_warm = _os.environ.get("DG_CHAIN_WARM", "0") == "1"
        # The same for the backward pass (dg_block_chain_bwd: dX-QKV + LayerNorm-1 backward of block l, dX-FFN2 / dX-FFN1 / LayerNorm-2
        # backward / dX-proj of block l - 1 in one launch; backward launches per block 8 -> 3).  Needs the bf16 gradient stream and one
        # gradient exchange (a chain straddles two blocks: no layer-group seams).  Opt-in (DG_CHAIN_BWD=1) until it beats the separate launches
        # inside the step (first measurement: 136 vs 128.5 us per layer).
        self.chain_bwd = (_os.environ.get("DG_CHAIN_BWD", "0") == "1" and self.chain_full and self.stream_dtype == torch.bfloat16
                          and ops.block_chain_bwd_supported(self.M, self.C, self.act))
        self.chain_ln = (_os.environ.get("DG_CHAIN_LN", "0") == "1" and not self.fp8 and self.NH * self.H == self.C
                         and ops.block_chain_supported(self.M, self.C, self.act))
        self._build_layout()
        self.dp_buckets = self._choose_buckets(self._dp_buckets_arg)
        if self.dp_buckets > 1:
            self.chain_bwd = False
        self._alloc_and_adopt()
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay], dtype=torch.float32, device=self.dev)
        # dropout stream differs per data-parallel rank; the step word also drives Adam's bias correction
        self.state = ops.new_rng_state(seed + 0x9E3779B97F4A7C15 * rank & 0xFFFFFFFFFFFFFFFF, self.dev, 0)
        # window offsets: a staged block [rows, B] the captured step walks through by itself (row = step word - off_ctl[0],
        # clamped to off_ctl[1] rows); set_offsets() is the one-row form (row 0, off_ctl[1] = 1)
        self.off_block = torch.zeros((self.OFFSET_ROWS, self.B), dtype=torch.int64, device=self.dev)
        self.offsets = self.off_block[0]
        self.off_ctl = torch.tensor([0, 1], dtype=torch.int32, device=self.dev)
        self._off_rows, self._off_left = 1, None
        self.x = torch.zeros((self.B, self.T), dtype=torch.int64, device=self.dev)
        self.y = torch.zeros((self.B, self.T), dtype=torch.int64, device=self.dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=self.dev)
        # fp8 head: dequantisation factor of the e5m2 dlogits the loss kernel writes with the a-priori scale 57344 M (|dlogits| <= 1 / M)
        self.dl_scale = torch.full((1,), 1.0 / (self.M * 57344.0), dtype=torch.float32, device=self.dev)
        self.loss_scratch = torch.zeros((2048 + 1,), dtype=torch.float32, device=self.dev)     # fused loss head: shares + arrival counter
        self.corpus: Optional[Tensor] = None
        self._graphs = None
        self._eval_graph = None
        self.force_dp_path = False      # rehearsal hook (tools/dp_rccl_smoke.py): take the multi-rank path with one rank
        self.debug_timing = False       # multi-rank step only: HIP events around backward graph(s) / exchange / optimizer graph -> last_timing (ms)
        self.last_timing: Optional[dict] = None
        self.keep_logits = False        # parity tests: keep the step's logits [M, V] alive as `last_logits` (also inside a captured graph)
        self.last_logits: Optional[Tensor] = None
        self.refresh_shadows()

    # -------------------------------------------------------------------------------- layout
    def _build_layout(self):
        C, V, L, NH, H = self.C, self.V, self.L, self.NH, self.H
        A, Bv, E, Z = _Layout(), _Layout(), _Layout(), _Layout()
        for l in range(L):
            A.add(f"{l}.wqkv", (3 * NH * H, C))
            A.add(f"{l}.wproj", (C, NH * H))
            A.add(f"{l}.w1", (4 * C, C))
            A.add(f"{l}.w2", (C, 4 * C))
        A.add("lm.w", (V, C))
        for l in range(L):
            for k, n in (("bproj", C), ("b1", 4 * C), ("b2", C), ("ln1w", C), ("ln1b", C), ("ln2w", C), ("ln2b", C)):
                Bv.add(f"{l}.{k}", (n,))
        Bv.add("lm.b", (V,))
        E.add("tok", (V, C))
        E.add("pos", (self.model.context_length, C))
        Z.add("lnf.w", (C,))
        Z.add("lnf.b", (C,))
        self.layA, self.layB, self.layE, self.layZ = A, Bv, E, Z
        self.offA, self.offB = 0, A.size
        self.offE = A.size + Bv.size
        self.n_active = self.offE + E.size           # optimizer / all-reduce range
        self.offZ = self.n_active
        self.n_total = self.n_active + Z.size

    # -------------------------------------------------------------------------------- data-parallel plan
    def _choose_buckets(self, arg: Optional[int]) -> int:
        """How many layer groups the backward pass is cut into for the gradient exchange (1 = one all-reduce of the whole flat
        gradient between the backward graph and the optimizer graph).  Bucketing overlaps the exchange of a finished group with
        the backward pass of the next one; it needs the grouped dW GEMM (bf16 mode).  Default (None, or DG_DP_BUCKETS): bucket
        when the gradient is large enough for the exchange to matter against the step -- >= 128 MB, i.e. the GPT-2 shapes
        (652 MB / 1.6 GB) -- and not for the 43 MB of the scaled model, whose whole exchange is ~0.4 ms over xGMI while every
        cut costs a graph seam and a less well filled dW launch."""
        import os
        if arg is None and os.environ.get("DG_DP_BUCKETS"):
            arg = int(os.environ["DG_DP_BUCKETS"])
        if not self.grouped_dw or (self.world == 1 and arg is None):
            return 1
        if arg is None:
            arg = 4 if self.n_active * 4 >= (128 << 20) else 1
        return max(1, min(int(arg), self.L))

    def _dp_plan(self):
        """[(layers of the group, in backward order; [(lo, hi) ranges of the flat gradient that are final after the group])].
        Region A is laid out layer 0 .. L-1 then lm_head, so a group's weight gradients are ONE contiguous range; the first group
        also carries lm_head.weight (its dW problem is the first one recorded), the last group the biases / LayerNorm vectors
        and the embeddings (B | E: contiguous, final only after the last LayerNorm backward and the embedding backward)."""
        nb, L = self.dp_buckets, self.L
        per = (L + nb - 1) // nb
        plan = []
        hi_layer = L
        while hi_layer > 0:
            lo_layer = max(0, hi_layer - per)
            lo = self.offA + self.layA.entries[f"{lo_layer}.wqkv"][0]
            hi = self.offA + (self.layA.size if hi_layer == L else self.layA.entries[f"{hi_layer}.wqkv"][0])
            ranges = [(lo, hi)]
            if lo_layer == 0:
                ranges.append((self.offB, self.n_active))
            plan.append((list(range(hi_layer - 1, lo_layer - 1, -1)), ranges))
            hi_layer = lo_layer
        return plan

    def _region(self, key: str):
        for lay, base in ((self.layA, self.offA), (self.layB, self.offB), (self.layE, self.offE), (self.layZ, self.offZ)):
            if key in lay.entries:
                off, shape = lay.entries[key]
                return base + off, shape
        raise KeyError(key)

    def param_view(self, key: str) -> Tensor:
        off, shape = self._region(key)
        n = 1
        for d in shape:
            n *= d
        return self.flat[off:off + n].view(shape)

    def grad_view(self, key: str) -> Tensor:
        off, shape = self._region(key)
        n = 1
        for d in shape:
            n *= d
        return self.gflat[off:off + n].view(shape)

    def named_grads(self) -> Dict[str, Tensor]:
        """the step's gradient as views of the flat buffer, keyed by the reference's parameter names (what `p.grad` holds after
        `loss.backward()` in ref: src/train.py:150; after a data-parallel step: the SUM over ranks).  `ln_f.*` is absent: it never
        receives a gradient."""
        NH, H = self.NH, self.H
        out: Dict[str, Tensor] = {}
        for l in range(self.L):
            wqkv = self.grad_view(f"{l}.wqkv")
            pre = f"blocks.{l}."
            for h in range(NH):
                out[f"{pre}sa_head.heads.{h}.key.weight"] = wqkv[(NH + h) * H:(NH + h + 1) * H]
                out[f"{pre}sa_head.heads.{h}.query.weight"] = wqkv[h * H:(h + 1) * H]
                out[f"{pre}sa_head.heads.{h}.value.weight"] = wqkv[(2 * NH + h) * H:(2 * NH + h + 1) * H]
            for ref, key in (("sa_head.proj.weight", "wproj"), ("sa_head.proj.bias", "bproj"), ("ffwd.net.0.weight", "w1"),
                             ("ffwd.net.0.bias", "b1"), ("ffwd.net.2.weight", "w2"), ("ffwd.net.2.bias", "b2"),
                             ("ln1.weight", "ln1w"), ("ln1.bias", "ln1b"), ("ln2.weight", "ln2w"), ("ln2.bias", "ln2b")):
                out[pre + ref] = self.grad_view(f"{l}.{key}")
        out["lm_head.weight"], out["lm_head.bias"] = self.grad_view("lm.w"), self.grad_view("lm.b")
        out["token_embedding_table.weight"] = self.grad_view("tok")
        out["position_embedding_table.weight"] = self.grad_view("pos")
        return out

    def _alloc_and_adopt(self):
        """copy the model's current weights into the flat buffer and re-point its Parameters at it"""
        dev, m = self.dev, self.model
        self.flat = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.m_ = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.v_ = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.tn_workspaces: Dict[int, Tensor] = {}      # split-K workspace of the grouped dW GEMM, one per launch group
        self.small_dw: Dict[str, Tuple[Tensor, int]] = {}
        # one-hot rows of the batch (bf16 [M, V rounded up to 8]): rewritten by every forward, read by the grouped dW GEMM
        self.onehot = None
        if self.grouped_dw and ops.layernorm_bwd_fused_supported(self.C) and (self.C // 4) * 8 >= S.pad_to(self.V, 8):
            self.onehot = torch.zeros((self.M, S.pad_to(self.V, 8)), dtype=torch.bfloat16, device=dev)
        if self.dp_buckets > 1:
            # bucketed step: few-tile problems leave the grouped launches (see FlatSink.matrix)
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            for key, P, Q in (("lm.w", self.V, self.C),) + ((("tok", self.V, self.C),) if self.onehot is not None else ()):
                if ((P + 255) // 256) * ((Q + 127) // 128) <= 4:
                    n = max(1, min(64, self.M // 256, ncu // (((P + 127) // 128) * ((Q + 127) // 128))))
                    self.small_dw[key] = (torch.zeros((n, P * Q), dtype=torch.float32, device=dev), n)
        self.slabs = None if self.grouped_dw else torch.zeros((self.S, self.layA.size), dtype=torch.float32, device=dev)
        # partial rows of the bias / LayerNorm gradients: G row chunks, or as many as the GEMM epilogue that emits the column
        # sums (FeedForward's first bias) asks for; rows a producer never writes stay zero
        self.Gv = max(self.G, ops.gemm_nt_colsum_rows(self.act, self.M, 4 * self.C, self.C))
        if self.chain_bwd:
            self.Gv = max(self.Gv, 2 * (self.M // 64))            # dg_block_chain_bwd: two partial rows per 64-row block
        self.vparts = torch.zeros((self.Gv, self.layB.size), dtype=torch.float32, device=dev)
        NH, H = self.NH, self.H

        def adopt(param: torch.nn.Parameter, view: Tensor):
            view.copy_(param.data)
            param.data = view

        with torch.no_grad():
            for l, blk in enumerate(m.blocks):
                wqkv = self.param_view(f"{l}.wqkv")
                for h, head in enumerate(blk.sa_head.heads):
                    adopt(head.query.weight, wqkv[h * H:(h + 1) * H])
                    adopt(head.key.weight, wqkv[(NH + h) * H:(NH + h + 1) * H])
                    adopt(head.value.weight, wqkv[(2 * NH + h) * H:(2 * NH + h + 1) * H])
                adopt(blk.sa_head.proj.weight, self.param_view(f"{l}.wproj"))
                adopt(blk.sa_head.proj.bias, self.param_view(f"{l}.bproj"))
                adopt(blk.ffwd.net[0].weight, self.param_view(f"{l}.w1"))
                adopt(blk.ffwd.net[0].bias, self.param_view(f"{l}.b1"))
                adopt(blk.ffwd.net[2].weight, self.param_view(f"{l}.w2"))
                adopt(blk.ffwd.net[2].bias, self.param_view(f"{l}.b2"))
                adopt(blk.ln1.weight, self.param_view(f"{l}.ln1w"))
                adopt(blk.ln1.bias, self.param_view(f"{l}.ln1b"))
                adopt(blk.ln2.weight, self.param_view(f"{l}.ln2w"))
                adopt(blk.ln2.bias, self.param_view(f"{l}.ln2b"))
            adopt(m.lm_head.weight, self.param_view("lm.w"))
            adopt(m.lm_head.bias, self.param_view("lm.b"))
            adopt(m.token_embedding_table.weight, self.param_view("tok"))
            adopt(m.position_embedding_table.weight, self.param_view("pos"))
            adopt(m.ln_f.weight, self.param_view("lnf.w"))
            adopt(m.ln_f.bias, self.param_view("lnf.b"))
        # GEMM operand shadows
        self.weights = ShadowWeights()
        self.shadow = None
        if self.act == torch.bfloat16:
            self.shadow = torch.zeros(self.n_active, dtype=torch.bfloat16, device=dev)
        self._mats: List[Tuple[Tensor, Tensor]] = []
        # every W^T lives in ONE flat buffer (like the forward shadows), so that the fp8 mode can quantise all of them with one
        # segmented launch pair
        wt_sizes = {key: shape[1] * S.k_pad(shape[0], self.act) for key, (off, shape) in self.layA.entries.items()}
        self.wt_flat = torch.zeros(sum(_round(n) for n in wt_sizes.values()), dtype=self.act, device=dev)
        seg_f, seg_b, wt_off = [], [], 0
        if self.fp8:
            self.shadow8 = torch.zeros(self.layA.size, dtype=S.E4M3, device=dev)
            self.wt8_flat = torch.zeros(self.wt_flat.numel(), dtype=S.E4M3, device=dev)
            # lm_head joins when its two contraction lengths suit the fp8 K step: K = C forward, K = V padded (50257 -> 50304 = 393 x
            # 128) in the dX direction; a char-level vocabulary (80 -> 128 columns of padding) stays bf16
            self.fp8_head = (_os_env("DG_FP8_HEAD", "1") != "0" and self.bf16_logits and S.fp8_k_ok(self.C) and S.fp8_k_ok(S.k_pad(self.V, self.act))
                             and self.last_block_act)
            n_fp8 = sum(1 for key in self.layA.entries if key != "lm.w" or self.fp8_head)
            self.wscale_f = torch.ones(n_fp8, dtype=torch.float32, device=dev)
            self.wscale_b = torch.ones(n_fp8, dtype=torch.float32, device=dev)
        self.wpack_flat = torch.zeros(self.layA.size, dtype=torch.bfloat16, device=dev) if (self.chain_ln or self.chain_full) else None
        self.wtpack_flat = torch.zeros(self.layA.size, dtype=torch.bfloat16, device=dev) if self.chain_bwd else None
        self._pack_pairs = []
        self._packT_pairs = []
        self._u8_pairs = []
        self.fp8_wt8 = self.fp8 and _os_env("DG_FP8_WT8", "1") != "0"
        for key, (off, shape) in self.layA.entries.items():
            W = self.param_view(key)
            n = shape[0] * shape[1]
            if (self.chain_ln and (key.endswith(".wproj") or key.endswith(".w2"))) or (self.chain_full and key != "lm.w"):
                pk = self.wpack_flat[off:off + n].view(shape)
                self.weights.pack_map[W.data_ptr()] = pk
                self._pack_pairs.append((self.shadow[self.offA + off:self.offA + off + n].view(shape), pk))
            if self.shadow is not None:
                self.weights.fwd_map[W.data_ptr()] = self.shadow[self.offA + off:self.offA + off + n].view(shape)
            else:
                self.weights.fwd_map[W.data_ptr()] = W
            wt_shape = (shape[1], S.k_pad(shape[0], self.act))
            Wt = self.wt_flat[wt_off:wt_off + wt_sizes[key]].view(wt_shape)
            self.weights.bwd_map[W.data_ptr()] = Wt
            self._mats.append((W, Wt))
            if self.chain_bwd and key != "lm.w":
                pkT = self.wtpack_flat[off:off + n].view(wt_shape)
                self.weights.packT_map[W.data_ptr()] = pkT
                self._packT_pairs.append((Wt, pkT))
            if self.fp8 and (key != "lm.w" or self.fp8_head):
                i = len(seg_f)
                seg_f.append([off, n])
                seg_b.append([wt_off, wt_sizes[key]])
                self.weights.fwd8_map[W.data_ptr()] = (self.shadow8[off:off + n].view(shape), self.wscale_f[i:i + 1])
                # round 3: the e4m3 W^T is the byte transposition of the e4m3 W (same values, same per-matrix scale) -- the bf16 W^T of
                # these matrices is neither refreshed nor read (DG_FP8_WT8=0: bf16 transposition + a second cast, as before)
                w8t = self.wt8_flat[wt_off:wt_off + wt_sizes[key]].view(wt_shape)
                if self.fp8_wt8:
                    self.weights.bwd8_map[W.data_ptr()] = (w8t, self.wscale_f[i:i + 1])
                    if key != "lm.w":               # (lm_head keeps its bf16 W^T too: the keep_logits / fp32-logits paths take the bf16 dX GEMM)
                        self.weights.bwd_stale.add(W.data_ptr())
                    self._u8_pairs.append((self.shadow8[off:off + n].view(shape), w8t))
                else:
                    self.weights.bwd8_map[W.data_ptr()] = (w8t, self.wscale_b[i:i + 1])
            wt_off += _round(wt_sizes[key])
        if self.fp8:
            self.seg_f = torch.tensor(seg_f, dtype=torch.int64, device=dev)
            self.seg_b = torch.tensor(seg_b, dtype=torch.int64, device=dev)
            self.w_amax = torch.zeros(len(seg_f) * ops.FP8_AMAX_PARTS, dtype=torch.float32, device=dev)

    def refresh_shadows(self):
        """bf16 copy of the GEMM weights + every W^T.  Call after the weights change outside step()
        (load_state_dict, manual edits); step() keeps them current by itself."""
        if self.shadow is not None:
            ops.cast(self.flat[:self.layA.size], torch.bfloat16, out=self.shadow[:self.layA.size])
        self._refresh_transposes()

    def _refresh_transposes(self):
        # bf16: transpose the bf16 shadow the optimizer has just written (bit-identical to casting the fp32 master, half the read)
        from_shadow = self.shadow is not None
        if getattr(self, "_tr_table", None) is None:
            pairs = [(self.weights.fwd(W) if from_shadow else W, Wt) for W, Wt in self._mats if W.data_ptr() not in self.weights.bwd_stale]
            self._tr_table = ops.make_transpose_table(pairs, self.dev) if pairs else ()
        if self._tr_table:
            ops.transpose_cast_batched(*self._tr_table, self.act, in_dtype=torch.bfloat16 if from_shadow else torch.float32)
        if self._pack_pairs:
            if getattr(self, "_pack_table", None) is None:
                self._pack_table = ops.make_pack_table(self._pack_pairs + self._packT_pairs, self.dev)
            ops.pack_chain_weights_batched(*self._pack_table)
        if self.fp8:
            # e4m3 copies of every block matrix and of every W^T, per-matrix scales: two launch pairs for the whole model
            n = self.seg_f.shape[0]
            ops.fp8_quantize(self.shadow[self.offA:self.offA + self.layA.size], S.E4M3, seg=self.seg_f, n_seg=n, out=self.shadow8,
                             scale_inv=self.wscale_f, amax=self.w_amax)
            if self.fp8_wt8:
                if getattr(self, "_u8_table", None) is None:
                    self._u8_table = ops.make_transpose_u8_table(self._u8_pairs, self.dev)
                ops.transpose_u8_batched(*self._u8_table)
            else:
                ops.fp8_quantize(self.wt_flat, S.E4M3, seg=self.seg_b, n_seg=n, out=self.wt8_flat, scale_inv=self.wscale_b, amax=self.w_amax,
                                 reuse_amax=True)          # W^T holds W's values: same per-matrix maxima, no second amax pass

    # -------------------------------------------------------------------------------- programs
    def _layer_params(self, l: int):
        pv = self.param_view
        return dict(ln1w=pv(f"{l}.ln1w"), ln1b=pv(f"{l}.ln1b"), wqkv=pv(f"{l}.wqkv"), wproj=pv(f"{l}.wproj"),
                    bproj=pv(f"{l}.bproj"), ln2w=pv(f"{l}.ln2w"), ln2b=pv(f"{l}.ln2b"), w1=pv(f"{l}.w1"), b1=pv(f"{l}.b1"),
                    w2=pv(f"{l}.w2"), b2=pv(f"{l}.b2"))

    def _forward(self, run: S.Run, x_idx: Tensor, y_idx: Optional[Tensor], want_grad: bool, gather: bool = False):
        B, T = x_idx.shape
        M = B * T
        p = self.p_drop
        # backward gets the token-table gradient as one more problem of the grouped dW GEMM: one-hot(idx)^T dx
        onehot = self.onehot if (want_grad and self.onehot is not None and M == self.M) else None
        if gather:       # get_batch inside the embedding launch: ids / targets land in self.x / self.y (= x_idx / y_idx)
            h = ops.batch_embed_fwd(self.corpus, self.off_block, self.state, self.off_ctl, x_idx, y_idx, self.param_view("tok"),
                                    self.param_view("pos"), onehot=onehot).view(M, self.C)
        else:
            h = ops.embed_fwd(x_idx, self.param_view("tok"), self.param_view("pos"), onehot=onehot).view(M, self.C)
        if self.chain_full and ops.block_chain_supported(M, self.C, self.act):
            h, saved = self._blocks_chain(run, h, B, T, want_grad)
            return self._head(run, h, y_idx, want_grad, saved, M)
        saved = []
        pre = None          # LayerNorm output for the next sub-layer, when the GEMM in front of it produced it (chain_ln)
        for l in range(self.L):
            P = self._layer_params(l)
            nxt = [] if self.chain_ln else None
            h, sa = S.attn_fwd(run, h, P["ln1w"], P["ln1b"], P["wqkv"], P["wproj"], P["bproj"], True, B, T, self.NH, self.H, p, p, l,
                               pre=pre, fuse_ln=(P["ln2w"], P["ln2b"]) if self.chain_ln else None, nxt=nxt)
            pre = nxt[0] if nxt else None
            last = l == self.L - 1
            nxt = [] if (self.chain_ln and not last) else None
            Pn = self._layer_params(l + 1) if nxt is not None else None
            h, sf = S.ffn_fwd(run, h, P["ln2w"], P["ln2b"], P["w1"], P["b1"], P["w2"], P["b2"], True, p, l,
                              out_dtype=self.act if (last and self.last_block_act) else torch.float32,
                              pre=pre, fuse_ln=(Pn["ln1w"], Pn["ln1b"]) if nxt is not None else None, nxt=nxt)
            pre = nxt[0] if nxt else None
            if want_grad:
                saved.append((sa, sf))
        return self._head(run, h, y_idx, want_grad, saved, M)

    def _blocks_chain(self, run: S.Run, h: Tensor, B: int, T: int, want_grad: bool):
        """the residual blocks with everything between two attention calls in ONE launch (dg_block_chain_fwd): head (LayerNorm 1 +
        QKV of block 0), then per block attention + chain (proj .. the next block's QKV; the last block stops behind its second
        residual add and hands lm_head a bf16 tensor).  Leaves exactly the tensors the separate launches leave for backward."""
        M, C = h.shape
        p = run.p(self.p_drop)
        pk = self.weights.pack
        saved = []
        P = self._layer_params(0)
        r = ops.block_chain_fwd(2, M, C, x=h, ln1w=P["ln1w"], ln1b=P["ln1b"], wqkv=pk(P["wqkv"]))
        x, h1, m1, r1, qkv = h, r["h1"], r["mean1"], r["rstd1"], r["qkv"]
        for l in range(self.L):
            o, lse = ops.attn_fwd(qkv, B, T, self.NH, self.H, self.H ** -0.5, p, run.rng, S.site_attn(l), keep=want_grad)
            last = l == self.L - 1
            if self.chain_warm:
                # the layer's packed weight stream (wproj | w1 | w2 | the next block's wqkv: contiguous in the packed buffer)
                lo = self.layA.entries[f"{l}.wproj"][0]
                hi = self.layA.entries[f"{l + 1}.wproj"][0] if not last else self.layA.entries["lm.w"][0]
                ops.l2_warm(self.wpack_flat[lo:hi])
            kw = dict(o=o, x=x, wproj=pk(P["wproj"]), bproj=P["bproj"], ln2w=P["ln2w"], ln2b=P["ln2b"], w1=pk(P["w1"]), b1=P["b1"],
                      w2=pk(P["w2"]), b2=P["b2"], dropout_p=p, rng_state=run.rng, site_proj=S.site_proj(l), site_ffn=S.site_ffn(l))
            if last:
                r = ops.block_chain_fwd(1, M, C, **kw)
            else:
                Pn = self._layer_params(l + 1)
                r = ops.block_chain_fwd(0, M, C, ln1w=Pn["ln1w"], ln1b=Pn["ln1b"], wqkv=pk(Pn["wqkv"]), **kw)
            if want_grad:
                saved.append(((x, h1, m1, r1, qkv, o, lse), (r["x1"], r["h2"], r["mean2"], r["rstd2"], r["f"], r["bits"])))
            x = r["x2"]
            if not last:
                P, h1, m1, r1, qkv = Pn, r["h1"], r["mean1"], r["rstd1"], r["qkv"]
        return x, saved

    def _head(self, run: S.Run, h: Tensor, y_idx: Optional[Tensor], want_grad: bool, saved, M: int):
        if self.bf16_logits and y_idx is not None and not self.keep_logits:
            # large vocabulary: lm_head writes bf16 logits into the buffer that becomes dlogits -- the cross-entropy kernel holds
            # a whole row in registers and overwrites it in place with its gradient (1.65 GB less written and 0.82 GB less read
            # per step at the GPT-2 vocabulary, M = 8192, than fp32 logits + a separate bf16 gradient)
            buf = torch.empty((M, S.k_pad(self.V, self.act)), dtype=self.act, device=self.dev)
            if self.fp8_head and run.fp8 and h.dtype == torch.bfloat16:
                # precision fp8 (round 3): lm_head's three contractions on the fp8 MFMA too.  Forward on e4m3 x e4m3; the loss kernel
                # leaves its gradient a second time as e5m2 (a-priori scale: |dlogits| <= 1 / M) -- operand of the dX GEMM and of the
                # weight gradient (the bf16 gradient in `buf` still feeds the bias column sums)
                Wl = self.param_view("lm.w")
                xq, xs = S._quantize_operand(run, h, S.E4M3, "lm.x")
                h.dg_fp8x = (xq, xs)
                wq, ws = self.weights.fwd8(Wl)
                logits = ops.gemm_nt(xq, wq, self.act, scale_a=xs, scale_b=ws, bias=self.param_view("lm.b"), out=buf[:, :self.V])
                if want_grad and M == self.M:
                    q8 = torch.empty((M, buf.shape[1]), dtype=S.E5M2, device=self.dev)
                    rows = ops.cross_entropy_fp8(logits, y_idx.view(M), self.V, buf, 1.0 / M, q8)
                    buf.dg_q8 = (q8, self.dl_scale)
                else:
                    rows = ops.cross_entropy(logits, y_idx.view(M), self.V, dlogits=buf if want_grad else None, grad_scale=1.0 / M)
                return None, rows, (saved, h, buf)
            logits, (xa,) = S.linear_fwd(run, h, self.param_view("lm.w"), self.param_view("lm.b"), out=buf[:, :self.V])
            rows = ops.cross_entropy(logits, y_idx.view(M), self.V, dlogits=buf if want_grad else None, grad_scale=1.0 / M)
            return None, rows, (saved, xa, buf)
        logits, (xa,) = S.linear_fwd(run, h, self.param_view("lm.w"), self.param_view("lm.b"), pad_rows=True)
        if y_idx is None:
            return logits, None, None
        dlogits = None
        if want_grad:
            dlogits = torch.empty((M, S.k_pad(self.V, self.act)), dtype=self.act, device=self.dev)
            if M == self.M and ops.cross_entropy_fused_supported(logits, dlogits, self.G):
                # small vocabulary: the loss head in one launch -- gradient rows, the lm_head bias partials and the mean loss
                part, stride, n = FlatSink(self).vector("lm.b", self.V)
                rows = ops.cross_entropy_fused(logits, y_idx.view(M), self.V, dlogits, 1.0 / M, part, stride, n, self.loss_scratch,
                                               self.loss, 1.0 / M)
                return logits, rows, (saved, xa, dlogits, True)
        rows = ops.cross_entropy(logits, y_idx.view(M), self.V, dlogits=dlogits, grad_scale=1.0 / M)
        return logits, rows, (saved, xa, dlogits)

    def _backward_begin(self, run: S.Run, x_idx: Tensor, ctx) -> dict:
        saved, xa, dlogits = ctx[:3]
        head_done = len(ctx) > 3            # the fused loss head already left the lm_head bias partials (and the loss) behind
        st = dict(run=run, x_idx=x_idx, saved=saved, sink=FlatSink(self), g_next=None, g0=None)
        g = dlogits[:, :self.V]
        q8 = getattr(dlogits, "dg_q8", None)
        if q8 is not None:
            g.dg_fp8 = (q8[0][:, :self.V], q8[1])          # the e5m2 copy the loss kernel left (fp8 head)
        st["dh"] = S.linear_bwd_from_act(run, (xa,), g, self.param_view("lm.w"), True, st["sink"], {"w": "lm.w", "b": "lm.b"},
                                         bias_done=head_done)
        return st

    def _backward_layers(self, st: dict, layers) -> None:
        """backward of the given blocks (descending).  st["g_next"]: dropout-backward of dh for the sub-layer that runs next,
        fused into the LayerNorm backward that produced dh"""
        run, sink, saved, x_idx = st["run"], st["sink"], st["saved"], st["x_idx"]
        B, T = x_idx.shape
        p = self.p_drop
        dh, g_next = st["dh"], st["g_next"]
        for l in layers:
            P = self._layer_params(l)
            sa, sf = saved[l]
            dh, g_next = S.ffn_bwd(run, sf, dh, P["ln2w"], P["w1"], P["w2"], True, p, l, sink,
                                   {"w1": f"{l}.w1", "b1": f"{l}.b1", "w2": f"{l}.w2", "b2": f"{l}.b2", "ln_w": f"{l}.ln2w", "ln_b": f"{l}.ln2b"},
                                   g_in=g_next, emit=(p, S.site_proj(l), f"{l}.bproj", self.C, f"{l}.g_proj"))
            keys = {"wqkv": f"{l}.wqkv", "wproj": f"{l}.wproj", "bproj": f"{l}.bproj", "ln_w": f"{l}.ln1w", "ln_b": f"{l}.ln1b"}
            if l > 0:
                dh, g_next = S.attn_bwd(run, sa, dh, P["ln1w"], P["wqkv"], P["wproj"], True, B, T, self.NH, self.H, p, p, l, sink, keys,
                                        g_in=g_next, emit=(p, S.site_ffn(l - 1), f"{l - 1}.b2", self.C, f"{l - 1}.g_ffn"))
            elif self.onehot is not None:
                # first block: also take dx in bf16 (no dropout, no bias behind it) -- the X operand of the token-table problem
                dh, st["g0"] = S.attn_bwd(run, sa, dh, P["ln1w"], P["wqkv"], P["wproj"], True, B, T, self.NH, self.H, p, p, l, sink, keys,
                                          g_in=g_next, emit=(0.0, 0, None, self.C))
            elif self.stream_dtype != torch.float32:
                # (bf16 gradient stream: the fused LayerNorm backward is the only form that writes it; its g output is unused)
                dh, _ = S.attn_bwd(run, sa, dh, P["ln1w"], P["wqkv"], P["wproj"], True, B, T, self.NH, self.H, p, p, l, sink, keys,
                                   g_in=g_next, emit=(0.0, 0, None, self.C))
            else:
                dh = S.attn_bwd(run, sa, dh, P["ln1w"], P["wqkv"], P["wproj"], True, B, T, self.NH, self.H, p, p, l, sink, keys, g_in=g_next)
        st["dh"], st["g_next"] = dh, g_next

    def _backward_layers_chain(self, st: dict) -> None:
        """backward of all residual blocks with everything between two attention-backward calls in ONE launch (dg_block_chain_bwd):
        the top block's second half (mode 1), per block boundary the first half of block l + the second half of block l - 1 (mode 0),
        block 0's first half (mode 2).  Records the same (dY, X) operand pairs for the grouped dW launch and writes the same partial
        rows of the bias / LayerNorm gradients (two per 64-row block) as the separate launches."""
        run, sink, saved, x_idx = st["run"], st["sink"], st["saved"], st["x_idx"]
        B, T = x_idx.shape
        M, C, L = self.M, self.C, self.L
        p = run.p(self.p_drop)
        pkT = self.weights.packT
        stride = self.layB.size

        def vec(key, n):
            return sink.vector(key, n)[0]

        def second(l):
            P = self._layer_params(l)
            x1, h2, mean2, rstd2, f, bits = saved[l][1]
            return dict(w2T=pkT(P["w2"]), bits=bits, db1_part=vec(f"{l}.b1", 4 * C), w1T=pkT(P["w1"]), x1=x1, mean2=mean2, rstd2=rstd2,
                        ln2w=P["ln2w"], dln2w_part=vec(f"{l}.ln2w", C), dln2b_part=vec(f"{l}.ln2b", C), gbias2_part=vec(f"{l}.bproj", C),
                        wprojT=pkT(P["wproj"]), site_proj=S.site_proj(l))

        def first(l, dqkv, dresid):
            P = self._layer_params(l)
            x, h1, m1, r1, qkv, o, lse = saved[l][0]
            return dict(dqkv=dqkv, wqkvT=pkT(P["wqkv"]), x=x, mean1=m1, rstd1=r1, ln1w=P["ln1w"], dresid1=dresid,
                        dln1w_part=vec(f"{l}.ln1w", C), dln1b_part=vec(f"{l}.ln1b", C),
                        gbias1_part=vec(f"{l - 1}.b2", C) if l > 0 else None, site_ffn_below=S.site_ffn(l - 1) if l > 0 else 0)

        def weights_and_attention(l, g, r):
            """the four weight-gradient problems of block l and its attention backward; returns dqkv"""
            x, h1, m1, r1, qkv, o, lse = saved[l][0]
            x1, h2, mean2, rstd2, f, bits = saved[l][1]
            S.weight_grad(sink, f"{l}.w2", g, f, C, 4 * C)
            S.weight_grad(sink, f"{l}.w1", r["df"], h2, 4 * C, C)
            S.weight_grad(sink, f"{l}.wproj", r["g2"], o, C, C)
            dqkv = ops.attn_bwd(qkv, o, r["dout"], lse, B, T, self.NH, self.H, self.H ** -0.5, p, run.rng, S.site_attn(l))
            S.weight_grad(sink, f"{l}.wqkv", dqkv, h1, 3 * C, C)
            return dqkv

        kw = dict(part_stride=stride, dropout_p=p, rng_state=run.rng)
        dh = st["dh"]
        l = L - 1
        part, pstride, n = sink.vector(f"{l}.b2", C)
        g = ops.dropout_bwd_cast(dh, run.act, p, run.rng, S.site_ffn(l), colsum_part=part, part_stride=pstride, n_partials=n)
        r = ops.block_chain_bwd(1, M, C, g_in=g, dresid2=dh, **second(l), **kw)
        while True:
            dqkv = weights_and_attention(l, g, r)
            if l == 0:
                break
            r = ops.block_chain_bwd(0, M, C, **first(l, dqkv, r["dx2"]), **second(l - 1), **kw)
            g = r["g1"]
            l -= 1
        r = ops.block_chain_bwd(2, M, C, **first(0, dqkv, r["dx2"]), **kw)
        st["dh"], st["g0"], st["g_next"] = r["dx1"], r["g1"], None

    def _backward_end(self, st: dict, group: int = 0) -> None:
        x_idx, dh, sink = st["x_idx"], st["dh"], st["sink"]
        B, T = x_idx.shape
        if self.onehot is not None:
            S.weight_grad(sink, "tok", self.onehot[:, :self.V], st["g0"], self.V, self.C)
            ops.embed_bwd(x_idx, dh.view(B, T, self.C), None, self.grad_view("pos")[:T], V=self.V)
        else:
            ops.embed_bwd(x_idx, dh.view(B, T, self.C), self.grad_view("tok"), self.grad_view("pos")[:T])
        if self.grouped_dw:
            sink.flush(group)
        else:
            ops.reduce_partials(self.slabs, self.layA.size, self.S, self.gflat[self.offA:], self.layA.size)
        ops.reduce_partials(self.vparts, self.layB.size, self.Gv, self.gflat[self.offB:], self.layB.size)

    def _backward(self, run: S.Run, x_idx: Tensor, ctx):
        st = self._backward_begin(run, x_idx, ctx)
        if self.chain_bwd and x_idx.numel() == self.M:
            self._backward_layers_chain(st)
        else:
            self._backward_layers(st, reversed(range(self.L)))
        self._backward_end(st)

    def _train_run(self) -> S.Run:
        """per-step runtime configuration of the training program.  fp8: every quantisation site keeps its amax history in
        self.fp8_sites (one-pass delayed scaling); the first execution (the eager warm-up before capture, or the first eager
        step) quantises just in time and seeds the history"""
        seed = self.fp8 and not self._fp8_seeded
        if seed:
            self._fp8_seeded = True
        return S.Run(act=self.act, rng=self.state if self.p_drop > 0.0 else None, weights=self.weights, fp8=self.fp8,
                     fp8_sites=self.fp8_sites if self.fp8 else None, fp8_seed=seed, step_word=self.state, stream=self.stream_dtype,
                     fp8_only=self.fp8 and self.fp8_dw and self.grouped_dw and _os_env("DG_FP8_ONLY", "1") != "0")

    def _prog_fwd_bwd(self):
        """gather the batch, forward, backward, reduce the gradient partials"""
        run = self._train_run()
        logits, rows, ctx = self._forward(run, self.x, self.y, True, gather=self.corpus is not None)
        if self.keep_logits:
            self.last_logits = logits
        if len(ctx) == 3:
            ops.reduce_sum(rows, 1.0 / self.M, out=self.loss)
        self._backward(run, self.x, ctx)

    def _prog_segments(self):
        """the same step cut at the layer-group boundaries of _dp_plan(): segment k ends with the grouped dW launch of its
        group, after which the group's range of the flat gradient is final and its all-reduce can start"""
        plan = self._dp_plan()
        st = {}

        def first():
            run = self._train_run()
            logits, rows, ctx = self._forward(run, self.x, self.y, True, gather=self.corpus is not None)
            if self.keep_logits:
                self.last_logits = logits
            if len(ctx) == 3:
                ops.reduce_sum(rows, 1.0 / self.M, out=self.loss)
            st.clear()
            st.update(self._backward_begin(run, self.x, ctx))

        segs = []
        for k, (layers, _) in enumerate(plan):
            def seg(k=k, layers=layers):
                if k == 0:
                    first()
                self._backward_layers(st, layers)
                if k == len(plan) - 1:
                    self._backward_end(st, group=k)
                    st.clear()                      # release the activations (they live in the graph pool anyway)
                else:
                    st["sink"].flush(k)
            segs.append(seg)
        return segs, [r for _, r in plan]

    def _prog_update(self):
        # (the step word moves on inside the AdamW launch: nothing after it reads the word)
        ops.adamw_step(self.flat, self.gflat, self.m_, self.v_, self.hyper, self.state, 1.0 / self.world,
                       shadow_bf16=self.shadow, n=self.n_active, advance=True)
        self._refresh_transposes()

    def _dp(self) -> bool:
        return self.world > 1 or self.force_dp_path

    def _allreduce(self):
        if self._dp():
            import torch.distributed as dist
            dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=self.pg)     # mean = sum * 1/world in AdamW

    def _allreduce_ranges_async(self, ranges, works: list) -> None:
        """start the SUM all-reduce of finished ranges of the flat gradient.  RCCL enqueues it on its own stream behind the work
        already submitted to the current stream (the segment that produced the ranges) and it then runs beside the next
        segment; `works` are joined before the optimizer graph."""
        if self._dp():
            import torch.distributed as dist
            for lo, hi in ranges:
                works.append(dist.all_reduce(self.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    # -------------------------------------------------------------------------------- capture
    def _capture(self):
        snap = (self.flat.clone(), self.m_.clone(), self.v_.clone(), self.state.clone())
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up: loads every code object and allocates the dW workspaces before capture
            if self._dp() and self.dp_buckets > 1:
                for seg in self._prog_segments()[0]:
                    seg()
            else:
                self._prog_fwd_bwd()
            self._prog_update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.flat.copy_(snap[0]); self.m_.copy_(snap[1]); self.v_.copy_(snap[2]); self.state.copy_(snap[3])
        self.refresh_shadows()
        torch.cuda.synchronize(self.dev)
        if not self._dp():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._prog_fwd_bwd()
                self._prog_update()
            self._graphs = (g,)
        elif self.dp_buckets > 1:
            # one graph per layer group + the optimizer graph, all in one memory pool (activations cross the seams)
            segs, self._seg_ranges = self._prog_segments()
            graphs = []
            for seg in segs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None):
                    seg()
                graphs.append(g)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=graphs[0].pool()):
                self._prog_update()
            self._graphs = tuple(graphs) + (g,)
        else:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._prog_fwd_bwd()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._prog_update()
            self._graphs = (g1, g2)

    # -------------------------------------------------------------------------------- public
    def set_corpus(self, data: Tensor):
        """keep the token stream resident in HBM (int64, as train_data.pt stores it)"""
        if data.dim() != 1 or data.numel() < self.T + 1:
            raise ValueError(f"set_corpus: need a 1-D token stream of more than context_length = {self.T} tokens")
        self.corpus = data.to(self.dev, dtype=torch.int64).contiguous()
        ops.check_ids(self.corpus, self.V, "the corpus")          # once, here: the captured step never checks
        self._graphs = None

    def set_lr(self, lr: float):
        self.hyper[0:1].fill_(float(lr))

    def set_offsets(self, ix: Tensor):
        """window offsets of THIS rank's rows for the next step (drawn by the host CPU generator, ref: preprocessing.py:43)"""
        if self._off_rows != 1:
            self.off_ctl[1:2].fill_(1)
            self._off_rows, self._off_left = 1, None
        if ix.is_cuda:
            self.offsets.copy_(ix, non_blocking=True)      # stream-ordered device copy (range-checked where they were staged)
        else:
            self.check_offsets(ix)
            self.offsets.copy_(ix)                         # synchronous: stage blocks of offsets instead (stage_offsets)

    def stage_offsets(self, block: Tensor):
        """window offsets of the next block.shape[0] steps, [n, B] (host tensors are range-checked, device tensors are taken as
        checked): step k after this call gathers row k by itself -- the captured step reads its row from the device-side step
        counter, so a training loop is nothing but graph launches between two stagings"""
        if block.dim() != 2 or block.shape[1] != self.B or block.shape[0] < 1:
            raise ValueError(f"stage_offsets: need [n >= 1, B = {self.B}] offsets")
        if not block.is_cuda:
            self.check_offsets(block)
        n = block.shape[0]
        if n > self.off_block.shape[0]:
            # a larger block buffer is a new address: captured graphs go (and with them the view set_offsets writes)
            self.off_block = torch.zeros((n, self.B), dtype=torch.int64, device=self.dev)
            self.offsets = self.off_block[0]
            self._graphs = None
        self.off_block[:n].copy_(block, non_blocking=block.is_cuda)
        self.off_ctl[0:1].copy_(self.state[2:3])          # device-side: row = step word - step word now
        self.off_ctl[1:2].fill_(n)
        self._off_rows, self._off_left = n, n

    def check_offsets(self, ix: Tensor):
        """window offsets must leave room for T + 1 tokens (ref: randint(len(data) - context_length), src/preprocessing.py:43);
        host tensors are checked for free, device tensors cost one round trip: check a staged block once, not every step"""
        if self.corpus is not None and ix.numel():
            lo, hi = (int(v) for v in torch.aminmax(ix))
            if lo < 0 or hi + self.T + 1 > self.corpus.numel():
                raise IndexError(f"window offsets in [{lo}, {hi}] do not fit a corpus of {self.corpus.numel()} tokens at T = {self.T}")

    def set_batch(self, x: Tensor, y: Tensor):
        """token / target ids of this rank's rows, given directly (ids are validated here; the captured step never checks)"""
        ops.check_ids(x, self.V, "x")
        ops.check_ids(y, self.V, "y")
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def step(self) -> Tensor:
        """one training iteration on the current offsets / batch; returns the device loss scalar"""
        if self._off_left is not None:
            if self._off_left <= 0:
                raise RuntimeError(f"step(): the {self._off_rows} staged offset rows are used up; stage_offsets() or set_offsets() first")
            self._off_left -= 1
        if not self.use_graph:
            if self._dp() and self.dp_buckets > 1:
                segs, ranges = self._prog_segments()
                works = []
                for seg, r in zip(segs, ranges):
                    seg()
                    self._allreduce_ranges_async(r, works)
                for w in works:
                    w.wait()
            else:
                self._prog_fwd_bwd()
                self._allreduce()
            self._prog_update()
            return self.loss
        if self._graphs is None:
            self._capture()
        if not self._dp():
            self._graphs[0].replay()
        elif self.dp_buckets > 1:
            works = []
            ev = self._timing_events(3)
            for g, r in zip(self._graphs[:-1], self._seg_ranges):
                g.replay()
                self._allreduce_ranges_async(r, works)
            if ev: ev[1].record()
            for w in works:
                w.wait()                    # stream-level join with the RCCL stream (no host block with the nccl backend)
            if ev: ev[2].record()
            self._graphs[-1].replay()
            self._timing_done(ev, ("backward_graphs", "exposed_exchange", "optimizer_graph"))
        else:
            ev = self._timing_events(3)
            self._graphs[0].replay()
            if ev: ev[1].record()
            self._allreduce()
            if ev: ev[2].record()
            self._graphs[1].replay()
            self._timing_done(ev, ("backward_graph", "exchange", "optimizer_graph"))
        return self.loss

    def _timing_events(self, n: int):
        if not self.debug_timing:
            return None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        return ev

    def _timing_done(self, ev, names) -> None:
        """debug_timing: where a multi-rank step spends its time (one host sync; never on in a timed run)"""
        if ev:
            ev[-1].record()
            ev[-1].synchronize()
            self.last_timing = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}

    @torch.no_grad()
    def eval_loss(self, x: Tensor, y: Tensor) -> Tensor:
        """forward only, dropout off (ref: evaluate_loss, src/train.py:61-75)"""
        run = S.Run(act=self.act, rng=None, weights=self.weights, fp8=self.fp8)
        _, rows, _ = self._forward(run, x, y, False)
        return ops.reduce_sum(rows, 1.0 / rows.numel())

    @torch.no_grad()
    def eval_losses(self, data: Tensor, offsets: Tensor) -> Tensor:
        """losses of offsets.shape[0] forward-only batches (dropout off) drawn from the resident token stream `data` at the
        window offsets `offsets` [n, B] (int64, on the device): the inner loop of evaluate_loss (ref: src/train.py:66-72)
        as one gather launch + one captured forward graph per batch and no host synchronisation; returns [n] on the device.
        Eagerly it was ~55 ctypes launches and a `.item()` per batch -- more host time than the 0.85 ms the GPU needs."""
        n, B = offsets.shape
        if B != self.B:
            raise ValueError(f"eval_losses: batches of {B} rows, the engine was built for {self.B}")
        out = torch.empty(n, dtype=torch.float32, device=self.dev)
        if not self.use_graph:
            for i in range(n):
                x, y = ops.batch_gather(data, offsets[i], self.T)
                out[i:i + 1].copy_(self.eval_loss(x, y).view(1))
            return out
        if self._eval_graph is None:
            self.ev_x = torch.zeros((self.B, self.T), dtype=torch.int64, device=self.dev)
            self.ev_y = torch.zeros((self.B, self.T), dtype=torch.int64, device=self.dev)
            self.ev_loss = torch.zeros(1, dtype=torch.float32, device=self.dev)

            def prog():
                run = S.Run(act=self.act, rng=None, weights=self.weights, fp8=self.fp8)
                _, rows, _ = self._forward(run, self.ev_x, self.ev_y, False)
                ops.reduce_sum(rows, 1.0 / rows.numel(), out=self.ev_loss)
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # warm-up outside capture
                prog()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                prog()
            self._eval_graph = g
        for i in range(n):
            ops.batch_gather(data, offsets[i], self.T, self.ev_x, self.ev_y)
            self._eval_graph.replay()
            out[i:i + 1].copy_(self.ev_loss)
        return out

    def check_status(self) -> None:
        """raise if a bounded device-side wait of the grouped dW GEMM ever ran out (dg_gemm_tn_grouped: the sticky error word in
        the last 16 bytes of its workspace).  One device round trip: for tests / end-of-run checks, not for every step."""
        for ws in self.tn_workspaces.values():
            if int(ws[-16:].view(torch.int32)[0].item()) != 0:
                raise RuntimeError("dg_gemm_tn_grouped: a split-K hand-over timed out; weight gradients since then are invalid")

    def step_count(self) -> int:
        return int(self.state[2].item())
