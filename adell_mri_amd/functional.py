"""torch.autograd glue over the HIP kernels (``ops.py``).

Each Function is one fused site of the reference's hot path:

* ``conv3d``            Conv3d (+ virtual concat of two sources, + residual add,
                        + per-channel statistics for the norm that follows)
* ``conv_transpose3d_k2s2``  ConvTranspose3d(k=2, s=2)
* ``norm_drop_act``     Norm -> Dropout -> Activation of ActDropNorm ("NDA")

Statistics produced by the conv epilogue ride on the output tensor as
``y._adell_partials`` so that the following ``norm_drop_act`` does not re-read
``y``; they are dropped by any other consumer.
"""
import itertools
import os
import weakref

import numpy as np
import torch

from . import _lib, ops

_dropout_counter = itertools.count(1)

# How the MFMA convolutions multiply fp32 operands:
#   "f16x3": three f16 MFMAs on error-compensated hi/lo splits (fp32-class accuracy,
#            ~2^-22 per product; 5.3x the fp32 MFMA rate)            [default]
#   "fp32":  v_mfma_f32_32x32x2_f32, bit-exact k-ordered fp32 FMA chain
CONV_PRECISION = os.environ.get("ADELL_CONV_PRECISION", "f16x3")
# dispatch switches for A/B tests, read from the environment once at import (tests flip the
# dictionary entries): parity-class stride-2 backward-data at every size / never; no folding
# of the x taps of <= 4-channel inputs into the 16-channel MFMA chunk
FLAGS = {"s2class_always": bool(os.environ.get("ADELL_S2CLASS_ALWAYS")),
         "no_s2class": bool(os.environ.get("ADELL_NO_S2CLASS")),
         "no_fold": bool(os.environ.get("ADELL_NO_FOLD")),
         # norm / dropout / activation backward left to its own two passes (no fused epilogue)
         "no_adn_fuse": bool(os.environ.get("ADELL_NO_ADN_FUSE")),
         # the logits head's backward-data tensor is formed (no factored site backward)
         "no_lowrank": bool(os.environ.get("ADELL_NO_LOWRANK")),
         # weight gradients of the convolutions / linear layers on a second HIP stream (side_run);
         # ADELL_WGRAD_STREAM=0 keeps every launch on the caller's stream
         "wgrad_stream": os.environ.get("ADELL_WGRAD_STREAM", "1") != "0",
         # an ADN output whose only reader is a 3x3x3 stride-1 conv is written as SPLIT ROWS (the
         # conv kernels' LDS row image: ops.SplitRows) instead of fp32; ADELL_NO_ROWS=1: always fp32
         "no_rows": bool(os.environ.get("ADELL_NO_ROWS")),
         # 1x1x1 stride-1 convolutions with >= 64 channels on one side (>= 8 on the other) stay on the implicit-GEMM
         # conv kernels instead of the Linear-layer GEMMs (conv3d)
         "no_pointwise_gemm": bool(os.environ.get("ADELL_NO_POINTWISE_GEMM")),
         # full-sequence attention by slices + copies around the kernels (the pre-round-4 form)
         "no_seq_attention": bool(os.environ.get("ADELL_NO_SEQ_ATTENTION")),
         # Linear -> activation -> Linear as separate layers with an element-wise pass between them.
         # The DEFAULT: with the wide epilogue on every GEMM the fused form (functional.mlp,
         # ADELL_MLP_FUSE=1) measured 20.03 against 19.71 ms on the VICReg ConvNeXt step -- the erf of
         # 100 M hidden elements costs more inside a GEMM epilogue than in two passes that run at the
         # HBM rate -- and 21.6 against 21.3 ms on UNETR.
         "no_mlp_fuse": not bool(os.environ.get("ADELL_MLP_FUSE")),
         # (fused form only from this many hidden elements on)
         "mlp_min_elems": int(os.environ.get("ADELL_MLP_MIN_ELEMS", str(1 << 20))),
         # weight gradient of the narrow-input convs on the exact fp32-MFMA kernel (A/B)
         "no_cinfold_wgrad_f16": bool(os.environ.get("ADELL_NO_CINFOLD_WGRAD_F16"))}


def set_conv_precision(mode):
    global CONV_PRECISION
    if mode not in ("f16x3", "fp32"):
        raise ValueError("conv precision must be 'f16x3' or 'fp32'")
    CONV_PRECISION = mode
    _sync_dw_precision()


def _sync_dw_precision():
    """The depthwise 7^3 kernels choose their arithmetic inside the library (csrc/dw_mfma.hip: the
    f16x3 Toeplitz form on the MFMA): "fp32" keeps them on the exact fp32-FMA kernels too."""
    from . import _lib
    if os.path.exists(_lib.LIB_PATH) and not os.environ.get("ADELL_DW_NOMFMA"):
        _lib.lib().adell_set_tuning(b"dw_nomfma", 1 if CONV_PRECISION == "fp32" else 0)


if CONV_PRECISION == "fp32":
    _sync_dw_precision()


class _Ref:
    """Opaque holder so a Parameter object can ride through autograd.Function.apply."""

    __slots__ = ("obj",)

    def __init__(self, obj):
        self.obj = obj


# Conv weights packed for the f16x3 kernels (modes 0 / 1) are registered here so that, after an
# optimiser step has rewritten the parameters, ALL of them are repacked by one launch
# (adell_pack_weight_f16x3_multi) instead of one ~6 us launch per weight and mode.
_PACK_REG = {}      # (id(weight), mode) -> [weakref(weight), mode, SplitWeight, tag, fence]
_PACK_BATCH = {"epoch": -1, "sig": None, "table": None, "blocks": 0}


class _PackFence:
    """Where a pack was launched: a consumer on ANOTHER stream waits for the event first (the
    packed copies and the absmax arena used to assume that one stream drives every forward /
    backward-data launch of the process; two micro-batches on two streams read half-written
    packs -- round 3's diverged A/B run)."""

    __slots__ = ("stream", "event")

    def __init__(self):
        self.stream = ops._stream().value
        self.event = torch.cuda.Event()
        self.event.record()

    def order(self):
        if ops._stream().value != self.stream:
            torch.cuda.current_stream().wait_event(self.event)


def _pack_tag(w):
    return (w._version, w.data_ptr(), ops.WEIGHT_EPOCH)


def _repack_registered():
    """Repack every live registered weight in one launch; returns False when nothing is
    registered."""
    live = []
    for key, ent in list(_PACK_REG.items()):
        w = ent[0]()
        if w is None or not w.is_contiguous() or not w.is_cuda:
            del _PACK_REG[key]
            continue
        live.append((w, ent))
    if not live:
        return False
    rows, first = [], 0
    for w, ent in live:
        d0, d1 = w.shape[0], w.shape[1]
        taps = w.shape[2] * w.shape[3] * w.shape[4]
        rows.append((w.data_ptr(), ent[2].halfs.data_ptr(), ent[2].scale.data_ptr(), ent[1], d0,
                     d1, taps, first))
        first += d0 if ent[1] == 0 else d1
    sig = tuple(rows)
    if _PACK_BATCH["sig"] != sig and torch.cuda.is_current_stream_capturing():
        # the set of live weights changed inside a graph capture (another module died since the
        # warm-up step): the table upload is a pageable host copy, which a capture cannot hold --
        # the caller packs weight by weight (captured launches of the single-weight kernel)
        return False
    if _PACK_BATCH["sig"] != sig:
        _PACK_BATCH["table"] = torch.tensor(rows, dtype=torch.int64, device=live[0][0].device)
        _PACK_BATCH["sig"] = sig
        _PACK_BATCH["blocks"] = first
    ops.pack_weight_f16x3_multi(_PACK_BATCH["table"], len(rows), _PACK_BATCH["blocks"])
    fence = _PackFence()
    for w, ent in live:
        ent[3] = _pack_tag(w)
        ent[4] = fence
    _PACK_BATCH["epoch"] = ops.WEIGHT_EPOCH
    return True


def _packed(w, mode):
    """Repacked copy of a weight. The cache lives ON the tensor object (so it dies
    with it) and is valid only for the same storage address and version counter."""
    split = CONV_PRECISION == "f16x3" and mode in (0, 1, 2, 3)
    if split and mode in (0, 1) and w.dim() == 5 and w.is_contiguous():
        import weakref
        key = (id(w), mode)
        ent = _PACK_REG.get(key)
        if ent is not None and ent[0]() is not w:
            ent = None
        if ent is not None:
            if ent[3] == _pack_tag(w):
                ent[4].order()
                return ent[2]
            if _PACK_BATCH["epoch"] != ops.WEIGHT_EPOCH and _repack_registered() \
                    and ent[3] == _pack_tag(w):
                ent[4].order()
                return ent[2]
        p = ops.pack_weight_f16x3(w.detach(), mode)
        _PACK_REG[key] = [weakref.ref(w), mode, p, _pack_tag(w), _PackFence()]
        return p
    cache = getattr(w, "_adell_packs", None)
    if cache is None:
        cache = {}
        w._adell_packs = cache
    key = (mode, split)
    hit = cache.get(key)
    tag = (w._version, w.data_ptr(), ops.WEIGHT_EPOCH)
    if hit is not None and hit[0] == tag:
        if w.is_cuda:
            hit[2].order()
        return hit[1]
    wd = w.detach()
    if wd.dim() == 2:  # torch.nn.Linear weight == 1x1x1 convolution weight
        wd = wd.view(wd.shape[0], wd.shape[1], 1, 1, 1)
    if split and mode == 2:
        # transposed conv forward = 1x1x1 conv with F*Cout outputs: V[(f, co)][ci] = w[ci][co][f]
        v = wd.permute(2, 3, 4, 1, 0).reshape(-1, wd.shape[0], 1, 1, 1)
        p = ops.pack_weight_f16x3(v, 0)
    elif split and mode == 3:
        # its backward-data = kernel == stride conv with Cin outputs / Cout inputs: w as is
        p = ops.pack_weight_f16x3(wd, 0)
    else:
        p = ops.pack_weight_f16x3(wd, mode) if split else ops.pack_weight(wd, mode)
    cache[key] = (tag, p, _PackFence() if w.is_cuda else None)
    return p


def _packed_s2_classes(w, padding):
    """The 8 parity-class sub-kernels of a stride-2 k = 3 conv weight, packed for the
    backward-data kernel (mode 1); cached on the tensor."""
    cache = getattr(w, "_adell_packs", None)
    if cache is None:
        cache = {}
        w._adell_packs = cache
    tag = _pack_tag(w)
    key = ("s2", tuple(padding))
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        hit[2].order()
        return hit[1]
    wd = w.detach()
    packs = []
    for c in range(8):
        par = (c >> 2, (c >> 1) & 1, c & 1)
        t0 = [(par[a] + padding[a]) & 1 for a in range(3)]
        sub = wd[:, :, t0[0]::2, t0[1]::2, t0[2]::2].contiguous()
        packs.append(ops.pack_weight_f16x3(sub, 1))
    cache[key] = (tag, packs, _PackFence())
    return packs


def _packed_folded(w):
    """f16x3 pack of W'[co][kx * Cin + ci][kz][ky][0] = w[co][ci][kz][ky][kx], zero-padded to 16
    input channels (the weight of the folded conv, see ops.fold_x_taps); cached on the tensor."""
    cache = getattr(w, "_adell_packs", None)
    if cache is None:
        cache = {}
        w._adell_packs = cache
    tag = _pack_tag(w)
    hit = cache.get("fold")
    if hit is not None and hit[0] == tag:
        hit[2].order()
        return hit[1]
    wd = w.detach()
    co, ci, kd, kh, kw = wd.shape
    wf = wd.permute(0, 4, 1, 2, 3).reshape(co, kw * ci, kd, kh, 1)
    if kw * ci < 16:
        wf = torch.cat([wf, wf.new_zeros(co, 16 - kw * ci, kd, kh, 1)], 1)
    p = ops.pack_weight_f16x3(wf.contiguous(), 0)
    cache["fold"] = (tag, p, _PackFence())
    return p


# absmax by-product slots ([0]: input(s) of a conv, [1]: its dy), zeroed in bulk: every slot is
# handed out once and never reused, so a slot stays valid as long as the graph that holds it --
# one fill per 256 conv calls instead of one per call
# (one arena per launching stream: the bulk fill is ordered before the kernels that use its slots
# only on the stream it was issued on)
_AMAX_ARENAS = {}


def _amax_pair(device):
    key = (device, ops._stream().value)
    a = _AMAX_ARENAS.get(key)
    if a is None:
        a = _AMAX_ARENAS[key] = {"buf": None, "next": 0}
    if a["buf"] is None or a["next"] + 2 > a["buf"].numel():
        a["buf"] = torch.zeros(512, device=device, dtype=torch.int32)
        a["next"] = 0
    pair = a["buf"][a["next"]:a["next"] + 2]
    a["next"] += 2
    return pair


# ---- the weight-gradient branch on a second HIP stream --------------------------------------------
# dW of a conv depends on (x, dY) only and nothing in the backward pass waits for it: on its own
# stream the MFMA-bound weight-gradient kernels (and their small partial-slab folds) run beside the
# HBM-bound norm / dropout / activation passes and the launch tails of the backward-data chain.
# Ordering: the side stream waits for everything queued on the main stream so far (dY, the absmax
# slots); the main stream waits for the side stream when the backward pass ends (engine callback)
# and wherever a gradient is read before that (join_side_stream: GradSync buckets, the optimiser).
_SIDE = {"stream": None, "pending": False, "callback": False, "mains": [], "keep": [],
         "bytes": 0, "task": -1}
# the tensors the side stream reads are kept alive until the join: past this many bytes the main
# stream joins early (mid-backward) and lets them go -- bounds what the overlap adds to the
# step's peak memory (default: 1/8 of the device; ADELL_SIDE_KEEP_GB overrides)
_SIDE_KEEP_LIMIT = {"bytes": None}


def _side_keep_limit(device):
    if _SIDE_KEEP_LIMIT["bytes"] is None:
        env = os.environ.get("ADELL_SIDE_KEEP_GB")
        if env is not None:
            _SIDE_KEEP_LIMIT["bytes"] = int(float(env) * (1 << 30))
        else:
            _SIDE_KEEP_LIMIT["bytes"] = torch.cuda.get_device_properties(device).total_memory // 8
    return _SIDE_KEEP_LIMIT["bytes"]


def _side_join_callback():
    _SIDE["callback"] = False
    join_side_stream()


def join_side_stream():
    """Make the current stream -- and every stream a backward node handed work over from (the
    engine's final callbacks need not run under the stream of the forward pass) -- wait for the
    weight gradients still running on the side stream."""
    # (also re-arms the end-of-backward callback: a backward pass that raised never ran it)
    _SIDE["callback"] = False
    if _SIDE["pending"]:
        side = _SIDE["stream"]
        cur = torch.cuda.current_stream(side.device)
        cur.wait_stream(side)
        for m in _SIDE["mains"]:
            if m != cur:
                m.wait_stream(side)
        _SIDE["mains"] = []
        _SIDE["pending"] = False
        # the tensors the side stream read may go back to the allocator now: whatever stream
        # reuses their memory does so behind the wait just queued
        _SIDE["keep"].clear()
        _SIDE["bytes"] = 0


def side_stream_behind(main):
    """The weight-gradient stream, made to wait for ``main``, when it holds work of the running
    backward pass -- for consumers that would rather queue behind the gradients than stall the
    main stream for them (GradSync's bucket gathers); else None."""
    if not _SIDE["pending"] or _SIDE["stream"] is None or _SIDE["stream"].device != main.device:
        return None
    _SIDE["stream"].wait_stream(main)
    return _SIDE["stream"]


def side_keep(t):
    """Keep ``t`` alive until the side stream has been joined (for work a caller queued on it)."""
    _SIDE["keep"].append(t)


def side_run(fn, reads):
    """``fn()`` with its launches on the side stream; ``reads``: the tensors it reads. They are
    kept alive until the join (not ``record_stream``-ed: that defers the reuse of every such block
    behind an allocator event, and with the host a step ahead of the GPU the pools kept growing
    through fresh device allocations -- sporadic 0.3 s stalls in the first steps of a process)."""
    main = torch.cuda.current_stream()
    side = _SIDE["stream"]
    if side is None or side.device != main.device:
        # (a LOW-priority stream -- hipStreamCreateWithPriority, torch only hands out normal / high --
        # was measured: the main stream ends 0.3 ms earlier and waits that much longer at the join)
        side = _SIDE["stream"] = torch.cuda.Stream(device=main.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    for t in reads:
        if t is not None:
            _SIDE["keep"].append(t)
            _SIDE["bytes"] += t.numel() * t.element_size()
    _SIDE["pending"] = True
    if main not in _SIDE["mains"]:
        _SIDE["mains"].append(main)
    # one end-of-backward join per GraphTask: a backward pass that raised after a side_run never
    # ran its callback, and the next pass must queue its own (the flag alone would suppress it)
    task = torch._C._current_graph_task_id()
    if not _SIDE["callback"] or _SIDE["task"] != task:
        _SIDE["callback"], _SIDE["task"] = True, task
        torch.autograd.Variable._execution_engine.queue_callback(_side_join_callback)
    if _SIDE["bytes"] > _side_keep_limit(main.device):
        join_side_stream()      # early join: the kept tensors go back to the allocator
    return out


def _note_use(weight):
    """Forward side of _side_ok: counts the graph nodes that hold this weight. Two nodes in one
    graph (the semi-supervised step runs the network twice) have their gradients added by the
    autograd engine on the main stream, which knows nothing of the side stream."""
    if torch.is_grad_enabled() and weight.requires_grad:
        n = getattr(weight, "_adell_uses", 0) + 1
        weight._adell_uses = n
        if n > 1 or _inside_torch_ddp():
            weight._adell_multi = True


def _inside_torch_ddp():
    """True while a torch DistributedDataParallel wrapper runs the forward pass: its Reducer reads
    every gradient from a C++ hook on the AccumulateGrad node, on the main stream, as soon as the
    node returns -- invisible to _side_ok's hook checks, so such weights stay on the main stream."""
    try:
        from torch.nn.parallel import DistributedDataParallel as _DDP
        return getattr(_DDP, "_active_ddp_module", None) is not None
    except Exception:       # torch built without distributed
        return False


def reset_uses(params):
    """A new step: forget graph nodes that never ran their backward (FlatParameters.zero_grad)."""
    if torch.cuda.is_available() and _SIDE["pending"]:
        join_side_stream()
    for p in params:
        if getattr(p, "_adell_uses", 0):
            p._adell_uses = 0
            p._adell_multi = False


def _side_ok(weight, *params):
    """Whether the gradients of these leaf parameters may be produced off the main stream: nothing
    on the main stream reads them before the join (one graph node per weight, no accumulation into
    an existing .grad, no tensor hooks, no further autograd nodes). Called once per backward node."""
    uses = getattr(weight, "_adell_uses", 0)
    multi = getattr(weight, "_adell_multi", False)
    if uses > 0:
        weight._adell_uses = uses - 1
        if uses == 1:
            weight._adell_multi = False
    if uses != 1 or multi:
        return False
    if not FLAGS["wgrad_stream"] or torch.is_grad_enabled():     # (create_graph: dW feeds a graph)
        return False
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    for p in (weight,) + params:
        if p is None:
            continue
        if not p.is_leaf or p.grad is not None or p._backward_hooks:
            return False
        # OPT-IN: the gradient's consumer must be known to join the side stream before it reads.
        # That is optim.FlatParameters (collect / zero_grad join) on one process, and with a
        # process group up additionally parallel.GradSync (its hooks and its all_reduce() go
        # through collect). Anything else -- a torch optimiser, torch DDP's Reducer (a C++ hook on
        # the AccumulateGrad node that copies dW into its bucket on the main stream the moment the
        # node returns) -- gets its gradients on the main stream.
        if not getattr(p, "_adell_flat", False):
            return False
        if dist_on and not getattr(p, "_adell_gradsync", False):
            return False
        # post-accumulate hooks read the gradient on the main stream: only GradSync's own (which
        # join first, FlatParameters.collect) are known to be safe
        if getattr(p, "_post_accumulate_grad_hooks", None) and not getattr(p, "_adell_gradsync", False):
            return False
    return True


# ---- split rows between an ADN site and the conv that is its only reader -----------------------------
# Module code that knows "the output of THIS ActDropNorm is read by THAT Conv3d and by nothing else"
# says so before it runs the ADN (expect_rows); the ADN's forward takes the note (take_rows_reader)
# and, when the conv's kernels stage split rows for this shape, writes rows instead of fp32 values
# (same bytes). The note is keyed by the ADN module: no other norm_drop_act call can pick it up.
_ROWS_EXPECT = weakref.WeakKeyDictionary()   # keyed by the ADN module itself (an id() can be reused)
_ROWS_PLAN = {}


def _hooked(producers, readers):
    """Could anything but the announced conv see the tensor? Forward hooks on a module whose output
    it is (the ADN, the link / decoder block around it, anything inside), forward pre-hooks on a
    module whose input it is, or global module hooks (feature extraction, Grad-CAM, profilers):
    they would be handed fp32-typed memory holding fp16 row pairs."""
    gm = torch.nn.modules.module
    if gm._global_forward_hooks or gm._global_forward_pre_hooks:
        return True
    for top in producers:
        for m in top.modules():
            if m._forward_hooks:
                return True
    for top in readers:
        for m in top.modules():
            if m._forward_pre_hooks:
                return True
    return False


def expect_rows(adn_module, conv, as_x1=False, producers=(), readers=()):
    """``conv``: a modules.layers.conv.Conv3d; ``as_x1``: the tensor will be the conv's X_cat.
    ``producers`` / ``readers``: the enclosing modules that return / receive the same tensor (the
    link block and the decoder block of a skip hand-over); with the ADN and the conv themselves
    they are searched for hooks that would observe the tensor (``_hooked``): then it stays fp32."""
    spec = conv.rows_spec() if (not FLAGS["no_rows"] and hasattr(conv, "rows_spec")) else None
    if spec is not None and _hooked((adn_module,) + tuple(producers), (conv,) + tuple(readers)):
        spec = None
    if spec is None:
        _ROWS_EXPECT.pop(adn_module, None)
    else:
        _ROWS_EXPECT[adn_module] = spec + (bool(as_x1),)


def clear_row_expectations():
    """A new top-level forward: notes left by a pass that never reached its ADN are void."""
    if _ROWS_EXPECT:
        _ROWS_EXPECT.clear()


def take_rows_reader(adn_module):
    return _ROWS_EXPECT.pop(adn_module, None) if _ROWS_EXPECT else None


def _rows_exponent(x, reader, norm, gamma, beta, act, act_p, act_w, p):
    """Exponent for a split-row output of this site, or None when it must stay fp32: instance
    statistics without affine parameters bound the normalised value by sqrt(V), dropout scales by
    1 / (1 - p), and |act(u)| <= |u| for the activations listed -- so 2^exp is chosen on the host,
    without a pass over the data, with the largest possible value at 2^14 (fp16 tops out at 2^16;
    typical values sit ~2^10 below the bound, where hi + lo still carry 22 bits and the absolute
    error floor is 2^-25 of the scaled unit)."""
    if (reader is None or FLAGS["no_rows"] or CONV_PRECISION != "f16x3" or norm != "instance"
            or p >= 1.0
            or gamma is not None or beta is not None or act_w is not None or x.dim() != 5
            or act not in ("identity", "swish", "silu", "relu", "leaky_relu", "gelu")
            or (act == "leaky_relu" and abs(act_p) > 1.0)):
        return None
    weight, stride, padding, as_x1 = reader
    N, C = x.shape[:2]
    spatial = tuple(x.shape[2:])
    other = weight.shape[1] - C
    if weight.dim() != 5 or other < 0 or C < 16 or (C & (C - 1)) or C > 1024:
        return None
    C0, C1 = (other, C) if as_x1 else (C, other)
    if C0 == 0:
        C0, C1 = C1, 0
    grad = torch.is_grad_enabled() and weight.requires_grad
    key = (N, spatial, C0, C1, tuple(weight.shape), stride, padding, grad, ops.plan_epoch())
    ok = _ROWS_PLAN.get(key)
    if ok is None:
        k = tuple(weight.shape[2:])
        ok = ops.conv3d_rows_ok(N, spatial, C0, C1, weight.shape[0], k, stride, padding)
        if ok and grad:
            ok = ops.conv3d_bwd_weight_rows_ok(N, spatial, C0, C1, weight.shape[0], k, stride,
                                               padding)
        _ROWS_PLAN[key] = ok
    if not ok:
        return None
    V = spatial[0] * spatial[1] * spatial[2]
    bound = float(np.sqrt(V)) / (1.0 - p)
    return 14 - int(np.ceil(np.log2(max(bound, 1.0))))


def materialize(t):
    """``t`` as plain fp32 values (a split-row tensor converted back; anything else unchanged). For
    the rare reader that is not the conv the rows were written for. Not differentiable: use on
    tensors whose gradient does not matter or inside an autograd.Function."""
    rows = getattr(t, "_adell_rows", None)
    return t if rows is None else ops.rows_to_f32(t.detach(), rows)


class _Conv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, residual, wp, conf):
        (stride, padding, want_stats, wref, ctx.carry_in, ctx.carry_out, ctx.carry_x0,
         ctx.carry_cat, ctx.adn, rows0, rows1) = conf
        if (rows0 is not None or rows1 is not None) and not isinstance(wp, ops.SplitWeight):
            # a reader other than the f16x3 implicit-GEMM path: it gets fp32 values (and so does
            # its backward: the converted tensors are the ones saved)
            ops.ROWS_FALLBACKS[0] += 1
            if rows0 is not None:
                x0, rows0 = ops.rows_to_f32(x0, rows0), None
            if rows1 is not None:
                x1, rows1 = ops.rows_to_f32(x1, rows1), None
        ctx.rows = (rows0, rows1)
        k = tuple(weight.shape[2:]) if weight.dim() == 5 else (1, 1, 1)
        # the statistics partials are a non-differentiable by-product: without this autograd
        # materialises a zero gradient for them in every backward (a 2 MB fill per conv site)
        ctx.set_materialize_grads(False)
        ctx.bias_ref = _Ref(bias)
        ctx.cin_small = wp == "cin_small"
        ctx.cinfold = wp == "cinfold"
        if ctx.cinfold:
            y, part = ops.conv_cinfold_fwd(x0, weight, bias, padding, want_stats,
                                           f16x3=CONV_PRECISION == "f16x3")
            ctx.small1 = False
            ctx.amax = None
            ctx.save_for_backward(x0, x1, weight)
            ctx.conf = (k, stride, padding, bias is not None, False, wref)
            if part is None:
                part = y.new_empty(0)
            ctx.mark_non_differentiable(part)
            return y, part
        if ctx.cin_small:  # 2-channel input block: exact fp32 on the vector ALU, canonical weights
            y, part = ops.conv_cin_small_fwd(x0, weight, bias, padding, want_stats)
            ctx.small1 = False
            ctx.amax = None
            ctx.save_for_backward(x0, x1, weight)
            ctx.conf = (k, stride, padding, bias is not None, False, wref)
            if part is None:
                part = y.new_empty(0)
            ctx.mark_non_differentiable(part)
            return y, part
        if isinstance(wp, tuple) and wp[0] == "fold":
            # Cin <= 4, Kw * Cin <= 16: the x taps ride in the 16-channel MFMA chunk (forward
            # only: the gradients below see the original tensors and weights)
            xf = ops.fold_x_taps(x0, k[2], padding[2])
            y, part = ops.conv3d_fwd(xf, wp[1], bias, weight.shape[0], (k[0], k[1], 1), stride,
                                     (padding[0], padding[1], 0), residual=residual,
                                     want_stats=want_stats)
            ctx.small1 = False
            ctx.amax = None
            ctx.save_for_backward(x0, x1, weight)
            ctx.conf = (k, stride, padding, bias is not None, residual is not None, wref)
            if part is None:
                part = y.new_empty(0)
            ctx.mark_non_differentiable(part)
            return y, part
        ctx.small1 = wp is None
        if ctx.small1:  # logits head: 1x1x1, Cout <= 4 -- one HBM-bound pass on canonical weights
            y = ops.conv1_small_fwd(x0, x1, weight, bias)
            ctx.save_for_backward(x0, x1, weight)
            ctx.conf = (k, stride, padding, bias is not None, False, wref)
            part = y.new_empty(0)
            ctx.mark_non_differentiable(part)
            return y, part
        # [0]: absmax bits of the input(s), [1]: of dy -- by-products of the f16x3 forward /
        # backward-data kernels that the backward-weight kernel uses as operand scales
        amax = None
        if isinstance(wp, ops.SplitWeight) and ctx.needs_input_grad[2]:
            amax = _amax_pair(x0.device)
        y, part = ops.conv3d_fwd(x0, wp, bias, weight.shape[0], k, stride, padding, x1=x1,
                                 residual=residual, want_stats=want_stats,
                                 amax=None if amax is None else amax[0:1], rows0=rows0, rows1=rows1)
        ctx.amax = amax
        ctx.save_for_backward(x0, x1, weight)
        ctx.conf = (k, stride, padding, bias is not None, residual is not None, wref)
        if part is None:
            part = y.new_empty(0)
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    def backward(ctx, dy, _dpart):
        x0, x1, weight = ctx.saved_tensors
        k, stride, padding, has_bias, has_res, wref = ctx.conf
        need = ctx.needs_input_grad
        if dy is None:   # the output took no part in the loss: only a parked gradient passes through
            _side_ok(wref.obj)      # (bookkeeping: this node is done)
            add0 = ctx.carry_in.take() if ctx.carry_in is not None else None
            return (add0 if need[0] else None), None, None, None, None, None, None
        dy = ops.ndhwc(dy)
        dx0 = dx1 = dw = db = dres = None
        add0 = ctx.carry_in.take() if ctx.carry_in is not None else None
        C0 = x0.shape[1]
        C1 = 0 if x1 is None else x1.shape[1]
        if ctx.small1:
            if ctx.adn is not None and need[0]:
                # dX = dy (x) weight stays factored: a stride-0 placeholder goes down the graph
                ctx.adn[1].lowrank = (dy, weight.detach())
                dx0 = dy.new_zeros(()).expand(x0.shape)
            elif need[0] or (x1 is not None and need[1]):
                dx0, dx1 = ops.conv1_small_bwd_data(dy, weight, tuple(x0.shape[2:]), C0, C1)
                dx0 = dx0 if need[0] else None
                dx1 = dx1 if (x1 is not None and need[1]) else None
            if need[2] or (has_bias and need[3]):
                if _side_ok(wref.obj, ctx.bias_ref.obj):
                    dw, db = side_run(lambda: ops.conv1_small_bwd_weight(
                        x0, x1, dy, has_bias and need[3]), (x0, x1, dy))
                else:
                    dw, db = ops.conv1_small_bwd_weight(x0, x1, dy, has_bias and need[3])
                dw = dw.view(weight.shape) if need[2] else None
            if add0 is not None and dx0 is not None:
                dx0 = dx0 + add0
            dx0, dx1 = _park_input_grads(ctx, dx0, dx1)
            return dx0, dx1, dw, db, None, None, None
        amax = ctx.amax
        dy_amax = None
        if getattr(ctx, "cinfold", False) and need[0]:
            dx0 = ops.conv_cinfold_bwd_data(dy, weight, tuple(x0.shape[2:]), padding,
                                            f16x3=CONV_PRECISION == "f16x3")
        if dx0 is not None:
            pass
        elif ctx.cin_small and need[0] and dy.shape[1] % 4 == 0:
            dx0 = ops.conv_cin_small_bwd_data(dy, weight, tuple(x0.shape[2:]), padding)
        elif (ctx.cin_small and need[0] and dy.shape[1] <= 4 and stride == (1, 1, 1)
              and tuple(dy.shape[2:]) == tuple(x0.shape[2:])):
            # Cout <= 4 too (the 2 -> 2 conv of an input block whose input carries a gradient:
            # SWIN-UNet): dX is the same small conv of dY with the taps flipped and the channel
            # axes swapped (padding k - 1 - p) -- on the vector-ALU forward kernel instead of a
            # 32-column MFMA tile that is 94 % empty (1.37 ms -> 0.1 ms at 256 x 256 x 128)
            wt = weight.detach().transpose(0, 1).flip(2, 3, 4).contiguous()
            pt = tuple(kk - 1 - pp for kk, pp in zip(k, padding))
            dx0, _ = ops.conv_cin_small_fwd(dy, wt, None, pt, False)
        elif (need[0] and x1 is None and CONV_PRECISION == "f16x3"
              and ops.conv3d_bwd_data_s2_fused_ok(x0.shape[2:], C0, C1, dy.shape[1], k, stride,
                                                  padding)
              and isinstance(_packed(wref.obj, 1), ops.SplitWeight)):
            # the U-Net downsampling layer at 32 channels: one persistent launch
            if amax is not None:
                dy_amax = amax[1:2]
            dx0 = ops.conv3d_bwd_data_s2_fused(dy, _packed(wref.obj, 1), tuple(x0.shape[2:]),
                                               amax=dy_amax, add0=add0)
            add0 = None
        elif (need[0] and x1 is None and CONV_PRECISION == "f16x3" and stride == (2, 2, 2)
              and k == (3, 3, 3) and all(p <= 1 for p in padding)
              and all(s % 2 == 0 for s in x0.shape[2:])
              and (x0.numel() // C0 >= (1 << 20) or FLAGS["s2class_always"])
              and not FLAGS["no_s2class"]):
            # stride-2 backward-data by parity classes (no zero-inserted MFMA work). Eight
            # launches: pays from ~1 M voxels (measured: 128^3 0.52 -> 0.33 ms, 32^3 0.06 -> 0.16)
            if amax is not None:
                dy_amax = amax[1:2]
            dx0 = ops.conv3d_bwd_data_s2(dy, _packed_s2_classes(wref.obj, padding),
                                         tuple(x0.shape[2:]), C0, padding, amax=dy_amax,
                                         add0=add0)
            add0 = None
        elif (ctx.adn is not None and need[0] and (x1 is None or need[1])
              and (add0 is None or C1 == 0) and isinstance(_packed(wref.obj, 1), ops.SplitWeight)):
            # the input(s) are outputs of norm -> dropout -> activation sites read by this conv
            # only: their activation / dropout derivative and the two sums of the norm's backward
            # come out of this kernel's epilogue (AdnSite)
            site0, site1, ntiles, epoch = ctx.adn
            wpb = _packed(wref.obj, 1)
            if amax is not None:
                dy_amax = amax[1:2]
            if epoch != ops.plan_epoch():
                # the launch plan changed between forward and backward (adell_set_tuning): the row
                # count of the partial-sum buffer is the CURRENT plan's, or the plain path if that
                # plan has no fused epilogue (the sites then run their own two passes)
                ntiles = ops.conv3d_bwd_data_adn_ntiles(tuple(x0.shape[2:]), x0.shape[0], C0, C1,
                                                        dy.shape[1], k, stride, padding)
            if ntiles > 0:
                dx0, dx1, part = ops.conv3d_bwd_data_adn(dy, wpb, tuple(x0.shape[2:]), C0, C1, k,
                                                         stride, padding, ntiles, site0=site0,
                                                         site1=site1, amax=dy_amax, add0=add0)
                add0 = None
                if site0 is not None:
                    site0.fused, site0.part, site0.poff = True, part, 0
                if site1 is not None:
                    site1.fused, site1.part, site1.poff = True, part, C0
            else:
                fused = add0 is not None and C1 == 0
                dx0, dx1 = ops.conv3d_bwd_data(dy, wpb, tuple(x0.shape[2:]), C0, C1, k, stride,
                                               padding, amax=dy_amax, add0=add0 if fused else None)
                if fused:
                    add0 = None
                if x1 is None or not need[1]:
                    dx1 = None
        elif need[0] or (x1 is not None and need[1]):
            wpb = _packed(wref.obj, 1)
            if amax is not None and isinstance(wpb, ops.SplitWeight):
                dy_amax = amax[1:2]
            fused = add0 is not None and C1 == 0 and isinstance(wpb, ops.SplitWeight)
            dx0, dx1 = ops.conv3d_bwd_data(dy, wpb, tuple(x0.shape[2:]), C0, C1, k, stride,
                                           padding, amax=dy_amax, add0=add0 if fused else None)
            if fused:
                add0 = None
            if not need[0]:
                dx0 = None
            if x1 is None or not need[1]:
                dx1 = None
        want_db = has_bias and need[3]

        def weight_grads():
            dw = db = None
            if need[2] and getattr(ctx, "cinfold", False):
                dw, db = ops.conv_cinfold_bwd_weight(
                    x0, dy, padding, want_db,
                    # (0.297 vs 0.339 ms for 2 -> 32 at 2 x 128^3, 0.167 vs 0.207 ms for 1 -> 16 at
                    # 4 x 96^3: tools/cinfold_wgrad_time.py)
                    f16x3=(CONV_PRECISION == "f16x3" and not FLAGS["no_cinfold_wgrad_f16"]))
                dw = dw.view(weight.shape)
            elif need[2]:
                rows0, rows1 = getattr(ctx, "rows", (None, None))
                dw = ops.conv3d_bwd_weight(x0, dy, k, stride, padding, x1=x1, want_db=want_db,
                                           f16x3=(CONV_PRECISION == "f16x3"),
                                           x_amax=None if amax is None else amax[0:1],
                                           dy_amax=dy_amax, rows0=rows0, rows1=rows1)
                if want_db:
                    dw, db = dw
                dw = dw.view(weight.shape)
            elif want_db:
                db = ops.bias_grad(dy)
            return dw, db

        if (need[2] or want_db) and _side_ok(wref.obj, ctx.bias_ref.obj):
            dw, db = side_run(weight_grads, (x0, x1, dy))
        else:
            dw, db = weight_grads()
        if add0 is not None and dx0 is not None:   # a path without the fused add
            dx0 = dx0 + add0
        dx0, dx1 = _park_input_grads(ctx, dx0, dx1)
        if has_res and need[4]:
            if ctx.carry_out is not None:
                ctx.carry_out.grad = dy      # the head conv of the block adds it to its dX
            else:
                dres = dy
        return dx0, dx1, dw, db, dres, None, None


def _park_input_grads(ctx, dx0, dx1):
    """Skip fork: leave dX of x0 / x1 with the consumer of the same tensor that runs later."""
    if dx0 is not None and ctx.carry_x0 is not None:
        ctx.carry_x0.grad, dx0 = dx0, None
    if dx1 is not None and ctx.carry_cat is not None:
        ctx.carry_cat.grad, dx1 = dx1, None
    return dx0, dx1


def grad_observed(t):
    """True when someone watches the gradient of ``t`` (tensor hooks, ``retain_grad()``): a
    GradCarry routes part of that gradient around autograd, so the watcher would see a partial
    value -- callers then leave the fork to autograd's own accumulation. ``ADELL_NO_GRAD_CARRY=1``
    switches the carries off altogether (gradient-inspection tools, partial
    ``torch.autograd.grad(inputs=...)`` calls that run the parking conv but not the taking one)."""
    return bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "retains_grad", False))


class GradCarry:
    """Hands a gradient from one consumer of a tensor to another consumer of the SAME tensor whose
    backward runs later and whose backward-data kernel adds it in its epilogue -- one full-size
    add pass (autograd's accumulation) less per fork.
    * residual link (``op(X) + X``, res_blocks.py:192): the conv that adds the link parks its dy
      (``carry_out``), the conv at the head of the block takes it (``carry_in``);
    * U-Net skip fork (unet.py:768-822): the first conv of the link op parks its dX
      (``carry_x0``; with identity links it is the decoder conv that reads the level output as
      the second half of its channel concat, ``carry_cat``), the strided conv that downsamples the
      same level output takes it (``carry_in``). The downsampling conv's backward depends on the
      decoder's (through the bottleneck), so it always runs later."""

    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None

    def take(self):
        g, self.grad = self.grad, None
        return g


class AdnSite:
    """A norm -> dropout -> activation site as the backward-data kernel of its consumer needs it
    (ops.conv3d_bwd_data_adn): the site's input ``x``, its instance statistics, the keep bits its
    forward stored and the activation. The consumer's backward sets ``fused`` / ``part`` / ``poff``
    when its epilogue produced dt and the partial sums; the site's own backward then runs ONE
    elementwise pass (ops.norm_act_bwd_from_dt) instead of two."""

    __slots__ = ("x", "mean", "rstd", "mask", "drop_p", "act", "act_p", "fused", "part", "poff",
                 "lowrank")

    def __init__(self, x, mean, rstd, mask, drop_p, act, act_p):
        self.x, self.mean, self.rstd, self.mask = x, mean, rstd, mask
        self.drop_p, self.act, self.act_p = drop_p, act, act_p
        self.fused, self.part, self.poff = False, None, 0
        # (dy, weight) of a 1x1x1 consumer with <= 4 output channels: its backward-data result is
        # never formed, the site's backward reads the two factors (ops.norm_act_bwd_lowrank)
        self.lowrank = None


def single_use(t):
    """Module code declares: the next ``conv3d`` that reads ``t`` (as x0 or x1) is the ONLY
    consumer of ``t``. Only then may that conv's backward-data kernel hand the site behind ``t``
    a pre-multiplied gradient (AdnSite): a second consumer would add a plain gradient to it."""
    if t is not None and getattr(t, "_adell_site", None) is not None:
        t._adell_single = True
    return t


_ADN_PLAN = {}


def _adn_sites_of(x0, x1, weight, stride, padding):
    """(site0, site1, ntiles) when x0 / x1 carry single-use ADN sites and the backward-data
    launch of this conv takes the fused epilogue, else None."""
    # (someone watching the gradient of the site's output -- a tensor hook, retain_grad() -- must see
    # the true gradient, not dt: such a site keeps its own two-pass backward)
    s0 = (getattr(x0, "_adell_site", None)
          if getattr(x0, "_adell_single", False) and not grad_observed(x0) else None)
    s1 = (getattr(x1, "_adell_site", None)
          if x1 is not None and getattr(x1, "_adell_single", False) and not grad_observed(x1)
          else None)
    if (s0 is None and s1 is None) or CONV_PRECISION != "f16x3" or FLAGS["no_adn_fuse"] \
            or weight.dim() != 5 or not torch.is_grad_enabled():
        return None
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    # (the row count is a property of the library's launch plan, which process-wide switches --
    # adell_set_tuning -- change: one answer per plan epoch)
    key = (tuple(x0.shape), C1, tuple(weight.shape), stride, padding, ops.plan_epoch())
    nt = _ADN_PLAN.get(key)
    if nt is None:
        nt = ops.conv3d_bwd_data_adn_ntiles(tuple(x0.shape[2:]), x0.shape[0], C0, C1,
                                            weight.shape[0], tuple(weight.shape[2:]), stride,
                                            padding)
        _ADN_PLAN[key] = nt
    if nt <= 0:
        return None
    return (s0, s1, nt, key[-1])


def conv3d(x0, weight, bias=None, stride=1, padding=0, x1=None, residual=None, want_stats=True,
           carry_in=None, carry_out=None, carry_x0=None, carry_cat=None):
    """Conv3d over the virtual concatenation [x0, x1] (+ bias + residual). ``carry_in`` /
    ``carry_out`` / ``carry_x0`` / ``carry_cat``: see GradCarry."""
    stride, padding = ops._triple(stride), ops._triple(padding)
    # A 1x1x1 stride-1 conv over channels-last memory IS a Linear layer over the voxels: the GEMM
    # kernels take it (fused bias). On the implicit-GEMM conv kernels a brick's voxels x one 16-channel
    # chunk per step is the wrong tiling for it: ConvNeXt's stage transitions (384 -> 768 at 64 x 2^3
    # voxels) ran 0.79 ms forward and 0.78 ms backward-data at 0.4 TF, UNETR's 512 -> 512 at 12^3
    # 0.19 / 0.21 / 0.32 ms. (No statistics partials then: a following instance norm computes its own.)
    if (weight.dim() == 5 and tuple(weight.shape[2:]) == (1, 1, 1) and stride == (1, 1, 1)
            and padding == (0, 0, 0) and x1 is None and residual is None and carry_in is None
            and carry_out is None and carry_x0 is None and carry_cat is None and x0.dim() == 5
            and getattr(x0, "_adell_rows", None) is None and max(weight.shape[:2]) >= 64
            and min(weight.shape[:2]) >= 8 and not FLAGS["no_pointwise_gemm"]):
        xr = ops.ndhwc(x0)
        N, C, D, H, W = xr.shape
        y2 = linear(xr.permute(0, 2, 3, 4, 1).reshape(-1, C), weight.view(weight.shape[0], C), bias)
        return y2.view(N, D, H, W, -1).permute(0, 4, 1, 2, 3)
    _note_use(weight)
    adn = _adn_sites_of(x0, x1, weight, stride, padding)
    rows0 = getattr(x0, "_adell_rows", None)
    rows1 = getattr(x1, "_adell_rows", None) if x1 is not None else None
    conf = (stride, padding, want_stats, _Ref(weight), carry_in, carry_out, carry_x0, carry_cat,
            adn, rows0, rows1)
    Cin = x0.shape[1] + (0 if x1 is None else x1.shape[1])
    small1 = ops.conv1_small_ok(weight, Cin, stride, padding, residual)
    if small1:
        wp = None
        # the logits head behind a single-use site: the site's backward takes (dy, weight)
        site = getattr(x0, "_adell_site", None) if getattr(x0, "_adell_single", False) else None
        if (site is not None and x1 is None and carry_in is None and carry_x0 is None
                and not grad_observed(x0)
                and not FLAGS["no_adn_fuse"] and not FLAGS["no_lowrank"]
                and torch.is_grad_enabled() and ops.norm_act_lowrank_ok(x0, weight.shape[0])):
            conf = conf[:8] + (("lowrank", site),) + conf[9:]
    elif ops.conv_cin_small_ok(weight, x0, x1, stride, padding, residual):
        wp = "cin_small"
    elif ops.conv_cinfold_ok(weight, x0, x1, stride, padding, residual):
        wp = "cinfold"   # K = 27 Cin im2col GEMM on the fp32 MFMA (forward, dW); dX: f16x3 igemm
    elif (CONV_PRECISION == "f16x3" and x1 is None and weight.dim() == 5 and x0.shape[1] <= 4
          and 3 <= weight.shape[4] and weight.shape[4] * x0.shape[1] <= 16
          and stride == (1, 1, 1) and not FLAGS["no_fold"]):
        wp = ("fold", _packed_folded(weight))
    else:
        wp = _packed(weight, 0)
    y, part = _Conv3dFn.apply(x0, x1, weight, bias, residual, wp, conf)
    if want_stats and part.numel() > 0:
        y._adell_partials = part
    return y


class _ConvT3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, wp, wref):
        factors = tuple(weight.shape[2:])
        ctx.k2 = wp is None    # streaming factor-2 kernels on the canonical weight
        if ctx.k2:
            y = ops.convt_k2_fwd(x, weight, bias)
        else:
            y = ops.convtranspose3d_fwd(x, wp, bias, weight.shape[1], factors)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_ref = _Ref(bias)
        ctx.wref = wref
        ctx.factors = factors
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need = ctx.needs_input_grad
        dy = ops.ndhwc(dy)
        dx = dw = db = None
        if need[0]:
            dx = (ops.convt_k2_bwd_data(dy, weight) if ctx.k2 else
                  ops.convtranspose3d_bwd_data(dy, _packed(ctx.wref.obj, 3), weight.shape[0],
                                               ctx.factors))
        want_db = ctx.has_bias and need[2]

        def weight_grads():
            dw = db = None
            if need[1] and ctx.k2 and want_db:
                dw, db = ops.convt_k2_bwd_weight(x, dy, want_db=True, factors=ctx.factors)   # db from the same pass over dy
            elif need[1]:
                dw = (ops.convt_k2_bwd_weight(x, dy, factors=ctx.factors) if ctx.k2 else
                      ops.convtranspose3d_bwd_weight(x, dy, ctx.factors))
            if want_db and db is None:
                db = ops.bias_grad(dy)
            return dw, db

        if (need[1] or want_db) and _side_ok(ctx.wref.obj, ctx.bias_ref.obj):
            dw, db = side_run(weight_grads, (x, dy))
        else:
            dw, db = weight_grads()
        return dx, dw, db, None, None


def conv_transpose3d(x, weight, bias=None):
    """ConvTranspose3d whose kernel equals its stride (each 1 or 2 per dim), padding 0."""
    wp = None if (x.dim() == 5 and ops.convt_k2_ok(x.shape, weight)) else _packed(weight, 2)
    _note_use(weight)
    return _ConvT3dFn.apply(x, weight, bias, wp, _Ref(weight))


conv_transpose3d_k2s2 = conv_transpose3d


class _NormDropActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, rstd, gamma, beta, act_w, conf):
        act, act_p, per_item, drop_p, seed, offset, site, split_exp = conf
        if site is not None:
            out, mask = ops.norm_act_fwd(x, mean, rstd, act, gamma=gamma, beta=beta, act_w=act_w,
                                         act_p=act_p, stats_per_item=per_item, drop_p=drop_p,
                                         seed=seed, rng_offset=offset, want_mask=True,
                                         split_exp=split_exp)
            site.x, site.mean, site.rstd, site.mask = x, mean, rstd, mask
        else:
            out = ops.norm_act_fwd(x, mean, rstd, act, gamma=gamma, beta=beta, act_w=act_w,
                                   act_p=act_p, stats_per_item=per_item, drop_p=drop_p, seed=seed,
                                   rng_offset=offset, split_exp=split_exp)
        ctx.save_for_backward(x, mean, rstd, gamma, beta, act_w)
        ctx.conf = conf
        return out

    @staticmethod
    def backward(ctx, dout):
        x, mean, rstd, gamma, beta, act_w = ctx.saved_tensors
        act, act_p, per_item, drop_p, seed, offset, site, _split_exp = ctx.conf
        if site is not None:
            fused, part, poff = site.fused, site.part, site.poff
            site.fused, site.part = False, None
            lowrank, site.lowrank = site.lowrank, None
            # the site is done: do not pin x / statistics / mask until the graph is freed
            site.x = site.mean = site.rstd = site.mask = None
            if lowrank is not None:
                dx = ops.norm_act_bwd_lowrank(x, lowrank[0], lowrank[1], mean, rstd, act,
                                              act_p=act_p, drop_p=drop_p, seed=seed,
                                              rng_offset=offset)
                return dx, None, None, None, None, None, None
            if fused:
                # dout is dt: the consumer's backward-data epilogue applied act' / dropout and
                # left the two sums of the normalisation's backward in `part`
                dx = ops.norm_act_bwd_from_dt(x, ops.ndhwc(dout), mean, rstd, part, poff)
                return dx, None, None, None, None, None, None
        dact_w = None
        if act_w is not None and ctx.needs_input_grad[5]:
            dact_w = ops.prelu_wgrad(x, dout, mean, rstd, act_w, gamma=gamma, beta=beta,
                                     stats_per_item=per_item, drop_p=drop_p, seed=seed,
                                     rng_offset=offset)
        want_affine = gamma is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        dx, dgamma, dbeta = ops.norm_act_bwd(
            x, dout, mean, rstd, act, gamma=gamma, beta=beta, act_w=act_w, act_p=act_p,
            stats_per_item=per_item, drop_p=drop_p, seed=seed, rng_offset=offset,
            want_affine_grads=want_affine)
        if beta is None:
            dbeta = None
        return dx, None, None, dgamma, dbeta, dact_w, None


def norm_drop_act(x, *, norm="none", eps=1e-5, gamma=None, beta=None, running=None,
                  momentum=0.1, act="identity", act_p=0.0, act_w=None, drop_p=0.0,
                  training=False, rows_reader=None):
    """Fused Norm -> Dropout -> Activation.

    norm: "none" | "instance" | "batch". ``running`` = (running_mean, running_var,
    num_batches_tracked) buffers of a BatchNorm module (updated in training).
    """
    part = getattr(x, "_adell_partials", None)
    x = ops.ndhwc(x)
    N, C = x.shape[:2]
    V = x.shape[2] * x.shape[3] * x.shape[4]
    mean = rstd = None
    per_item = 1
    if norm == "instance":
        if part is None:
            part = ops.channel_partials(x.detach())
        mean, rstd = ops.stats_finalize(part, V, eps, per_item=True)
    elif norm == "batch":
        per_item = 0
        if training or running is None or running[0] is None:
            if part is None:
                part = ops.channel_partials(x.detach())
            mean, rstd = ops.stats_finalize(part, V, eps, per_item=False)
            if training and running is not None and running[0] is not None:
                rm, rv, nbt = running
                if nbt is None and momentum is None:
                    raise ValueError("BatchNorm with momentum=None needs num_batches_tracked")
                ops.bn_running_update(mean, rstd, rm, rv, nbt, N * V, eps, momentum)
        else:
            # running statistics are constants, not functions of x: fold them into the affine
            # pair (y = x * (rstd gamma) + (beta - mean rstd gamma)) so that the backward is the
            # plain elementwise one; [C]-sized parameter algebra, autograd carries dgamma / dbeta
            rstd_c = torch.rsqrt(running[1] + eps)
            gamma = rstd_c if gamma is None else rstd_c * gamma
            shift = -running[0] * gamma
            beta = shift if beta is None else beta + shift
    elif norm != "none":
        raise NotImplementedError(f"norm {norm!r} has no HIP kernel yet")
    p = float(drop_p) if training else 0.0
    seed, offset = 0, 0
    if p > 0.0:
        seed = torch.initial_seed()
        offset = next(_dropout_counter)
    # a site whose only consumer is a conv can leave half of its backward to that conv's
    # backward-data kernel (AdnSite): instance statistics, no affine parameters, an activation the
    # epilogue knows; the forward then also stores its keep bits
    site = None
    if (norm == "instance" and gamma is None and beta is None and act_w is None
            and act in ("identity", "swish", "silu", "relu", "leaky_relu") and x.requires_grad
            and torch.is_grad_enabled() and not FLAGS["no_adn_fuse"] and CONV_PRECISION == "f16x3"
            and ops.norm_act_mask_ok(x)):
        site = AdnSite(None, None, None, None, p, act, float(act_p))
    # (rows_reader: the conv that is the ONLY reader of the output, take_rows_reader)
    split_exp = _rows_exponent(x, rows_reader, norm, gamma, beta, act, float(act_p), act_w, p)
    conf = (act, float(act_p), per_item, p, seed, offset, site, split_exp)
    out = _NormDropActFn.apply(x, mean, rstd, gamma, beta, act_w, conf)
    if site is not None:
        out._adell_site = site
    if split_exp is not None:
        out._adell_rows = ops.SplitRows(split_exp, ops.split_exponents(N, C, split_exp, x.device))
    return out


# ---- token-sequence functions (ViT encoder of UNETR) -------------------------------------
def _rows_as_volume(x):
    """[..., C] contiguous -> logical [1, C, 1, 1, rows] tensor with NDHWC memory (a view)."""
    x = x.contiguous()
    C = x.shape[-1]
    return x.view(1, 1, 1, -1, C).permute(0, 4, 1, 2, 3)


def _volume_as_rows(y, lead_shape):
    """inverse of _rows_as_volume: [1, C, 1, 1, rows] (NDHWC memory) -> [*lead, C]."""
    C = y.shape[1]
    return y.permute(0, 2, 3, 4, 1).reshape(*lead_shape, C)


class _LinearFn(torch.autograd.Function):
    """x W^T (+ bias) (+ residual) on the fp32-MFMA GEMM (ops.gemm). Opt-in
    (``ops.FLAGS["gemm_f16x3"]`` / ADELL_GEMM_F16X3=1, with CONV_PRECISION "f16x3"): the three GEMMs
    on the f16 MFMA with the error-compensated split (ops.gemm_f16x3) whenever the operands
    qualify -- faster per launch, slower per step at the measured configurations."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        N, K = weight.shape
        x2 = x.reshape(-1, K).contiguous()
        rows = x2.shape[0]
        w = weight.contiguous()
        res2 = None if residual is None else residual.reshape(rows, N).contiguous()
        split = CONV_PRECISION == "f16x3" and ops.gemm_f16x3_ok(rows, N, K, x2, K, True, w, K, True)
        if split:
            y = ops.gemm_f16x3(rows, N, K, x2, K, True, w, K, True, bias=bias, residual=res2)
        else:
            y = ops.gemm(rows, N, K, x2, K, True, w, K, True, bias=bias, residual=res2)
        ctx.save_for_backward(x2, w)
        ctx.split = split
        ctx.refs = (_Ref(weight), _Ref(bias))
        ctx.meta = (x.shape, bias is not None, residual is not None and residual.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        xshape, has_bias, res_shape = ctx.meta
        split = ctx.split
        N, K = w.shape
        rows = x2.shape[0]
        need = ctx.needs_input_grad
        dy2 = dy.reshape(rows, N).contiguous()
        dx = dw = db = dres = None
        if need[0]:
            if split and ops.gemm_f16x3_ok(rows, K, N, dy2, N, True, w, K, False):
                dx = ops.gemm_f16x3(rows, K, N, dy2, N, True, w, K, False)
            else:
                dx = ops.gemm(rows, K, N, dy2, N, True, w, K, False)
            dx = dx.view(xshape)
        def weight_grads():
            dw = db = None
            if need[1]:
                if split and ops.gemm_f16x3_ok(N, K, rows, dy2, N, False, x2, K, False):
                    dw = ops.gemm_f16x3(N, K, rows, dy2, N, False, x2, K, False)
                else:
                    dw = ops.gemm(N, K, rows, dy2, N, False, x2, K, False)
            if has_bias and need[2]:
                db = ops.bias_grad(_rows_as_volume(dy2))
            return dw, db

        if (need[1] or (has_bias and need[2])) and _side_ok(ctx.refs[0].obj, ctx.refs[1].obj):
            dw, db = side_run(weight_grads, (x2, dy2))
        else:
            dw, db = weight_grads()
        if res_shape and need[3]:
            dres = dy2.view(res_shape)
        return dx, dw, db, dres


class _MlpFn(torch.autograd.Function):
    """Linear -> activation -> Linear (+ residual) as three + four GEMMs and NO element-wise pass
    (ConvNeXt's pwconv1 -> GELU -> pwconv2 -> + input, res_blocks.py:559-566): the first GEMM's
    epilogue stores the pre-activation (for the backward) and the activation (for the second GEMM)
    in one pass, and the backward's dY W2 GEMM multiplies by act'(pre-activation) in its epilogue,
    so the gradient of the hidden layer is written once. f16x3 GEMMs only (the caller checks
    ``mlp_ok``)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, act, act_p):
        H, K = w1.shape
        N = w2.shape[0]
        x2 = x.reshape(-1, K).contiguous()
        rows = x2.shape[0]
        w1c, w2c = w1.contiguous(), w2.contiguous()
        res2 = None if residual is None else residual.reshape(rows, N).contiguous()
        h, g = ops.gemm_f16x3_act(rows, H, K, x2, K, True, w1c, K, True, act, act_p, bias=b1,
                                  want_act=True)
        y = ops.gemm_f16x3(rows, N, H, g, H, True, w2c, H, True, bias=b2, residual=res2)
        ctx.save_for_backward(x2, w1c, w2c, h, g)
        ctx.act = (act, act_p)
        ctx.refs = (_Ref(w1), _Ref(b1), _Ref(w2), _Ref(b2))
        ctx.meta = (x.shape, b1 is not None, b2 is not None,
                    residual is not None and residual.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, g = ctx.saved_tensors
        xshape, has_b1, has_b2, res_shape = ctx.meta
        H, K = w1.shape
        N = w2.shape[0]
        rows = x2.shape[0]
        need = ctx.needs_input_grad
        dy2 = dy.reshape(rows, N).contiguous()
        # dh = (dy W2) * act'(h): the activation's backward rides the GEMM epilogue
        dh, _ = ops.gemm_f16x3_act(rows, H, N, dy2, N, True, w2, H, False, *ctx.act, dact_in=h)
        dx = None
        if need[0]:
            dx = ops.gemm_f16x3(rows, K, H, dh, H, True, w1, K, False).view(xshape)

        def grads2():
            dw2 = db2 = None
            if need[3]:
                dw2 = ops.gemm_f16x3(N, H, rows, dy2, N, False, g, H, False)
            if has_b2 and need[4]:
                db2 = ops.bias_grad(_rows_as_volume(dy2))
            return dw2, db2

        def grads1():
            dw1 = db1 = None
            if need[1]:
                dw1 = ops.gemm_f16x3(H, K, rows, dh, H, False, x2, K, False)
            if has_b1 and need[2]:
                db1 = ops.bias_grad(_rows_as_volume(dh))
            return dw1, db1

        r1, rb1, r2, rb2 = ctx.refs
        if (need[3] or (has_b2 and need[4])) and _side_ok(r2.obj, rb2.obj):
            dw2, db2 = side_run(grads2, (dy2, g))
        else:
            dw2, db2 = grads2()
        if (need[1] or (has_b1 and need[2])) and _side_ok(r1.obj, rb1.obj):
            dw1, db1 = side_run(grads1, (dh, x2))
        else:
            dw1, db1 = grads1()
        dres = dy2.view(res_shape) if (res_shape and need[5]) else None
        return dx, dw1, db1, dw2, db2, dres, None, None


def mlp_ok(x, w1, w2):
    """Whether ``mlp`` can take Linear(w1) -> act -> Linear(w2) on rows ``x``: all seven GEMMs on the
    f16x3 kernels."""
    if FLAGS["no_mlp_fuse"] or CONV_PRECISION != "f16x3" or not x.is_cuda:
        return False
    H, K = w1.shape
    N = w2.shape[0]
    rows = x.numel() // K
    if rows * H < FLAGS["mlp_min_elems"]:
        return False
    x2 = x.reshape(-1, K)
    ok = ops.gemm_f16x3_ok
    return (w2.shape[1] == H and x2.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()
            and ok(rows, H, K, x2, K, True, w1, K, True) and ok(rows, N, H, x2, H, True, w2, H, True)
            and ok(rows, K, H, x2, H, True, w1, K, False) and ok(rows, H, N, x2, N, True, w2, H, False)
            and ok(N, H, rows, x2, N, False, x2, H, False) and ok(H, K, rows, x2, H, False, x2, K, False))


def mlp(x, w1, b1, w2, b2, act="gelu", act_p=0.0, residual=None):
    """act(x W1^T + b1) W2^T + b2 (+ residual) with the activation inside the GEMM epilogues
    (``act_p``: the slope / alpha of leaky_relu / elu)."""
    _note_use(w1)
    _note_use(w2)
    return _MlpFn.apply(x, w1, b1, w2, b2, residual, act, float(act_p))


def linear(x, weight, bias=None, residual=None):
    """torch.nn.functional.linear on the MFMA GEMMs (+ fused bias / residual add)."""
    _note_use(weight)
    return _LinearFn.apply(x, weight, bias, residual)


def elementwise(x, act="identity", act_p=0.0, drop_p=0.0, training=False):
    """Dropout -> activation on a tensor of any shape (channel axis irrelevant)."""
    shp = x.shape
    y = norm_drop_act(_rows_as_volume(x.reshape(-1, shp[-1])), act=act, act_p=act_p,
                      drop_p=drop_p, training=training)
    return _volume_as_rows(y, shp[:-1])


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        want = gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx, dg, db = ops.layernorm_bwd(x, dy, gamma, mean, rstd, want)
        return dx, dg, db, None


def add(a, b):
    """a + b for same-shape tensors (HIP elementwise kernel)."""
    if a.shape != b.shape:
        raise ValueError("HF.add: shapes differ")
    return _AddBcastFn.apply(a.contiguous(), b.contiguous())


class _LayerNormRowsFn(torch.autograd.Function):
    """rows of C <= 512 values: several rows per wave (csrc/window.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        rows = x.numel() // C
        y, mean, rstd = ops.layernorm_rows_fwd(x, rows, C, 1, C, 0, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        want = gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx = torch.empty_like(x)
        dg, db = ops.layernorm_rows_bwd(x, dy.contiguous(), gamma, mean, rstd, rows, C, 1, C, 0,
                                        dx, C, 0, want)
        return dx, dg, db, None


def layer_norm(x, gamma=None, beta=None, eps=1e-5):
    """torch.nn.LayerNorm over the last dimension."""
    if x.shape[-1] <= 512:
        return _LayerNormRowsFn.apply(x.contiguous(), gamma, beta, float(eps))
    return _LayerNormFn.apply(x.contiguous(), gamma, beta, float(eps))


# ---- gather-based rearranges (SWIN: window partition / merge, cyclic shift, rescale) -----
def _inverse_gather(dims, axes):
    """Spec of the inverse of a bijective gather whose input is dense over ``axes`` (listed
    outermost first): (inverse dims, inverse axes, roll spec or None)."""
    nd = len(dims)
    out_strides = [1] * nd
    for d in range(nd - 2, -1, -1):
        out_strides[d] = out_strides[d + 1] * dims[d + 1][0]
    inv_dims = []
    for a, (extent, _, _) in enumerate(axes):
        feed = sorted([d for d in range(nd) if dims[d][1] == a], key=lambda d: -dims[d][2])
        run = 1
        for d in reversed(feed):
            if dims[d][2] != run:
                raise ValueError("gather is not a mixed-radix bijection")
            run *= dims[d][0]
        if run != extent:
            raise ValueError("gather does not cover its input axis")
        inv_dims += [(dims[d][0], d, 1) for d in feed]
    inv_axes = [(dims[d][0], out_strides[d], 0) for d in range(nd)]
    roll = None
    if any(a[2] != 0 for a in axes):
        roll = ([(a[0], i, 1) for i, a in enumerate(axes)],
                [(a[0], a[1], -a[2]) for a in axes])
    return inv_dims, inv_axes, roll


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dims, axes, out_shape):
        ctx.spec = (dims, axes, x.shape)
        return ops.gather_nd(x, dims, axes).view(out_shape)

    @staticmethod
    def backward(ctx, g):
        dims, axes, in_shape = ctx.spec
        inv_dims, inv_axes, roll = _inverse_gather(dims, axes)
        gi = ops.gather_nd(g.contiguous(), inv_dims, inv_axes)
        if roll is not None:
            gi = ops.gather_nd(gi, *roll)
        return gi.view(in_shape), None, None, None


def _window_spec(shape, nwin, ppw, patch, shift):
    """shift: cyclic shift of the (X, Y, Z, c) axes of the [b, X, Y, Z, c] input."""
    b, X, Y, Z, c = shape
    (w1, w2, w3), (h, w, d), (px, py, pz) = nwin, ppw, patch
    assert (w1 * h * px, w2 * w * py, w3 * d * pz) == (X, Y, Z), "window grid does not tile the image"
    sx, sy, sz, sc = shift
    if sc % c == 0:
        # the innermost image axis and the channel axis are one dense axis of Z*c elements
        axes = [(b, X * Y * Z * c, 0), (X, Y * Z * c, sx), (Y, Z * c, sy), (Z * c, 1, sz * c)]
        dims = [(b, 0, 1), (w1, 1, h * px), (w2, 2, w * py), (w3, 3, d * pz * c), (h, 1, px),
                (w, 2, py), (d, 3, pz * c), (px, 1, 1), (py, 2, 1), (pz * c, 3, 1)]
    else:
        axes = [(b, X * Y * Z * c, 0), (X, Y * Z * c, sx), (Y, Z * c, sy), (Z, c, sz), (c, 1, sc)]
        dims = [(b, 0, 1), (w1, 1, h * px), (w2, 2, w * py), (w3, 3, d * pz), (h, 1, px),
                (w, 2, py), (d, 3, pz), (px, 1, 1), (py, 2, 1), (pz, 3, 1), (c, 4, 1)]
    return dims, axes


def window_partition(x, nwin, ppw, patch, shift=(0, 0, 0, 0)):
    """einops 'b (w1 h x) (w2 w y) (w3 d z) c -> b (w1 w2 w3) (h w d) (x y z c)' of
    torch.roll(x, [-s for s in shift], dims=(1, 2, 3, 4)) -- one gather (vit.py:650-676,
    1197-1203). ``shift`` covers the three image axes and the channel axis."""
    x = x.contiguous()
    shift = tuple(shift) + (0,) * (4 - len(shift))
    dims, axes = _window_spec(x.shape, nwin, ppw, patch, shift)
    b, c = x.shape[0], x.shape[-1]
    out_shape = (b, nwin[0] * nwin[1] * nwin[2], ppw[0] * ppw[1] * ppw[2],
                 patch[0] * patch[1] * patch[2] * c)
    return _GatherFn.apply(x, dims, axes, out_shape)


def window_merge(tokens, image_shape, nwin, ppw, patch):
    """inverse of window_partition without a shift: [b, nW, T, (x y z c)] -> [b, X, Y, Z, c]."""
    tokens = tokens.contiguous()
    dims, axes = _window_spec(image_shape, nwin, ppw, patch, (0, 0, 0, 0))
    inv_dims, inv_axes, _ = _inverse_gather(dims, axes)
    return _GatherFn.apply(tokens, inv_dims, inv_axes, tuple(image_shape))


def space_to_depth(x, scale):
    """einops_rescale (vit.py:33-45): 'b c (h p1) (w p2) (d p3) -> b (c p1 p2 p3) h w d' for an
    NDHWC activation; returns the logical [b, c*p1*p2*p3, h, w, d] tensor (NDHWC memory)."""
    x = ops.ndhwc(x)
    b, c, H, W, D = x.shape
    p1, p2, p3 = scale
    h, w, d = H // p1, W // p2, D // p3
    axes = [(b, H * W * D * c, 0), (H, W * D * c, 0), (W, D * c, 0), (D, c, 0), (c, 1, 0)]
    dims = [(b, 0, 1), (h, 1, p1), (w, 2, p2), (d, 3, p3), (c, 4, 1), (p1, 1, 1), (p2, 2, 1),
            (p3, 3, 1)]
    xr = x.permute(0, 2, 3, 4, 1)  # [b, H, W, D, c], contiguous
    out = _GatherFn.apply(xr, dims, axes, (b, h, w, d, c * p1 * p2 * p3))
    return out.permute(0, 4, 1, 2, 3)


def conv3d_dilated(x, weight, bias, rate):
    """``Conv3d(kernel_size=3, dilation=rate, padding="same")`` (stride 1) -- the dilated convs of the
    atrous pyramid (multi_resolution.py:392-403). A tap of a dilated kernel reads
    ``x[o + (t - 1) rate]``: output voxel ``o = rate q + p`` only ever meets the sub-lattice of voxels
    congruent to ``p`` (per axis), on which the operation is the PLAIN 3x3x3 convolution with
    padding 1. So: one gather (space-to-batch: the rate^3 sub-lattices become batch items), the
    ordinary conv kernels on a batch of N rate^3 volumes, one inverse gather. Extents that are not
    multiples of ``rate`` are zero-framed at the far end first (``adell_window_ndhwc``; the frame is
    what the conv's own zero padding would be) and the result is cropped back."""
    rate = tuple(int(r) for r in ops._triple(rate))
    x = ops.ndhwc(x)
    if rate == (1, 1, 1):
        return conv3d(x, weight, bias, 1, 1)
    if tuple(weight.shape[2:]) != (3, 3, 3):
        raise NotImplementedError("dilated convolutions: 3x3x3 kernels only")
    N, C, D, H, W = x.shape
    full = tuple(-(-n // r) * r for n, r in zip((D, H, W), rate))
    if full != (D, H, W):
        x = _CropFn.apply(x, full, (0, 0, 0))
    De, He, We = full
    sub = tuple(n // r for n, r in zip(full, rate))

    def spec(ch):
        axes = [(N, De * He * We * ch, 0), (De, He * We * ch, 0), (He, We * ch, 0), (We, ch, 0), (ch, 1, 0)]
        dims = [(N, 0, 1)]
        dims += [(r, a + 1, 1) for a, r in enumerate(rate) if r > 1]                       # the phases
        dims += [(n, a + 1, r) for a, (n, r) in enumerate(zip(sub, rate))] + [(ch, 4, 1)]  # the sub-lattice
        return dims, axes

    phases = rate[0] * rate[1] * rate[2]
    dims, axes = spec(C)
    xb = _GatherFn.apply(x.permute(0, 2, 3, 4, 1), dims, axes, (N * phases, *sub, C))
    yb = conv3d(xb.permute(0, 4, 1, 2, 3), weight, bias, 1, 1, want_stats=False)
    Cout = weight.shape[0]
    odims, oaxes = spec(Cout)
    inv_dims, inv_axes, _ = _inverse_gather(odims, oaxes)
    y = _GatherFn.apply(ops.ndhwc(yb).permute(0, 2, 3, 4, 1).contiguous(), inv_dims, inv_axes,
                        (N, De, He, We, Cout)).permute(0, 4, 1, 2, 3)
    if full != (D, H, W):
        y = _CropFn.apply(y, (D, H, W), (0, 0, 0))
    return y


class _WindowAttnFn(torch.autograd.Function):
    """q-norm, k-norm and windowed attention on a QKV buffer [tokens, H*(2a+hd)] whose heads
    are laid out q | k | v (linear_blocks.py:369-417). Nothing is sliced or permuted: the
    LayerNorm kernel reads the q / k slices in place, the attention kernel reads v in place,
    and the backward writes the three gradients straight into dQKV."""

    @staticmethod
    def forward(ctx, qkv, qg, qb, kg, kb, rel, mask, conf):
        W, H, T, a, hd, scale, drop_p, seed, offset, eps = conf
        per = 2 * a + hd
        ts = H * per
        flat = qkv.view(-1)
        rows = W * T * H
        qn, qm, qr = ops.layernorm_rows_fwd(flat, rows, a, H, ts, per, qg, qb, eps)
        kn, km, kr = ops.layernorm_rows_fwd(flat[a:], rows, a, H, ts, per, kg, kb, eps)
        o, lse = ops.winattn_fwd(qn, kn, flat[2 * a:], ts, per, rel, mask, W, H, T, a, hd, scale,
                                 drop_p, seed, offset)
        ctx.save_for_backward(qkv, qg, kg, qn, kn, qm, qr, km, kr, o, lse, rel, mask)
        ctx.conf = conf
        return o.view(W * T, H * hd)

    @staticmethod
    def backward(ctx, do):
        qkv, qg, kg, qn, kn, qm, qr, km, kr, o, lse, rel, mask = ctx.saved_tensors
        W, H, T, a, hd, scale, drop_p, seed, offset, eps = ctx.conf
        per = 2 * a + hd
        ts = H * per
        rows = W * T * H
        need = ctx.needs_input_grad
        flat = qkv.view(-1)
        dqkv = torch.empty_like(qkv)
        dflat = dqkv.view(-1)
        dqn, dkn = torch.empty_like(qn), torch.empty_like(kn)
        ds = ops.winattn_bwd(qn, kn, flat[2 * a:], ts, per, rel, mask, o, do.contiguous(), lse, W,
                             H, T, a, hd, scale, drop_p, seed, offset, dqn, dkn, dflat[2 * a:],
                             rel is not None and need[5])
        dqg, dqb = ops.layernorm_rows_bwd(flat, dqn, qg, qm, qr, rows, a, H, ts, per, dflat, ts,
                                          per, need[1] or need[2])
        dkg, dkb = ops.layernorm_rows_bwd(flat[a:], dkn, kg, km, kr, rows, a, H, ts, per,
                                          dflat[a:], ts, per, need[3] or need[4])
        drel = None
        if ds is not None:
            drel = ops.bias_grad(_rows_as_volume(ds)).view(rel.shape)
        return dqkv, dqg, dqb, dkg, dkb, drel, None, None


def window_attention(qkv, q_gamma, q_beta, k_gamma, k_beta, n_windows, n_heads, tokens, a, hd,
                     rel=None, mask=None, drop_p=0.0, training=False, eps=1e-5):
    """qkv: [n_windows * tokens, n_heads * (2a + hd)] -> [n_windows * tokens, n_heads * hd]."""
    p = float(drop_p) if training else 0.0
    seed, offset = 0, 0
    if p > 0.0:
        seed = torch.initial_seed()
        offset = next(_dropout_counter)
    conf = (int(n_windows), int(n_heads), int(tokens), int(a), int(hd), 1.0 / (a ** 0.5), p, seed,
            offset, float(eps))
    rel = None if rel is None else rel.contiguous()
    mask = None if mask is None else mask.contiguous()
    return _WindowAttnFn.apply(qkv.contiguous(), q_gamma, q_beta, k_gamma, k_beta, rel, mask, conf)


class _SeqAttnFn(torch.autograd.Function):
    """q-norm, k-norm and full-sequence attention on the projection output [B*T, H*(2a+hd)]
    whose heads are laid out q | k | v (linear_blocks.py:372-417), for the MFMA-shaped heads.
    As in the windowed form nothing is sliced or permuted: the LayerNorm kernel reads the q / k
    slices in place (token-major outputs), the attention kernels address every operand by
    (item, head, row) strides -- V inside the projection, O as [B, T, H*hd] token rows -- and the
    backward writes the three gradients straight into dQKV."""

    @staticmethod
    def _strides(T, H, a, hd):
        per = 2 * a + hd
        tok = (T * H * a, a, H * a)            # [B, T, H, a] token-major q / k (and dq / dk)
        pak = (T * H * per, per, H * per)      # rows inside the packed projection
        out = (T * H * hd, hd, H * hd)         # [B, T, H * hd]
        return tok, pak, out

    @staticmethod
    def forward(ctx, qkv, qg, qb, kg, kb, bias, conf):
        B, H, T, a, hd, scale, drop_p, seed, offset, eps = conf
        per = 2 * a + hd
        flat = qkv.view(-1)
        rows = B * T * H
        qn, qm, qr = ops.layernorm_rows_fwd(flat, rows, a, 1, per, 0, qg, qb, eps)
        kn, km, kr = ops.layernorm_rows_fwd(flat[a:], rows, a, 1, per, 0, kg, kb, eps)
        tok, pak, out = _SeqAttnFn._strides(T, H, a, hd)
        o = torch.empty((B * T, H * hd), device=qkv.device, dtype=torch.float32)
        lse = ops.attention_fwd_strided(qn, kn, flat[2 * a:], o, tok + tok + pak + out, bias, B, H, T,
                                        a, hd, scale, drop_p, seed, offset)
        ctx.save_for_backward(qkv, qg, kg, qn, kn, qm, qr, km, kr, o, lse, bias)
        ctx.conf = conf
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, qg, kg, qn, kn, qm, qr, km, kr, o, lse, bias = ctx.saved_tensors
        B, H, T, a, hd, scale, drop_p, seed, offset, eps = ctx.conf
        if bias is not None and ctx.needs_input_grad[5]:
            raise NotImplementedError("attention bias gradient of the full-sequence form")
        per = 2 * a + hd
        rows = B * T * H
        need = ctx.needs_input_grad
        flat = qkv.view(-1)
        dqkv = torch.empty_like(qkv)
        dflat = dqkv.view(-1)
        dqn, dkn = torch.empty_like(qn), torch.empty_like(kn)
        tok, pak, out = _SeqAttnFn._strides(T, H, a, hd)
        ops.attention_bwd_strided(qn, kn, flat[2 * a:], o, do.contiguous(), lse, dqn, dkn,
                                  dflat[2 * a:], tok + tok + pak + out + out + tok + tok + pak, bias,
                                  B, H, T, a, hd, scale, drop_p, seed, offset)
        dqg, dqb = ops.layernorm_rows_bwd(flat, dqn, qg, qm, qr, rows, a, 1, per, 0, dflat, per, 0,
                                          need[1] or need[2])
        dkg, dkb = ops.layernorm_rows_bwd(flat[a:], dkn, kg, km, kr, rows, a, 1, per, 0, dflat[a:],
                                          per, 0, need[3] or need[4])
        return dqkv, dqg, dqb, dkg, dkb, None, None


def seq_attention_ok(tokens, a, hd):
    """Whether ``seq_attention`` covers these head sizes (otherwise: slices + ``attention``)."""
    return (not FLAGS["no_seq_attention"] and a % 4 == 0 and hd % 4 == 0
            and ops.attention_strided_ok(tokens, a, hd))


def seq_attention(qkv, q_gamma, q_beta, k_gamma, k_beta, batch, n_heads, tokens, a, hd, bias=None,
                  drop_p=0.0, training=False, eps=1e-5):
    """qkv: [batch * tokens, n_heads * (2a + hd)] -> [batch * tokens, n_heads * hd]; bias
    [nbias, tokens, tokens] or None is added to the scores of sequence (b, h) as
    bias[(b * n_heads + h) % nbias]."""
    p = float(drop_p) if training else 0.0
    seed, offset = 0, 0
    if p > 0.0:
        seed = torch.initial_seed()
        offset = next(_dropout_counter)
    conf = (int(batch), int(n_heads), int(tokens), int(a), int(hd), 1.0 / (a ** 0.5), p, seed,
            offset, float(eps))
    bias = None if bias is None else bias.contiguous()
    return _SeqAttnFn.apply(qkv.contiguous(), q_gamma, q_beta, k_gamma, k_beta, bias, conf)


class _AddBcastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.bshape = b.shape
        return ops.add_bcast(a, b)

    @staticmethod
    def backward(ctx, g):
        db = None
        if ctx.needs_input_grad[1]:
            same = int(np.prod(ctx.bshape)) == g.numel()
            db = g.reshape(ctx.bshape) if same else ops.sum_bcast(g, ctx.bshape)
        return g, db


def add_bcast(a, b):
    """a + b with b broadcast over a's leading dimensions (positional embedding)."""
    return _AddBcastFn.apply(a, b)


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, bias, scale, drop):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        if bias is not None:
            # an expanded mask ([1, 1, T, T] -> heads) reshapes to a stride-0 VIEW: the backward
            # must see the same materialised rows the forward used
            bias = bias.contiguous()
        out, lse = ops.attention_fwd(q, k, v, bias, scale, *drop)
        ctx.save_for_backward(q, k, v, bias, out, lse)
        ctx.scale, ctx.drop = scale, drop
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, bias, out, lse = ctx.saved_tensors
        if bias is not None and ctx.needs_input_grad[3]:
            raise NotImplementedError("attention bias gradient (SWIN relative positions): next row")
        dq, dk, dv = ops.attention_bwd(q, k, v, bias, out, dout, lse, ctx.scale, *ctx.drop)
        return dq, dk, dv, None, None, None


def attention(q, k, v, bias=None, scale=None, drop_p=0.0, training=False):
    """dropout(softmax(q k^T * scale + bias)) v for q,k [BH,T,A], v [BH,T,Dv]."""
    if scale is None:
        scale = 1.0 / (q.shape[-1] ** 0.5)
    drop = (0.0, 0, 0)
    if training and drop_p > 0:
        drop = (float(drop_p), torch.initial_seed(), next(_dropout_counter))
    return _AttentionFn.apply(q, k, v, bias, float(scale), drop)


# ---- data movement with autograd (U-Net++ dense links) ---------------------------------------
class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *tensors):
        ctx.sizes = [t.shape[1] for t in tensors]
        return ops.cat_channels(list(tensors))

    @staticmethod
    def backward(ctx, g):
        return tuple(ops.split_channels(g, ctx.sizes))


def cat_channels(tensors):
    """torch.cat(tensors, 1) for [N,C,D,H,W] activations, staying NDHWC ([N,C,H,W]: depth-1
    volumes)."""
    tensors = list(tensors)
    if len(tensors) == 1:
        return tensors[0]
    if tensors[0].dim() == 4:
        return _CatFn.apply(*[t.unsqueeze(2) for t in tensors]).squeeze(2)
    return _CatFn.apply(*tensors)


def cat_tokens(a, b):
    """torch.cat([a, b], 1) for token tensors [B, T, E] (class token / registers in front of the
    patch tokens, vit.py:871-880): per item the rows are contiguous, so it is the channel concat of
    depth-1 volumes with T * E "channels"."""
    B, E = a.shape[0], a.shape[2]
    y = _CatFn.apply(a.reshape(B, -1, 1, 1, 1), b.reshape(B, -1, 1, 1, 1))
    return y.reshape(B, a.shape[1] + b.shape[1], E)


class _NearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.in_size = tuple(x.shape[2:])
        return ops.interp_nearest(x, size)

    @staticmethod
    def backward(ctx, g):
        return ops.interp_nearest(g, None, backward_from=ctx.in_size), None


def interpolate_nearest(x, size):
    """F.interpolate(x, size) (default mode='nearest') for 5-D activations."""
    size = tuple(int(s) for s in size)
    if tuple(x.shape[2:]) == size:
        return x
    if x.dim() == 4:
        return _NearestFn.apply(x.unsqueeze(2), (1, *size)).squeeze(2)
    return _NearestFn.apply(x, size)


class _LinearUpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scales):
        ctx.in_size, ctx.scales = tuple(x.shape[2:]), scales
        return ops.interp_linear(x, scales)

    @staticmethod
    def backward(ctx, g):
        return ops.interp_linear(g, ctx.scales, backward_from=ctx.in_size), None


def upsample_linear(x, scale_factor):
    """torch.nn.Upsample(scale_factor, mode="bilinear" (4-D) / "trilinear" (5-D),
    align_corners=False) as used by the "upsample" upscaling path (unet.py:419-443)."""
    nd = x.dim() - 2
    sc = (float(scale_factor),) * nd if not isinstance(scale_factor, (tuple, list)) else \
        tuple(float(s) for s in scale_factor)
    if nd == 2:
        return _LinearUpFn.apply(x.unsqueeze(2), (1.0,) + sc).squeeze(2)
    return _LinearUpFn.apply(x, sc)


def resize_linear_aligned(x, size):
    """F.interpolate(x, size, mode="linear"-family, align_corners=True) for [N, C, *spatial]
    tensors that need no gradient (the deep-supervision targets, pl.py:305-309)."""
    size = [int(v) for v in size]
    x = x.detach()
    nd = x.dim() - 2
    if nd == 3:
        return ops.interp_linear(x, None, size=tuple(size))
    if nd == 2:
        return ops.interp_linear(x.unsqueeze(2), None, size=(1, *size)).squeeze(2)
    return ops.interp_linear(x[:, :, None, None], None, size=(1, 1, *size))[:, :, 0, 0]


class _RowScaleFn(torch.autograd.Function):
    """(gamma[:, None] * W, gamma * b) with its backward, one launch each way (adell_rowscale_*)."""

    @staticmethod
    def forward(ctx, gamma, W, b):
        gamma, W = gamma.contiguous(), W.contiguous()
        C, K = W.shape
        W2 = torch.empty_like(W)
        b2 = None if b is None else torch.empty_like(b)
        ops.check(_lib.lib().adell_rowscale_fwd(ops._ptr(gamma), ops._ptr(W), ops._ptr(b), ops._ptr(W2),
                                                ops._ptr(b2), C, K, ops._stream()))
        ctx.save_for_backward(gamma, W, b)
        return W2, b2

    @staticmethod
    def backward(ctx, dW2, db2):
        gamma, W, b = ctx.saved_tensors
        C, K = W.shape
        dW2 = dW2.contiguous()
        if b is not None and db2 is None:
            db2 = torch.zeros_like(b)
        dgamma, dW = torch.empty_like(gamma), torch.empty_like(W)
        db = None if b is None else torch.empty_like(b)
        ops.check(_lib.lib().adell_rowscale_bwd(
            ops._ptr(gamma), ops._ptr(W), ops._ptr(b), ops._ptr(dW2),
            None if db2 is None else ops._ptr(db2.contiguous()), ops._ptr(dgamma), ops._ptr(dW),
            ops._ptr(db), C, K, ops._stream()))
        return dgamma, dW, db


class _CropFn(torch.autograd.Function):
    """Centre window of a volume as a dense tensor, and a zero frame around the gradient: one launch
    each way (adell_window_ndhwc) where slicing + ``contiguous`` + ``slice_backward`` ran a strided
    copy, three zero-fills and four more strided copies."""

    @staticmethod
    def forward(ctx, x, out_size, offset):
        x = ops.ndhwc(x)
        ctx.in_size, ctx.offset = tuple(x.shape[2:]), tuple(offset)
        return ops.window_ndhwc(x, out_size, offset)

    @staticmethod
    def backward(ctx, dy):
        dy = ops.ndhwc(dy)
        return ops.window_ndhwc(dy, ctx.in_size, tuple(-o for o in ctx.offset)), None, None


def crop3d(x, out_size):
    """crop_to_size of a [N, C, D, H, W] volume (layers/utils.py:30-52: offset ``diff // 2`` per axis)."""
    ops._require_cuda(x)
    offset = [(cur - out) // 2 for cur, out in zip(x.shape[2:], out_size)]
    return _CropFn.apply(x, tuple(int(v) for v in out_size), tuple(offset))


def rowscale(gamma, W, b=None):
    """Layer scale folded into a Linear layer: (gamma[:, None] * W, gamma * b) for W [C, K]."""
    ops._require_cuda(gamma, W, b)
    return _RowScaleFn.apply(gamma, W, b)


class _ScaleBcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return ops.scale_bc(x, s)

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dx = ops.scale_bc(dy, s) if ctx.needs_input_grad[0] else None
        ds = ops.scale_bc_dscale(x, dy) if ctx.needs_input_grad[1] else None
        return dx, ds


def scale_per_item_channel(x, s):
    """x [N, C, *spatial] times s [N, C] broadcast over the spatial axes (feature gates of the
    decoder, unet.py:803-810; U-out, regularization.py:48-55)."""
    nd = x.dim()
    if nd == 5:
        return _ScaleBcFn.apply(x, s)
    if nd == 4:
        return _ScaleBcFn.apply(x.unsqueeze(2), s).squeeze(2)
    if nd == 3:
        return _ScaleBcFn.apply(x.unsqueeze(2).unsqueeze(2), s).squeeze(2).squeeze(2)
    if nd == 2:
        return _ScaleBcFn.apply(x[:, :, None, None, None], s)[:, :, 0, 0, 0]
    raise ValueError(f"scale_per_item_channel: [N, C, ...] with <= 3 spatial dims, got {tuple(x.shape)}")


class _ChannelMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        mean, _ = ops.instance_stats(x)
        return mean

    @staticmethod
    def backward(ctx, g):
        V = int(np.prod(ctx.shape[2:]))
        return ops.bcast_nc(g, ctx.shape, 1.0 / V)


def channel_mean(x):
    """[N, C, D, H, W] -> [N, C]: mean over the voxels (torch.flatten(X, 2).mean(-1))."""
    return _ChannelMeanFn.apply(x)


class _CseApplyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, c, inv, acc):
        ctx.save_for_backward(x, s, c, inv)
        return ops.cse_apply(x, s, c, inv, acc)

    @staticmethod
    def backward(ctx, dy):
        x, s, c, inv = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx = ds = dc = None
        if need[0] or need[1] or need[2]:
            dx, ds, dc = ops.cse_apply_bwd(x, dy, s, c, inv)
        return (dx if need[0] else None, ds if need[1] else None, dc if need[2] else None, None,
                dy if need[4] else None)


def cse_apply(x, s, c, inv=None, acc=None):
    """acc + x * (s + c) * inv: the concurrent squeeze-and-excite gate (spatial gate s [N,1,D,H,W],
    channel gate c [N,C]) with the branch sum and the division by the summed branch weights of
    BrUNet.forward (unet.py:1186-1207) folded in. inv [N] carries no gradient."""
    return _CseApplyFn.apply(x, s, c, inv, acc)


class _MaxPool3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        y, idx = ops.maxpool3d_fwd(x, kernel, stride, padding)
        ctx.save_for_backward(idx)
        ctx.conf = (tuple(x.shape), kernel, stride, padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, kernel, stride, padding = ctx.conf
        return ops.maxpool3d_bwd(dy, idx, shape, kernel, stride, padding), None, None, None


def max_pool3d(x, kernel, stride=None, padding=0):
    """torch.nn.functional.max_pool3d (ceil_mode=False, dilation=1)."""
    kernel = ops._triple(kernel)
    stride = kernel if stride is None else ops._triple(stride)
    return _MaxPool3dFn.apply(x, kernel, stride, ops._triple(padding))


# ---- ConvNeXt / VICReg ---------------------------------------------------------------------------
class _DwConv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        y = ops.dwconv3d_fwd(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx = dw = db = None
        if need[0]:
            dx = ops.dwconv3d_bwd_data(dy, weight)
        if need[1] or (ctx.has_bias and need[2]):
            dw, db = ops.dwconv3d_bwd_weight(x, dy, tuple(weight.shape[2:]), ctx.has_bias)
        return dx, dw, db


def dwconv3d(x, weight, bias=None):
    """Depthwise Conv3d (groups = channels, stride 1, 'same' padding)."""
    return _DwConv3dFn.apply(x, weight, bias)


def channel_scale(x, gamma):
    """gamma[c] * x for a 5-D activation (layer scale of the ConvNeXt block)."""
    return norm_drop_act(x, norm="none", gamma=gamma)


class _VICRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, min_var, eps):
        x1, x2 = x1.contiguous(), x2.contiguous()
        out, scratch = ops.vicreg_fwd(x1, x2, min_var, eps)
        ctx.save_for_backward(x1, x2, scratch)
        ctx.conf = (min_var, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        x1, x2, scratch = ctx.saved_tensors
        dx1, dx2 = ops.vicreg_bwd(x1, x2, scratch, *ctx.conf, g,
                                  ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx1, dx2, None, None


def vicreg_terms(x1, x2, min_var=1.0, eps=1e-4):
    """(invariance, variance, covariance) terms of VICReg, unweighted, as a 3-vector."""
    return _VICRegFn.apply(x1, x2, float(min_var), float(eps))


class _PairLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, kind, temperature, apply_relu):
        x1, x2 = x1.contiguous(), x2.contiguous()
        loss, scratch = ops.pair_loss_fwd(x1, x2, kind, temperature, apply_relu)
        ctx.save_for_backward(x1, x2, scratch)
        ctx.conf = (kind, temperature, apply_relu)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x1, x2, scratch = ctx.saved_tensors
        dx1, dx2 = ops.pair_loss_bwd(x1, x2, *ctx.conf, scratch, g, ctx.needs_input_grad[0],
                                     ctx.needs_input_grad[1])
        return dx1, dx2, None, None, None


def pair_loss(x1, x2, kind, temperature=1.0, apply_relu=False):
    """Scalar cosine-similarity loss between two [B, D] embedding batches: ``kind`` = "simsiam",
    "byol" (self_supervised/losses/functional.py:138-164) or "ntxent" (losses/ntxent.py:11-46)."""
    return _PairLossFn.apply(x1, x2, kind, float(temperature), bool(apply_relu))


class _LocoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, temperature, eps):
        f1, f2 = ops.ndhwc(f1), ops.ndhwc(f2)
        loss = ops.loco_loss_fwd(f1, f2, temperature, eps)
        ctx.save_for_backward(f1, f2)
        ctx.conf = (temperature, eps)
        return loss

    @staticmethod
    def backward(ctx, g):
        f1, f2 = ctx.saved_tensors
        df1, df2 = ops.loco_loss_bwd(f1, f2, g, *ctx.conf, ctx.needs_input_grad[0],
                                     ctx.needs_input_grad[1])
        return df1, df2, None, None


def loco_loss(f1, f2, temperature=0.1, eps=1e-8):
    """Per-item local contrastive loss [B] between the features of two views [B, C, *spatial]
    (LocalContrastiveLoss.forward, semi_supervised_segmentation/losses.py:498-526)."""
    return _LocoLossFn.apply(f1, f2, float(temperature), float(eps))


class _ChannelSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.channel_softmax_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.channel_softmax_bwd(y, dy)


def channel_softmax(x):
    """torch.nn.Softmax(dim=1) on a [N, C, *spatial] activation (the n_classes > 2 head,
    unet.py:641-655); 4-D inputs are depth-1 volumes."""
    if x.dim() == 4:
        return _ChannelSoftmaxFn.apply(x.unsqueeze(2)).squeeze(2)
    return _ChannelSoftmaxFn.apply(x)


class _ChannelMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        out, arg = ops.channel_max_fwd(x)
        ctx.save_for_backward(arg)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return ops.channel_max_bwd(g, arg, ctx.shape)


def channel_max(x):
    """X.flatten(start_dim=2).max(-1).values for a [N, C, *spatial] activation (the pooling in
    front of the bottleneck classifier, unet.py:826-828)."""
    if x.dim() == 4:
        return _ChannelMaxFn.apply(x.unsqueeze(2))
    return _ChannelMaxFn.apply(x)
