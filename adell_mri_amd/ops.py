"""Raw (non-autograd) Python front-ends of the HIP kernels.

Activations are torch CUDA tensors of logical shape ``[N, C, D, H, W]`` whose
memory is dense NDHWC (``torch.channels_last_3d``); ``ndhwc()`` / ``new_act()``
are the only places that care about strides. Everything is enqueued on
``torch.cuda.current_stream()``; PyTorch only provides memory and streams.
"""
import ctypes
import math
import os

import torch

from . import _lib
from ._lib import ACT_IDS, AdellHipError, ConvDesc, NormActDesc, check


class KernelTimer:
    """Per-launch HIP-event timing of the MFMA kernels (bench.py's roofline leg).

    Events are recorded on ``torch.cuda.current_stream()``, the stream the kernels
    are enqueued on; nothing synchronises until ``summary()`` is read.
    """

    def __init__(self, only=None):
        self.records = []
        self._agg = None
        self.only = None if only is None else set(only)  # kernel families to time (None: all)
        # an event pair costs ~6 us of stream idle time per launch (measured: rocprofv3 timeline,
        # DESIGN.md 8); bench.py switches the timer off for the steps it does not sample
        self.active = True

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, flops, e0, tag=None, nbytes=0.0, kernels=1):
        """``kernels``: device launches of the named kernel behind this call (8 for the
        parity-class backward-data), so that ms / launches is a per-kernel average."""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((name, float(flops), e0, e1, tag, float(nbytes), int(kernels)))

    def by_tag(self):
        """{(kernel, tag): {flops, ms, launches, tflops}} -- per-layer view (tools/)."""
        torch.cuda.synchronize()
        agg = {}
        for name, flops, e0, e1, tag, _nb, _k in self.records:
            a = agg.setdefault((name, tag), [0.0, 0.0, 0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        return {k: {"flops": v[0], "ms": v[1], "launches": v[2],
                    "tflops": v[0] / max(v[1], 1e-9) / 1e9} for k, v in agg.items()}

    def summary(self):
        if self._agg is None:
            torch.cuda.synchronize()
            agg = {}
            for name, flops, e0, e1, _tag, nb, k in self.records:
                a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
                a[0] += flops
                a[1] += e0.elapsed_time(e1)
                a[2] += k
                a[3] += nb
            self._agg = {k: {"flops": v[0], "ms": v[1], "launches": v[2],
                             "tflops": v[0] / max(v[1], 1e-9) / 1e9,
                             "algorithmic_bytes": v[3]} for k, v in agg.items()}
        return self._agg

    def dominant(self):
        """The kernel family with the most time among those that carry FLOPs (the HBM-bound
        families are recorded with flops = 0 and reported through their bytes)."""
        s = {k: v for k, v in self.summary().items() if v["flops"] > 0}
        if not s:
            return None
        name = max(s, key=lambda k: s[k]["ms"])
        return name, s[name]["flops"], s[name]["ms"], s[name]["launches"]

    def share(self, name, total_ms):
        return self.summary()[name]["ms"] / max(total_ms, 1e-9)


KERNEL_TIMER = None
# dispatch switches for A/B tests (read from the environment once at import)
FLAGS = {"no_cinfold": bool(os.environ.get("ADELL_NO_CINFOLD")),
         "no_convt_k2": bool(os.environ.get("ADELL_NO_CONVT_K2")),
         "no_grad_carry": bool(os.environ.get("ADELL_NO_GRAD_CARRY")),
         "no_skip_fork": bool(os.environ.get("ADELL_NO_SKIP_FORK")),
         "no_s2fused": bool(os.environ.get("ADELL_NO_S2FUSED")),
         # Linear layers on the f16x3 GEMM (1.2-2x the fp32-MFMA GEMM per launch) when both output
         # sides are >= 64: measured per step (bench.py secondary, alternating runs on one box) UNETR
         # 22.6 -> 21.8 ms, VICReg ConvNeXt 27.7 -> 24.8 ms, SWIN-UNet 198 -> 197.5 ms; with every
         # size SWIN's narrow projections (24 ... 48 wide) lose 1 % (round 3; round 5: 32-wide layers
         # over >= 64 k rows are streaming problems and take it too, gemm_f16x3_ok). ADELL_GEMM_F16X3=0: fp32-MFMA
         # GEMMs everywhere; =1: every applicable size (ADELL_GEMM_F16X3_MIN_K / _MIN_MN: knobs).
         "gemm_f16x3": os.environ.get("ADELL_GEMM_F16X3", "auto") != "0",
         "gemm_f16x3_min_k": int(os.environ.get("ADELL_GEMM_F16X3_MIN_K", "0")),
         "gemm_f16x3_min_mn": int(os.environ.get(
             "ADELL_GEMM_F16X3_MIN_MN", "64" if os.environ.get("ADELL_GEMM_F16X3", "auto") == "auto" else "0")),
         "no_cin_small": bool(os.environ.get("ADELL_NO_CIN_SMALL")),
         "cin_small_all": bool(os.environ.get("ADELL_CIN_SMALL_ALL"))}
NORM_ACT_FAMILY = "adell_norm_act_kernels"   # norm -> dropout -> activation, forward + backward


def _timed(name, flops, fn, tag=None, nbytes=0.0, kernels=1):
    t = KERNEL_TIMER
    if t is None or not t.active or (t.only is not None and name not in t.only):
        return fn()
    e0 = t.start()
    rc = fn()
    t.stop(name, flops, e0, tag() if callable(tag) else tag, nbytes, kernels)
    return rc


def _conv_bytes(d, residual=False):
    """Algorithmic HBM bytes of one conv launch (any direction): the input volume(s), the
    output volume (+ residual) and the weights, each touched once, fp32."""
    vin = d.N * d.D * d.H * d.W * (d.C0 + d.C1)
    vout = d.N * d.Do * d.Ho * d.Wo * d.Cout
    w = d.Cout * (d.C0 + d.C1) * d.KD * d.KH * d.KW
    return 4.0 * (vin + vout * (2 if residual else 1) + w)


def _conv_tag(d, kind):
    return lambda: (f"{kind} {d.C0 + d.C1}->{d.Cout} in {d.D}x{d.H}x{d.W} k{d.KD}{d.KH}{d.KW} "
                    f"s{d.SD}{d.SH}{d.SW}")


def _conv_flops(d):
    return 2.0 * d.N * d.Do * d.Ho * d.Wo * d.Cout * (d.C0 + d.C1) * d.KD * d.KH * d.KW


def _triple(v):
    if isinstance(v, (int,)):
        return (int(v),) * 3
    v = tuple(int(i) for i in v)
    if len(v) == 1:
        return v * 3
    assert len(v) == 3, v
    return v


_raw_stream = (None if os.environ.get("ADELL_TORCH_STREAM_API")
               else getattr(torch._C, "_cuda_getCurrentRawStream", None))


def _stream():
    """The current HIP stream of the current device as a raw handle (the private fast path
    when this torch build has it: ~0.3 us instead of ~10 us per launch)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_CHECK_DENSE = bool(os.environ.get("ADELL_CHECK_DENSE"))


def _ptr(t):
    """Device address of ``t``. The kernels index dense memory (row-major, or NDHWC for 5-D / NHWC
    for 4-D activations). Always checked: an expanded view (a stride-0 dimension of size > 1) raises
    -- a kernel would read ``numel`` elements from a storage that holds fewer (the masked-attention
    backward at batch 1 did, round 4). With ADELL_CHECK_DENSE=1 (tests/conftest.py sets it) every
    other non-dense layout -- a slice with gaps, a permuted view -- raises too."""
    if t is None:
        return None
    st = t.stride()
    if 0 in st:
        for s, n in zip(st, t.shape):
            if s == 0 and n > 1:
                raise _lib.AdellHipError(f"expanded (stride-0) tensor handed to a kernel: shape "
                                         f"{tuple(t.shape)}, strides {st}")
    if _CHECK_DENSE and not (t.is_contiguous()
                             or (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d))
                             or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        raise _lib.AdellHipError(f"non-dense tensor handed to a kernel: shape {tuple(t.shape)}, "
                                 f"strides {st}")
    return ctypes.c_void_p(t.data_ptr())


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AdellHipError(
                "adell_mri_amd kernels run on MI355X only: got a CPU tensor (no CPU fallback)")
        if t is not None and t.dtype != torch.float32:
            raise _lib.AdellHipError(f"adell_mri_amd kernels are fp32; got {t.dtype}")


def rng_advance(delta, set_value=False):
    """The library's replay counter of the dropout offsets (adell_rng_advance): add ``delta`` to the
    device word every dropout kernel adds to its offset argument, or set it. Stream-ordered; the
    last node of a captured training step (trainer.StepRunner.enable_graph)."""
    check(_lib.lib().adell_rng_advance(int(delta) & 0xFFFFFFFF, 1 if set_value else 0, _stream()))


def ndhwc(x):
    """Return a tensor sharing x's logical NCDHW shape whose memory is dense NDHWC."""
    if x.dim() != 5:
        raise _lib.AdellHipError(f"expected a 5-D [N,C,D,H,W] tensor, got {tuple(x.shape)}")
    # (the common case returns x itself: two permutes + a contiguity check cost 3.3 us of host time
    # per call, ~200 calls per UNETR step)
    if x.is_contiguous(memory_format=torch.channels_last_3d):
        return x
    xp = x.permute(0, 2, 3, 4, 1)
    if not xp.is_contiguous():
        xp = xp.contiguous()
    return xp.permute(0, 4, 1, 2, 3)


def bn_running_update(mean, rstd, running_mean, running_var, num_batches_tracked, count, eps,
                      momentum):
    """torch.nn.BatchNorm running statistics in training, in place, from the batch (mean, rstd) of
    stats_finalize(per_item=False) over ``count`` elements per channel (csrc/norm_act.hip)."""
    _require_cuda(mean, rstd, running_mean, running_var)
    nbt = num_batches_tracked
    if nbt is not None and (not nbt.is_cuda or nbt.dtype != torch.int64):
        raise _lib.AdellHipError("num_batches_tracked must be an int64 device tensor")
    check(_lib.lib().adell_bn_running_update(
        _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var),
        None if nbt is None else ctypes.c_void_p(nbt.data_ptr()), int(mean.numel()), int(count),
        float(eps), -1.0 if momentum is None else float(momentum), _stream()))


def window_ndhwc(x, out_size, offset):
    """out[n, :, d, h, w] = x[n, :, d + od, h + oh, w + ow] where that voxel exists, zeros elsewhere
    (csrc/layout.hip): ``x`` a dense-NDHWC [N, C, D, H, W] tensor; returns a dense-NDHWC tensor of the
    spatial size ``out_size``. Crop (positive offsets) and its gradient (negative offsets)."""
    _require_cuda(x)
    N, C, D, H, W = x.shape
    Do, Ho, Wo = (int(v) for v in out_size)
    out = new_act(N, C, Do, Ho, Wo, x.device)
    check(_lib.lib().adell_window_ndhwc(_ptr(x), _ptr(out), N, C, D, H, W, Do, Ho, Wo,
                                        int(offset[0]), int(offset[1]), int(offset[2]), _stream()))
    return out


def new_act(N, C, D, H, W, device):
    return torch.empty((N, D, H, W, C), device=device, dtype=torch.float32).permute(0, 4, 1, 2, 3)


def conv_out_size(size, k, s, p):
    return tuple((d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip(size, k, s, p))


def make_conv_desc(N, size, C0, C1, Cout, k, s, p):
    k, s, p = _triple(k), _triple(s), _triple(p)
    o = conv_out_size(size, k, s, p)
    return ConvDesc(N, *size, C0, C1, Cout, *k, *s, *p, *o)


def pack_weight(w, mode):
    """mode 0/1: conv weight [Cout,Cin,k,k,k]; mode 2/3: convT weight [Cin,Cout,2,2,2]."""
    _require_cuda(w)
    w = w.contiguous()
    out = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
    d0, d1, kd, kh, kw = w.shape
    check(_lib.lib().adell_pack_weight(_ptr(w), _ptr(out), mode, d0, d1, kd, kh, kw, _stream()))
    return out


class SplitWeight:
    """Weights packed for the f16x3 kernel: fp16 hi/lo tiles + the scale-undo scalar."""

    __slots__ = ("halfs", "scale")

    def __init__(self, halfs, scale):
        self.halfs, self.scale = halfs, scale


def pack_weight_f16x3(w, mode):
    """mode 0: forward operand, mode 1: backward-data operand, of a conv weight
    [Cout, Cin, kD, kH, kW]."""
    _require_cuda(w)
    w = w.contiguous()
    d0, d1, kd, kh, kw = w.shape
    nbytes = _lib.lib().adell_pack_weight_f16x3_bytes(mode, d0, d1, kd * kh * kw)
    if nbytes < 0:
        check(int(nbytes))
    halfs = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    scale = torch.empty(d0 if mode == 0 else d1, device=w.device, dtype=torch.float32)
    check(_lib.lib().adell_pack_weight_f16x3(_ptr(w), _ptr(halfs), _ptr(scale), mode, d0, d1, kd,
                                             kh, kw, _stream()))
    return SplitWeight(halfs, scale)


def pack_weight_f16x3_multi(table, entries, total_blocks):
    """table: int64 [entries, 8] device tensor (see adell_pack_weight_f16x3_multi)."""
    if not (table.is_cuda and table.dtype == torch.int64 and table.is_contiguous()):
        raise AdellHipError("pack_weight_f16x3_multi: table must be a contiguous int64 CUDA tensor")
    check(_lib.lib().adell_pack_weight_f16x3_multi(_ptr(table), int(entries), int(total_blocks),
                                                   _stream()))


# conv launches that had to convert a split-row source back to fp32 (a plan that cannot stage rows)
ROWS_FALLBACKS = [0]


def _splitk_workspace(d, backward, device):
    """(workspace tensor or None, bytes) for the split-K path of the small (8^3 - 16^3) layers."""
    nbytes = _lib.lib().adell_conv3d_splitk_workspace(ctypes.byref(d), backward)
    if nbytes <= 0:
        return None, 0
    return _workspace(nbytes, device), nbytes


def conv3d_fwd(x0, w_packed, bias, Cout, kernel, stride, padding, x1=None, residual=None,
               want_stats=False, amax=None, rows0=None, rows1=None):
    """y = conv(cat(x0, x1)) + bias + residual ; optional (sum, sumsq) partials.
    ``w_packed``: fp32 GEMM-B tensor (exact fp32 MFMA) or a SplitWeight (f16x3 MFMA).
    ``rows0`` / ``rows1`` (SplitRows): that source holds split rows; a launch plan that cannot
    stage rows gets the fp32 tensor back first (rows_to_f32: one extra pass, counted in
    ROWS_FALLBACKS)."""
    split = isinstance(w_packed, SplitWeight)
    if rows0 is not None or rows1 is not None:
        N_, C0_ = x0.shape[:2]
        C1_ = 0 if x1 is None else x1.shape[1]
        if not (split and conv3d_rows_ok(N_, tuple(x0.shape[2:]), C0_, C1_, Cout, kernel, stride,
                                         padding)):
            ROWS_FALLBACKS[0] += 1
            if rows0 is not None:
                x0, rows0 = rows_to_f32(x0, rows0), None
            if rows1 is not None:
                x1, rows1 = rows_to_f32(x1, rows1), None
    if not split:
        _require_cuda(w_packed)
    _require_cuda(x0, x1, bias, residual)
    x0 = ndhwc(x0)
    N, C0, D, H, W = x0.shape
    C1 = 0
    if x1 is not None:
        x1 = ndhwc(x1)
        assert x1.shape[0] == N and tuple(x1.shape[2:]) == (D, H, W)
        C1 = x1.shape[1]
    d = make_conv_desc(N, (D, H, W), C0, C1, Cout, kernel, stride, padding)
    y = new_act(N, Cout, d.Do, d.Ho, d.Wo, x0.device)
    if residual is not None:
        residual = ndhwc(residual)
        assert tuple(residual.shape) == tuple(y.shape)
    part = None
    # the 32 -> 32 stride-2 downsampling layer: one persistent launch (csrc/conv_fwd_s2.hip)
    s2fused = (split and x1 is None and residual is None and not FLAGS["no_s2fused"]
               and bool(_lib.lib().adell_conv3d_fwd_s2_fused_applicable(ctypes.byref(d))))
    if rows0 is not None or rows1 is not None:
        s2fused = False
    if want_stats:
        fn = (_lib.lib().adell_conv3d_fwd_s2_fused_ntiles if s2fused
              else _lib.lib().adell_conv3d_fwd_ntiles_f16x3 if (rows0 is not None or rows1 is not None)
              else _lib.lib().adell_conv3d_fwd_ntiles_f16x3_ws if split
              else _lib.lib().adell_conv3d_fwd_ntiles)
        nt = fn(ctypes.byref(d))
        if nt < 0:
            check(nt)
        part = torch.empty((N, nt, Cout, 2), device=x0.device, dtype=torch.float32)
    rows = 0 if part is None else part.shape[1]    # checked by the library against its launch plan
    if rows0 is not None or rows1 is not None:
        check(_timed("adell_conv_igemm_f16_kernel", _conv_flops(d),
                     lambda: _lib.lib().adell_conv3d_fwd_f16x3_rows(
                         ctypes.byref(d), _ptr(x0), None if rows0 is None else _ptr(rows0.xk),
                         _ptr(x1), None if rows1 is None else _ptr(rows1.xk),
                         _ptr(w_packed.halfs), _ptr(w_packed.scale), _ptr(bias), _ptr(residual),
                         _ptr(y), _ptr(part), rows, _ptr(amax), _stream()), _conv_tag(d, "fwd"),
                     _conv_bytes(d, residual is not None)))
        return y, part
    if s2fused:
        check(_timed("adell_fwd_s2_fused_kernel", _conv_flops(d),
                     lambda: _lib.lib().adell_conv3d_fwd_s2_fused(
                         ctypes.byref(d), _ptr(x0), _ptr(w_packed.halfs), _ptr(w_packed.scale),
                         _ptr(bias), _ptr(y), _ptr(part), rows, _ptr(amax), _stream()),
                     _conv_tag(d, "fwd"), _conv_bytes(d)))
    elif split:
        ws, wsb = _splitk_workspace(d, 0, x0.device)
        check(_timed("adell_conv_igemm_f16_kernel", _conv_flops(d),
                     lambda: _lib.lib().adell_conv3d_fwd_f16x3_ws(
                         ctypes.byref(d), _ptr(x0), _ptr(x1), _ptr(w_packed.halfs),
                         _ptr(w_packed.scale), _ptr(bias), _ptr(residual), _ptr(y), _ptr(part),
                         rows, _ptr(amax), _ptr(ws), wsb, _stream()), _conv_tag(d, "fwd"),
                     _conv_bytes(d, residual is not None)))
    else:
        check(_timed("adell_conv_igemm_kernel", _conv_flops(d), lambda: _lib.lib().adell_conv3d_fwd(
            ctypes.byref(d), _ptr(x0), _ptr(x1), _ptr(w_packed), _ptr(bias), _ptr(residual),
            _ptr(y), _ptr(part), rows, _stream()), _conv_tag(d, "fwd"),
            _conv_bytes(d, residual is not None)))
    return y, part


# ---- convolutions with Cin <= 4 (the 2-channel input block): canonical weights, vector ALU ----
def conv_cin_small_ok(weight, x0, x1, stride, padding, residual):
    if residual is not None or x1 is not None or weight.dim() != 5 or x0.shape[1] > 4:
        return False
    if FLAGS["no_cin_small"]:
        return False
    k = tuple(weight.shape[2:])
    # measured at 128^3 (tools/bench_layers.py): 2 -> 2 0.21 ms here vs 0.34 ms on the MFMA path,
    # but 2 -> 32 is slower here (the vector ALU does 1728 FMAs per voxel): narrow outputs only
    if weight.shape[0] > 4 and not FLAGS["cin_small_all"]:
        return False
    return k[0] in (1, 3) and k[1:] == (3, 3) and tuple(stride) == (1, 1, 1)


def conv_cin_small_fwd(x, weight, bias, padding, want_stats):
    _require_cuda(x, weight, bias)
    x = ndhwc(x)
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[0]
    d = make_conv_desc(N, (D, H, W), Cin, 0, Cout, tuple(weight.shape[2:]), 1, padding)
    y = new_act(N, Cout, d.Do, d.Ho, d.Wo, x.device)
    part = None
    wc = weight.contiguous()   # bound to a local: must outlive the launch
    if want_stats:
        nt = _lib.lib().adell_conv_cin_small_ntiles(ctypes.byref(d))
        if nt < 0:
            check(nt)
        part = torch.empty((N, nt, Cout, 2), device=x.device, dtype=torch.float32)
    check(_timed("adell_cin_small_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv_cin_small_fwd(
                     ctypes.byref(d), _ptr(x), _ptr(wc), _ptr(bias), _ptr(y),
                     _ptr(part), 0 if part is None else part.shape[1], _stream()),
                 _conv_tag(d, "fwd"), _conv_bytes(d)))
    return y, part


def conv_cin_small_bwd_data(dy, weight, in_size, padding):
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    Cin = weight.shape[1]
    d = make_conv_desc(N, tuple(in_size), Cin, 0, Cout, tuple(weight.shape[2:]), 1, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    dx = new_act(N, Cin, *in_size, dy.device)
    wc = weight.contiguous()
    check(_timed("adell_cin_small_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv_cin_small_bwd_data(
                     ctypes.byref(d), _ptr(dy), _ptr(wc), _ptr(dx), _stream()),
                 _conv_tag(d, "dgrad"), _conv_bytes(d)))
    return dx


# ---- 3x3x3 stride-1 conv with 1..4 input channels and a wide output: K = 27 Cin GEMM per brick ---
def conv_cinfold_ok(weight, x0, x1, stride, padding, residual):
    """True when the im2col-GEMM kernels (csrc/conv_cinfold.hip) take this conv: they replace the
    x-tap fold + 16-channel MFMA chunk (forward) and the vector-ALU weight gradient."""
    if residual is not None or x1 is not None or weight.dim() != 5 or x0.dim() != 5:
        return False
    if FLAGS.get("no_cinfold") or x0.shape[1] > 4 or weight.shape[0] <= 4:
        return False
    return (tuple(weight.shape[2:]) == (3, 3, 3) and tuple(stride) == (1, 1, 1)
            and all(0 <= p <= 1 for p in padding))


def conv_cinfold_fwd(x, weight, bias, padding, want_stats, f16x3=False):
    """``f16x3``: the split-f16 MFMA kernel (two input channels) instead of the exact fp32 one."""
    _require_cuda(x, weight, bias)
    x = ndhwc(x)
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[0]
    d = make_conv_desc(N, (D, H, W), Cin, 0, Cout, 3, 1, padding)
    y = new_act(N, Cout, d.Do, d.Ho, d.Wo, x.device)
    part = None
    wc = weight.contiguous()   # bound to a local: must outlive the launch
    if want_stats:
        nt = _lib.lib().adell_conv_cinfold_ntiles(ctypes.byref(d))
        if nt < 0:
            check(nt)
        part = torch.empty((N, nt, Cout, 2), device=x.device, dtype=torch.float32)
    fn = _lib.lib().adell_conv_cinfold_fwd_f16x3 if f16x3 else _lib.lib().adell_conv_cinfold_fwd
    check(_timed("adell_cinfold_kernel", _conv_flops(d),
                 lambda: fn(ctypes.byref(d), _ptr(x), _ptr(wc), _ptr(bias), _ptr(y), _ptr(part),
                            0 if part is None else part.shape[1], _stream()),
                 _conv_tag(d, "fwd"), _conv_bytes(d)))
    return y, part


def conv_cinfold_bwd_weight(x, dy, padding, want_db, f16x3=False):
    """(dw [Cout, Cin, 3, 3, 3], db or None). ``f16x3``: the split-f16 MFMA kernel (bound by the one
    pass over dy) instead of the exact fp32-MFMA one."""
    _require_cuda(x, dy)
    x, dy = ndhwc(x), ndhwc(dy)
    N, Cin, D, H, W = x.shape
    Cout = dy.shape[1]
    d = make_conv_desc(N, (D, H, W), Cin, 0, Cout, 3, 1, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    nbytes = _lib.lib().adell_conv_cinfold_wgrad_workspace(ctypes.byref(d))
    check(min(nbytes, 0))
    ws = _workspace(nbytes, x.device)
    dw = torch.empty((Cout, Cin, 3, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty(Cout, device=x.device, dtype=torch.float32) if want_db else None
    fn = (_lib.lib().adell_conv_cinfold_bwd_weight_f16x3 if f16x3
          else _lib.lib().adell_conv_cinfold_bwd_weight)
    check(_timed("adell_cinfold_kernel", _conv_flops(d),
                 lambda: fn(ctypes.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws),
                            ws.numel() * 4, _stream()), _conv_tag(d, "wgrad"), _conv_bytes(d)))
    return dw, db


def conv_cinfold_bwd_data(dy, weight, in_size, padding, f16x3=False):
    """dx [N, Cin, *in_size] of a narrow-input conv, or None when the kernel does not take the
    shape (Cout > 64 or not a multiple of 4: the caller falls to the implicit-GEMM kernel)."""
    _require_cuda(dy, weight)
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    Cin = weight.shape[1]
    d = make_conv_desc(N, tuple(in_size), Cin, 0, Cout, 3, 1, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    if not _lib.lib().adell_conv_cinfold_dx_applicable(ctypes.byref(d)):
        return None
    dx = new_act(N, Cin, *in_size, dy.device)
    wc = weight.contiguous()
    fn = (_lib.lib().adell_conv_cinfold_bwd_data_f16x3 if f16x3
          else _lib.lib().adell_conv_cinfold_bwd_data)
    check(_timed("adell_cinfold_kernel", _conv_flops(d),
                 lambda: fn(ctypes.byref(d), _ptr(dy), _ptr(wc), _ptr(dx), _stream()),
                 _conv_tag(d, "dgrad"), _conv_bytes(d)))
    return dx


# ---- 1x1x1 convolution with Cout <= 4 (logits head): canonical weights, one pass each way ----
def conv1_small_ok(weight, Cin, stride, padding, residual):
    return (residual is None and weight.dim() == 5 and tuple(weight.shape[2:]) == (1, 1, 1)
            and weight.shape[0] <= 4 and Cin <= 512 and tuple(stride) == (1, 1, 1)
            and tuple(padding) == (0, 0, 0))


def _conv1_desc(N, size, C0, C1, Cout):
    return make_conv_desc(N, tuple(size), C0, C1, Cout, (1, 1, 1), (1, 1, 1), (0, 0, 0))


def conv1_small_fwd(x0, x1, weight, bias):
    _require_cuda(x0, x1, weight, bias)
    x0 = ndhwc(x0)
    N, C0, D, H, W = x0.shape
    C1 = 0
    if x1 is not None:
        x1 = ndhwc(x1)
        C1 = x1.shape[1]
    Cout = weight.shape[0]
    d = _conv1_desc(N, (D, H, W), C0, C1, Cout)
    y = new_act(N, Cout, D, H, W, x0.device)
    wc = weight.contiguous()
    check(_timed("adell_conv1_small_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv1_small_fwd(ctypes.byref(d), _ptr(x0), _ptr(x1),
                                                          _ptr(wc), _ptr(bias),
                                                          _ptr(y), _stream()),
                 _conv_tag(d, "fwd"), _conv_bytes(d)))
    return y


def conv1_small_bwd_data(dy, weight, in_size, C0, C1):
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    d = _conv1_desc(N, in_size, C0, C1, Cout)
    dx0 = new_act(N, C0, *in_size, dy.device)
    dx1 = new_act(N, C1, *in_size, dy.device) if C1 > 0 else None
    wc = weight.contiguous()
    check(_timed("adell_conv1_small_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv1_small_bwd_data(ctypes.byref(d), _ptr(dy),
                                                               _ptr(wc),
                                                               _ptr(dx0), _ptr(dx1), _stream()),
                 _conv_tag(d, "dgrad"), _conv_bytes(d)))
    return dx0, dx1


def conv1_small_bwd_weight(x0, x1, dy, want_db):
    x0, dy = ndhwc(x0), ndhwc(dy)
    N, C0, D, H, W = x0.shape
    C1 = 0
    if x1 is not None:
        x1 = ndhwc(x1)
        C1 = x1.shape[1]
    Cout = dy.shape[1]
    d = _conv1_desc(N, (D, H, W), C0, C1, Cout)
    ws = _workspace(_lib.lib().adell_conv1_small_wgrad_workspace(ctypes.byref(d)), x0.device)
    dw = torch.empty((Cout, C0 + C1, 1, 1, 1), device=x0.device, dtype=torch.float32)
    db = torch.empty((Cout,), device=x0.device, dtype=torch.float32) if want_db else None
    check(_timed("adell_conv1_small_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv1_small_bwd_weight(
                     ctypes.byref(d), _ptr(x0), _ptr(x1), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws),
                     ws.numel() * 4, _stream()), _conv_tag(d, "wgrad"), _conv_bytes(d)))
    return dw, db


def conv3d_bwd_data(dy, w_packed_bwd, in_size, C0, C1, kernel, stride, padding, amax=None,
                    add0=None):
    """``add0``: a tensor of dx0's shape added to it inside the kernel epilogue (f16x3 path, one
    destination); the caller adds it itself when this returns it unused (third value)."""
    split = isinstance(w_packed_bwd, SplitWeight)
    if add0 is not None and split and C1 == 0:
        _require_cuda(dy, add0)
        dy, add0 = ndhwc(dy), ndhwc(add0)
        N, Cout = dy.shape[:2]
        d = make_conv_desc(N, tuple(in_size), C0, 0, Cout, kernel, stride, padding)
        dx0 = new_act(N, C0, *in_size, dy.device)
        ws, wsb = _splitk_workspace(d, 1, dy.device)
        check(_timed("adell_conv_igemm_f16_kernel", _conv_flops(d),
                     lambda: _lib.lib().adell_conv3d_bwd_data_f16x3_add(
                         ctypes.byref(d), _ptr(dy), _ptr(w_packed_bwd.halfs),
                         _ptr(w_packed_bwd.scale), _ptr(add0), _ptr(dx0), _ptr(amax), _ptr(ws), wsb,
                         _stream()), _conv_tag(d, "dgrad"), _conv_bytes(d, True)))
        return dx0, None
    _require_cuda(dy)
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    d = make_conv_desc(N, tuple(in_size), C0, C1, Cout, kernel, stride, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    dx0 = new_act(N, C0, *in_size, dy.device)
    dx1 = new_act(N, C1, *in_size, dy.device) if C1 > 0 else None
    if split:
        ws, wsb = _splitk_workspace(d, 1, dy.device)
        check(_timed("adell_conv_igemm_f16_kernel", _conv_flops(d),
                     lambda: _lib.lib().adell_conv3d_bwd_data_f16x3_ws(
                         ctypes.byref(d), _ptr(dy), _ptr(w_packed_bwd.halfs),
                         _ptr(w_packed_bwd.scale), _ptr(dx0), _ptr(dx1), _ptr(amax), _ptr(ws), wsb,
                         _stream()), _conv_tag(d, "dgrad"), _conv_bytes(d)))
    else:
        check(_timed("adell_conv_igemm_kernel", _conv_flops(d),
                     lambda: _lib.lib().adell_conv3d_bwd_data(
                         ctypes.byref(d), _ptr(dy), _ptr(w_packed_bwd), _ptr(dx0), _ptr(dx1),
                         _stream()), _conv_tag(d, "dgrad"), _conv_bytes(d)))
    return dx0, dx1


def plan_epoch():
    """The library's launch-plan epoch (adell_plan_epoch): changes with every adell_set_tuning flip;
    row counts taken from the *_ntiles queries are valid within one epoch."""
    return int(_lib.lib().adell_plan_epoch())


def conv3d_bwd_data_adn_ntiles(in_size, N, C0, C1, Cout, kernel, stride, padding):
    """Bricks per batch item when the backward-data of this conv takes the fused ADN epilogue
    (adell_conv3d_bwd_data_f16x3_adn), else 0."""
    d = make_conv_desc(N, tuple(in_size), C0, C1, Cout, kernel, stride, padding)
    return int(_lib.lib().adell_conv3d_bwd_data_f16x3_adn_ntiles(ctypes.byref(d)))


def _adn_site_struct(site):
    if site is None:
        return None
    return _lib.AdnSite(_ptr(site.x), _ptr(site.mean), _ptr(site.rstd), _ptr(site.mask),
                        float(site.drop_p), float(site.act_p), ACT_IDS[site.act])


def conv3d_bwd_data_adn(dy, w_packed_bwd, in_size, C0, C1, kernel, stride, padding, ntiles,
                        site0=None, site1=None, amax=None, add0=None):
    """Backward-data whose destinations are gradients of the outputs of norm -> dropout ->
    activation sites (``site0`` for dx0, ``site1`` for dx1; objects with x, mean, rstd, mask,
    drop_p, act, act_p): returns (dt0 or dx0, dt1 or dx1, partials [N, ntiles, C0 + C1, 2])."""
    _require_cuda(dy, add0)
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    d = make_conv_desc(N, tuple(in_size), C0, C1, Cout, kernel, stride, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    if add0 is not None:
        add0 = ndhwc(add0)
    dx0 = new_act(N, C0, *in_size, dy.device)
    dx1 = new_act(N, C1, *in_size, dy.device) if C1 > 0 else None
    part = torch.empty((N, ntiles, C0 + C1, 2), device=dy.device, dtype=torch.float32)
    s0, s1 = _adn_site_struct(site0), _adn_site_struct(site1)
    extra = sum(4.0 * t.numel() for t, s in ((dx0, site0), (dx1, site1)) if s is not None)
    check(_timed("adell_conv_igemm_f16_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv3d_bwd_data_f16x3_adn(
                     ctypes.byref(d), _ptr(dy), _ptr(w_packed_bwd.halfs), _ptr(w_packed_bwd.scale),
                     _ptr(add0), _ptr(dx0), _ptr(dx1), _ptr(amax),
                     None if s0 is None else ctypes.byref(s0),
                     None if s1 is None else ctypes.byref(s1), _ptr(part), int(ntiles),
                     _stream()),
                 _conv_tag(d, "dgrad"), _conv_bytes(d, add0 is not None) + extra))
    return dx0, dx1, part


def conv3d_bwd_data_s2(dy, class_weights, in_size, C0, padding, amax=None, add0=None):
    """Backward-data of a stride-2 k = 3 conv by parity classes; ``class_weights``: 8 SplitWeights
    (c = 4 pz + 2 py + px) of the sub-kernels packed with mode 1. ``add0`` (dX-shaped) is added in
    the epilogue of the class launches."""
    _require_cuda(dy, add0)
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    d = make_conv_desc(N, tuple(in_size), C0, 0, Cout, 3, 2, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    dx = new_act(N, C0, *in_size, dy.device)
    PA = ctypes.c_void_p * 8
    wh = PA(*[_ptr(w.halfs) for w in class_weights])
    ws = PA(*[_ptr(w.scale) for w in class_weights])
    if add0 is not None:
        add0 = ndhwc(add0)
        assert add0.shape == dx.shape
        call = lambda: _lib.lib().adell_conv3d_bwd_data_s2_f16x3_add(
            ctypes.byref(d), _ptr(dy), wh, ws, _ptr(add0), _ptr(dx), _ptr(amax), _stream())
    else:
        call = lambda: _lib.lib().adell_conv3d_bwd_data_s2_f16x3(
            ctypes.byref(d), _ptr(dy), wh, ws, _ptr(dx), _ptr(amax), _stream())
    check(_timed("adell_conv_igemm_f16_kernel", _conv_flops(d), call,
                 _conv_tag(d, "dgrad"), _conv_bytes(d), kernels=8))
    return dx


def conv3d_bwd_data_s2_fused_ok(in_size, C0, C1, Cout, kernel, stride, padding):
    """The one-launch backward-data of the 32 -> 32 stride-2 k = 3 padding-1 layer applies."""
    if FLAGS["no_s2fused"] or C1 != 0:
        return False
    d = make_conv_desc(1, tuple(in_size), C0, 0, Cout, kernel, stride, padding)
    return bool(_lib.lib().adell_conv3d_bwd_data_s2_fused_applicable(ctypes.byref(d)))


def conv3d_bwd_data_s2_fused(dy, w, in_size, amax=None, add0=None):
    """dX of that layer (csrc/conv_dgrad_s2.hip); ``w``: SplitWeight of the full weight, mode 1."""
    _require_cuda(dy, add0)
    dy = ndhwc(dy)
    N, Cout = dy.shape[:2]
    d = make_conv_desc(N, tuple(in_size), 32, 0, Cout, 3, 2, 1)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    dx = new_act(N, 32, *in_size, dy.device)
    if add0 is not None:
        add0 = ndhwc(add0)
        assert add0.shape == dx.shape
    check(_timed("adell_dgrad_s2_fused_kernel", _conv_flops(d),
                 lambda: _lib.lib().adell_conv3d_bwd_data_s2_fused(
                     ctypes.byref(d), _ptr(dy), _ptr(w.halfs), _ptr(w.scale), _ptr(add0), _ptr(dx),
                     _ptr(amax), _stream()),
                 _conv_tag(d, "dgrad"), _conv_bytes(d)))
    return dx


def _workspace(nbytes, device):
    return torch.empty((max(int(nbytes), 4) + 3) // 4, device=device, dtype=torch.float32)


def conv3d_bwd_weight_rows_ok(N, in_size, C0, C1, Cout, kernel, stride, padding):
    """The weight gradient of this conv reads split-row sources (z-ring kernel)."""
    d = make_conv_desc(N, tuple(in_size), C0, C1, Cout, kernel, stride, padding)
    return bool(_lib.lib().adell_conv3d_bwd_weight_f16x3_rows_ok(ctypes.byref(d)))


def conv3d_bwd_weight(x0, dy, kernel, stride, padding, x1=None, want_db=False, f16x3=False,
                      x_amax=None, dy_amax=None, rows0=None, rows1=None):
    """dW in torch's canonical [Cout, Cin, kD, kH, kW] layout (and db when want_db).
    f16x3: error-compensated f16 MFMA instead of the fp32 MFMA. ``rows0`` / ``rows1``
    (SplitRows): that source holds split rows (converted back when the kernel that serves the
    problem cannot read them: ROWS_FALLBACKS)."""
    _require_cuda(x0, x1, dy)
    if rows0 is not None or rows1 is not None:
        ok = f16x3 and conv3d_bwd_weight_rows_ok(
            x0.shape[0], tuple(x0.shape[2:]), x0.shape[1], 0 if x1 is None else x1.shape[1],
            dy.shape[1], kernel, stride, padding)
        if not ok:
            ROWS_FALLBACKS[0] += 1
            if rows0 is not None:
                x0, rows0 = rows_to_f32(x0, rows0), None
            if rows1 is not None:
                x1, rows1 = rows_to_f32(x1, rows1), None
    x0, dy = ndhwc(x0), ndhwc(dy)
    N, C0, D, H, W = x0.shape
    C1 = 0
    if x1 is not None:
        x1 = ndhwc(x1)
        C1 = x1.shape[1]
    Cout = dy.shape[1]
    k = _triple(kernel)
    d = make_conv_desc(N, (D, H, W), C0, C1, Cout, kernel, stride, padding)
    assert (d.Do, d.Ho, d.Wo) == tuple(dy.shape[2:])
    L = _lib.lib()
    wsfn, fn, name = ((L.adell_conv3d_bwd_weight_f16x3_workspace, L.adell_conv3d_bwd_weight_f16x3,
                       "adell_conv_wgrad_f16_kernel") if f16x3 else
                      (L.adell_conv3d_bwd_weight_workspace, L.adell_conv3d_bwd_weight,
                       "adell_conv_wgrad_kernel"))
    nbytes = wsfn(ctypes.byref(d))
    if nbytes < 0:
        check(int(nbytes))
    ws = _workspace(nbytes, x0.device)
    dw = torch.empty((Cout, C0 + C1, *k), device=x0.device, dtype=torch.float32)
    db = torch.empty((Cout,), device=x0.device, dtype=torch.float32) if want_db else None
    if rows0 is not None or rows1 is not None:
        # (x_amax: the forward's by-product covers the fp32 source only, as this call needs it)
        check(_timed(name, _conv_flops(d), lambda: L.adell_conv3d_bwd_weight_f16x3_rows(
            ctypes.byref(d), _ptr(x0), None if rows0 is None else _ptr(rows0.xk), _ptr(x1),
            None if rows1 is None else _ptr(rows1.xk), _ptr(dy), _ptr(dw), _ptr(db), _ptr(x_amax),
            _ptr(dy_amax), _ptr(ws), ws.numel() * 4, _stream()), _conv_tag(d, "wgrad"),
                     _conv_bytes(d)))
    elif f16x3:
        check(_timed(name, _conv_flops(d), lambda: fn(
            ctypes.byref(d), _ptr(x0), _ptr(x1), _ptr(dy), _ptr(dw), _ptr(db), _ptr(x_amax),
            _ptr(dy_amax), _ptr(ws), ws.numel() * 4, _stream()), _conv_tag(d, "wgrad"),
                     _conv_bytes(d)))
    else:
        check(_timed(name, _conv_flops(d), lambda: fn(
            ctypes.byref(d), _ptr(x0), _ptr(x1), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws),
            ws.numel() * 4, _stream()), _conv_tag(d, "wgrad"), _conv_bytes(d)))
    return (dw, db) if want_db else dw


def bias_grad(dy):
    _require_cuda(dy)
    dy = ndhwc(dy)
    C = dy.shape[1]
    rows = dy.numel() // C
    nbytes = _lib.lib().adell_bias_grad_workspace(rows, C)
    ws = _workspace(nbytes, dy.device)
    db = torch.empty((C,), device=dy.device, dtype=torch.float32)
    check(_lib.lib().adell_bias_grad(_ptr(dy), rows, C, _ptr(db), _ptr(ws), ws.numel() * 4,
                                     _stream()))
    return db


def convtranspose3d_bwd_weight(x, dy, factors=(2, 2, 2)):
    """dW in torch's canonical [Cin, Cout, FD, FH, FW] layout (kernel = stride = factors)."""
    _require_cuda(x, dy)
    x, dy = ndhwc(x), ndhwc(dy)
    N, Cin, D, H, W = x.shape
    Cout = dy.shape[1]
    fd, fh, fw = factors
    nbytes = _lib.lib().adell_convtranspose3d_bwd_weight_workspace(N, D, H, W, Cin, Cout, fd, fh, fw)
    if nbytes < 0:
        check(int(nbytes))
    ws = _workspace(nbytes, x.device)
    dw = torch.empty((Cin, Cout, fd, fh, fw), device=x.device, dtype=torch.float32)
    flops = 2.0 * N * D * H * W * Cin * Cout * fd * fh * fw
    check(_timed("adell_conv_wgrad_kernel", flops, lambda: _lib.lib().adell_convtranspose3d_bwd_weight(
        N, D, H, W, Cin, Cout, fd, fh, fw, _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel() * 4,
        _stream())))
    return dw


# ---- factor-2 transposed conv with 32 / 64 channels: streaming GEMMs, canonical weights ----------
def _convt_k2_fns(factors):
    L = _lib.lib()
    if tuple(factors) == (2, 2, 2):
        return (L.adell_convt_k2_applicable, L.adell_convt_k2_fwd, L.adell_convt_k2_bwd_data,
                L.adell_convt_k2_wgrad_workspace, L.adell_convt_k2_bwd_weight)
    if tuple(factors) == (2, 2, 1):
        return (L.adell_convt_k221_applicable, L.adell_convt_k221_fwd, L.adell_convt_k221_bwd_data,
                L.adell_convt_k221_wgrad_workspace, L.adell_convt_k221_bwd_weight)
    return None


def convt_k2_ok(x_shape, weight):
    """True when csrc/convt_k2.hip takes ConvTranspose3d(x) with this weight [Cin, Cout, 2, 2, 2]
    (or [Cin, Cout, 2, 2, 1]: depth and height doubled, width kept)."""
    if FLAGS.get("no_convt_k2") or weight.dim() != 5:
        return False
    fns = _convt_k2_fns(weight.shape[2:])
    if fns is None:
        return False
    N, _, D, H, W = x_shape
    return bool(fns[0](N, D, H, W, weight.shape[0], weight.shape[1]))


def convt_k2_fwd(x, weight, bias):
    _require_cuda(x, weight, bias)
    x = ndhwc(x)
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[1]
    f = tuple(weight.shape[2:])
    y = new_act(N, Cout, f[0] * D, f[1] * H, f[2] * W, x.device)
    wc = weight.contiguous()   # bound to a local: must outlive the launch
    fn = _convt_k2_fns(f)[1]
    nf = f[0] * f[1] * f[2]
    check(_timed("adell_convt_k2_kernel", 2.0 * nf * N * D * H * W * Cin * Cout,
                 lambda: fn(N, D, H, W, Cin, Cout, _ptr(x), _ptr(wc), _ptr(bias), _ptr(y), _stream()),
                 f"convT fwd {Cin}->{Cout} in {D}x{H}x{W} f{f[0]}{f[1]}{f[2]}",
                 4.0 * (x.numel() + y.numel())))
    return y


def convt_k2_bwd_data(dy, weight):
    _require_cuda(dy, weight)
    dy = ndhwc(dy)
    N, Cout, D2, H2, W2 = dy.shape
    Cin = weight.shape[0]
    f = tuple(weight.shape[2:])
    D, H, W = D2 // f[0], H2 // f[1], W2 // f[2]
    dx = new_act(N, Cin, D, H, W, dy.device)
    wc = weight.contiguous()
    fn = _convt_k2_fns(f)[2]
    nf = f[0] * f[1] * f[2]
    check(_timed("adell_convt_k2_kernel", 2.0 * nf * N * D * H * W * Cin * Cout,
                 lambda: fn(N, D, H, W, Cin, Cout, _ptr(dy), _ptr(wc), _ptr(dx), _stream()),
                 f"convT dgrad {Cin}->{Cout} in {D}x{H}x{W} f{f[0]}{f[1]}{f[2]}",
                 4.0 * (dy.numel() + dx.numel())))
    return dx


def convt_k2_bwd_weight(x, dy, want_db=False, factors=(2, 2, 2)):
    """dW (and, with want_db, the bias gradient from the same pass over dy: returns (dw, db))."""
    _require_cuda(x, dy)
    x, dy = ndhwc(x), ndhwc(dy)
    N, Cin, D, H, W = x.shape
    Cout = dy.shape[1]
    f = tuple(factors)
    fns = _convt_k2_fns(f)
    nbytes = fns[3](N, D, H, W, Cin, Cout)
    check(min(nbytes, 0))
    ws = _workspace(nbytes, x.device)
    dw = torch.empty((Cin, Cout, *f), device=x.device, dtype=torch.float32)
    db = torch.empty(Cout, device=x.device, dtype=torch.float32) if want_db else None
    nf = f[0] * f[1] * f[2]
    check(_timed("adell_convt_k2_kernel", 2.0 * nf * N * D * H * W * Cin * Cout,
                 lambda: fns[4](N, D, H, W, Cin, Cout, _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws),
                                ws.numel() * 4, _stream()),
                 f"convT wgrad {Cin}->{Cout} in {D}x{H}x{W} f{f[0]}{f[1]}{f[2]}",
                 4.0 * (x.numel() + dy.numel())))
    return (dw, db) if want_db else dw


def convtranspose3d_k2s2_bwd_weight(x, dy):
    return convtranspose3d_bwd_weight(x, dy, (2, 2, 2))


def convtranspose3d_fwd(x, w_packed, bias, Cout, factors=(2, 2, 2)):
    """ConvTranspose3d with kernel = stride = factors (each 1 or 2), padding 0. ``w_packed``:
    fp32 pack (mode 2) or the SplitWeight of the virtual 1x1x1 weight (f16x3)."""
    _require_cuda(x, bias)
    x = ndhwc(x)
    N, Cin, D, H, W = x.shape
    fd, fh, fw = factors
    y = new_act(N, Cout, fd * D, fh * H, fw * W, x.device)
    flops = 2.0 * N * D * H * W * Cin * Cout * fd * fh * fw
    if isinstance(w_packed, SplitWeight):
        check(_timed("adell_conv_igemm_f16_kernel", flops,
                     lambda: _lib.lib().adell_convtranspose3d_fwd_f16x3(
                         N, D, H, W, Cin, Cout, fd, fh, fw, _ptr(x), _ptr(w_packed.halfs),
                         _ptr(w_packed.scale), _ptr(bias), _ptr(y), None, _stream()),
                     f"convT fwd {Cin}->{Cout} in {D}x{H}x{W} f{fd}{fh}{fw}",
                     4.0 * (x.numel() + y.numel())))
        return y
    _require_cuda(w_packed)
    check(_timed("adell_conv_igemm_kernel", flops, lambda: _lib.lib().adell_convtranspose3d_fwd(
        N, D, H, W, Cin, Cout, fd, fh, fw, _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(y),
        _stream())))
    return y


def convtranspose3d_bwd_data(dy, w_packed_bwd, Cin, factors=(2, 2, 2)):
    _require_cuda(dy)
    dy = ndhwc(dy)
    N, Cout, D2, H2, W2 = dy.shape
    fd, fh, fw = factors
    D, H, W = D2 // fd, H2 // fh, W2 // fw
    dx = new_act(N, Cin, D, H, W, dy.device)
    flops = 2.0 * N * D * H * W * Cin * Cout * fd * fh * fw
    if isinstance(w_packed_bwd, SplitWeight):
        check(_timed("adell_conv_igemm_f16_kernel", flops,
                     lambda: _lib.lib().adell_convtranspose3d_bwd_data_f16x3(
                         N, D, H, W, Cin, Cout, fd, fh, fw, _ptr(dy), _ptr(w_packed_bwd.halfs),
                         _ptr(w_packed_bwd.scale), _ptr(dx), None, _stream()),
                     f"convT dgrad {Cin}->{Cout} in {D}x{H}x{W} f{fd}{fh}{fw}",
                     4.0 * (dy.numel() + dx.numel())))
        return dx
    _require_cuda(w_packed_bwd)
    check(_timed("adell_conv_igemm_kernel", flops, lambda: _lib.lib().adell_convtranspose3d_bwd_data(
        N, D, H, W, Cin, Cout, fd, fh, fw, _ptr(dy), _ptr(w_packed_bwd), _ptr(dx), _stream())))
    return dx


def convtranspose3d_k2s2_fwd(x, w_packed, bias, Cout):
    return convtranspose3d_fwd(x, w_packed, bias, Cout, (2, 2, 2))


def convtranspose3d_k2s2_bwd_data(dy, w_packed_bwd, Cin):
    return convtranspose3d_bwd_data(dy, w_packed_bwd, Cin, (2, 2, 2))


def stats_finalize(partials, count, eps, per_item=True):
    """(mean, rstd): [N, C] per item (instance norm) or [C] over the batch (batch norm)."""
    N, nt, C, _ = partials.shape
    mean = torch.empty((N, C) if per_item else (C,), device=partials.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    nbytes = _lib.lib().adell_stats_finalize_workspace(N, nt, C)
    ws = _workspace(nbytes, partials.device) if nbytes > 0 else None
    check(_lib.lib().adell_stats_finalize(_ptr(partials), N, nt, C, int(count), float(eps),
                                          1 if per_item else 0, _ptr(mean), _ptr(rstd), _ptr(ws),
                                          0 if ws is None else ws.numel() * 4, _stream()))
    return mean, rstd


def channel_partials(x):
    _require_cuda(x)
    x = ndhwc(x)
    N, C = x.shape[:2]
    V = x.shape[2] * x.shape[3] * x.shape[4]
    nt = _lib.lib().adell_channel_partials_ntiles(V)
    part = torch.empty((N, nt, C, 2), device=x.device, dtype=torch.float32)
    check(_lib.lib().adell_channel_partials(_ptr(x), N, V, C, _ptr(part), _stream()))
    return part


def instance_stats(x, eps=1e-5, partials=None):
    """(mean, rstd) of shape [N, C] over the spatial dims of x."""
    if partials is None:
        partials = channel_partials(x)
    V = x.shape[2] * x.shape[3] * x.shape[4]
    return stats_finalize(partials, V, eps)


def make_na_desc(x, act, stats_per_item=1, act_p=0.0, act_w_n=0, drop_p=0.0, seed=0,
                 rng_offset=0):
    N, C = x.shape[:2]
    V = x.shape[2] * x.shape[3] * x.shape[4]
    act_id = ACT_IDS[act] if isinstance(act, str) else int(act)
    return NormActDesc(N, V, C, stats_per_item, act_id, act_w_n, float(act_p), float(drop_p),
                       int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_offset) & 0xFFFFFFFF)


def norm_act_mask_ok(x):
    """The forward can store its dropout keep bits (the bandwidth-tuned kernel runs: power-of-two
    channel count, 16-byte aligned NDHWC tensor)."""
    C = x.shape[1]
    return x.dim() == 5 and C % 4 == 0 and C <= 1024 and (C & (C - 1)) == 0


# ---- split rows: activations stored as the f16x3 kernels' LDS row image ------------------------------
class SplitRows:
    """Marks a tensor whose MEMORY holds, per voxel and 16-channel chunk, the 64-byte row
    [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] of fp16 values of x * 2^exp (conv_igemm_f16.h) instead
    of fp32 values -- same logical shape [N, C, D, H, W], dtype and byte count. ``xk``: int32 device
    tensor [N, C / 16] of exponents (one value per tensor for the producers built so far: ``exp``)."""

    __slots__ = ("exp", "xk")

    def __init__(self, exp, xk):
        self.exp, self.xk = int(exp), xk


_XK_CACHE = {}


def split_exponents(N, C, exp, device):
    """Constant exponent table [N, C / 16] (cached: a handful of distinct (N, C, exp) per model)."""
    key = (device, int(N) * (int(C) // 16), int(exp))
    t = _XK_CACHE.get(key)
    if t is None:
        t = _XK_CACHE[key] = torch.full((key[1],), int(exp), device=device, dtype=torch.int32)
    return t


def rows_from_f32(x, exp):
    """fp32 NDHWC activation -> (tensor of split rows, SplitRows); |x| * 2^exp must stay below 2^15."""
    _require_cuda(x)
    x = ndhwc(x)
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    xk = split_exponents(N, C, exp, x.device)
    out = new_act(*x.shape, x.device)
    check(_lib.lib().adell_split_rows_from_f32(_ptr(x), N, V, C, _ptr(xk), _ptr(out), _stream()))
    return out, SplitRows(exp, xk)


def rows_to_f32(rows, sr):
    """The fp32 tensor a split-row tensor stands for ((hi + lo) * 2^-exp: 22 mantissa bits)."""
    _require_cuda(rows)
    rows = ndhwc(rows)
    N, C = rows.shape[:2]
    V = rows.numel() // (N * C)
    out = new_act(*rows.shape, rows.device)
    check(_lib.lib().adell_split_rows_to_f32(_ptr(rows), N, V, C, _ptr(sr.xk), _ptr(out), _stream()))
    return out


def conv3d_rows_ok(N, in_size, C0, C1, Cout, kernel, stride, padding):
    """The forward of this conv stages split-row sources (specialised f16x3 instances)."""
    k, st = _triple(kernel), _triple(stride)
    if k != (3, 3, 3) or st != (1, 1, 1) or C0 % 16 or C1 % 16:
        return False
    d = make_conv_desc(N, tuple(in_size), C0, C1, Cout, kernel, stride, padding)
    return bool(_lib.lib().adell_conv3d_f16x3_rows_ok(ctypes.byref(d)))


def norm_act_fwd(x, mean, rstd, act, gamma=None, beta=None, act_w=None, act_p=0.0,
                 stats_per_item=1, drop_p=0.0, seed=0, rng_offset=0, want_mask=False,
                 split_exp=None):
    """``want_mask`` (drop_p > 0): also returns the keep bits as an int64 tensor (1 bit per
    element) for the fused backward (conv3d_bwd_data_adn). ``split_exp``: the output is written as
    SPLIT ROWS scaled by 2^split_exp (SplitRows below) instead of fp32 values -- same shape, dtype
    and bytes, readable only by consumers that take rows (conv3d_fwd / conv3d_bwd_weight)."""
    _require_cuda(x, mean, rstd, gamma, beta, act_w)
    x = ndhwc(x)
    d = make_na_desc(x, act, stats_per_item, act_p, 0 if act_w is None else act_w.numel(),
                     drop_p, seed, rng_offset)
    out = new_act(*x.shape, x.device)
    mask = None
    if want_mask and drop_p > 0.0:
        nbytes = _lib.lib().adell_norm_act_mask_bytes(ctypes.byref(d))
        if nbytes < 0:
            check(int(nbytes))
        mask = torch.empty((nbytes // 8,), device=x.device, dtype=torch.int64)
    # HBM-bound family of the roofline report: algorithmic bytes = read x + write out
    if split_exp is not None:
        check(_timed(NORM_ACT_FAMILY, 0.0, lambda: _lib.lib().adell_norm_act_fwd_split(
            ctypes.byref(d), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(act_w),
            _ptr(out), int(split_exp), _ptr(mask), _stream()), "fwd", 8.0 * x.numel()))
    elif mask is not None:
        check(_timed(NORM_ACT_FAMILY, 0.0, lambda: _lib.lib().adell_norm_act_fwd_mask(
            ctypes.byref(d), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(act_w),
            _ptr(out), _ptr(mask), _stream()), "fwd", 8.0 * x.numel()))
    else:
        check(_timed(NORM_ACT_FAMILY, 0.0, lambda: _lib.lib().adell_norm_act_fwd(
            ctypes.byref(d), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(act_w),
            _ptr(out), _stream()), "fwd", 8.0 * x.numel()))
    return (out, mask) if want_mask else out


def norm_act_bwd_from_dt(x, dt, mean, rstd, partials, poff=0):
    """dx of an instance-norm site from dt = dout * act'(u) * keep / (1 - p) and the per-brick
    sums the fused backward-data epilogue left in ``partials`` [N, ntiles, pstride, 2] (the site's
    channels at columns [poff, poff + C)). Written in place over ``dt``."""
    _require_cuda(x, dt, mean, rstd, partials)
    x, dt = ndhwc(x), ndhwc(dt)
    d = make_na_desc(x, "identity", 1)
    N, C = x.shape[:2]
    ws = _workspace(4 * (2 + 2 * ((int(partials.shape[1]) + 255) // 256)) * N * C, x.device)
    check(_timed(NORM_ACT_FAMILY, 0.0, lambda: _lib.lib().adell_norm_act_bwd_from_dt(
        ctypes.byref(d), _ptr(x), _ptr(dt), _ptr(mean), _ptr(rstd), _ptr(partials),
        int(partials.shape[1]), int(partials.shape[2]), int(poff), _ptr(dt), _ptr(ws),
        ws.numel() * 4, _stream()), "bwd", 12.0 * x.numel()))
    return dt


def prelu_wgrad(x, dout, mean, rstd, act_w, gamma=None, beta=None, stats_per_item=1, drop_p=0.0,
                seed=0, rng_offset=0):
    """Gradient of the PReLU weight(s) of norm -> dropout -> PReLU (same operands as
    norm_act_bwd)."""
    _require_cuda(x, dout, mean, rstd, gamma, beta, act_w)
    x, dout = ndhwc(x), ndhwc(dout)
    d = make_na_desc(x, "prelu", stats_per_item, 0.0, act_w.numel(), drop_p, seed, rng_offset)
    ws = _workspace(_lib.lib().adell_prelu_wgrad_workspace(ctypes.byref(d)), x.device)
    dw = torch.empty_like(act_w)
    check(_lib.lib().adell_prelu_wgrad(ctypes.byref(d), _ptr(x), _ptr(dout), _ptr(mean), _ptr(rstd),
                                       _ptr(gamma), _ptr(beta), _ptr(dw), _ptr(ws),
                                       ws.numel() * 4, _stream()))
    return dw


def norm_act_bwd(x, dout, mean, rstd, act, gamma=None, beta=None, act_w=None, act_p=0.0,
                 stats_per_item=1, drop_p=0.0, seed=0, rng_offset=0, want_affine_grads=False):
    """dx (and dgamma, dbeta when want_affine_grads) of norm_act_fwd."""
    _require_cuda(x, dout, mean, rstd, gamma, beta, act_w)
    x, dout = ndhwc(x), ndhwc(dout)
    d = make_na_desc(x, act, stats_per_item, act_p, 0 if act_w is None else act_w.numel(),
                     drop_p, seed, rng_offset)
    dx = new_act(*x.shape, x.device)
    dgamma = dbeta = None
    if want_affine_grads:
        dgamma = torch.empty((x.shape[1],), device=x.device, dtype=torch.float32)
        dbeta = torch.empty_like(dgamma)
    ws = None
    if mean is not None or want_affine_grads:
        ws = _workspace(_lib.lib().adell_norm_act_bwd_workspace(ctypes.byref(d)), x.device)
    # algorithmic bytes: read x and dout once, write dx once
    check(_timed(NORM_ACT_FAMILY, 0.0, lambda: _lib.lib().adell_norm_act_bwd(
        ctypes.byref(d), _ptr(x), _ptr(dout), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
        _ptr(act_w), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
        0 if ws is None else ws.numel() * 4, _stream()), "bwd", 12.0 * x.numel()))
    return dx, dgamma, dbeta


def norm_act_lowrank_ok(x, co):
    """Whether norm_act_bwd_lowrank takes this site: power-of-two channels, <= 4 factors."""
    C = x.shape[1]
    return 1 <= co <= 4 and 4 <= C <= 1024 and (C & (C - 1)) == 0


def norm_act_bwd_lowrank(x, g, w, mean, rstd, act, act_p=0.0, drop_p=0.0, seed=0, rng_offset=0):
    """dx of norm_act_fwd when the upstream gradient is g (x) w: g [N, co, ...] the gradient of a
    1x1x1 conv's output, w [co, C] its weight -- that conv's backward-data result is never formed
    (adell_norm_act_bwd_lowrank)."""
    _require_cuda(x, g, w, mean, rstd)
    x, g = ndhwc(x), ndhwc(g)
    co = g.shape[1]
    w = w.reshape(co, x.shape[1]).contiguous()
    assert g.shape[0] == x.shape[0] and g.shape[2:] == x.shape[2:]
    d = make_na_desc(x, act, 1, act_p, 0, drop_p, seed, rng_offset)
    dx = new_act(*x.shape, x.device)
    ws = _workspace(_lib.lib().adell_norm_act_bwd_workspace(ctypes.byref(d)), x.device)
    # algorithmic bytes: read x and g once, write dx once
    nb = 8.0 * x.numel() + 4.0 * g.numel()
    check(_timed(NORM_ACT_FAMILY, 0.0, lambda: _lib.lib().adell_norm_act_bwd_lowrank(
        ctypes.byref(d), _ptr(x), _ptr(g), _ptr(w), co, _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(ws),
        ws.numel() * 4, _stream()), "bwd", nb))
    return dx


def dice_focal_fwd(prob, target, smooth, dice_eps, gamma, focal_eps, focal_alpha=1.0):
    """Per-item (dice[B], focal[B]) and the sums the backward needs. prob/target: [B, ...]."""
    _require_cuda(prob, target)
    prob, target = prob.contiguous(), target.contiguous()
    B = prob.shape[0]
    S = prob.numel() // B
    assert target.numel() == prob.numel()
    ws = _workspace(_lib.lib().adell_dice_focal_workspace(B, S), prob.device)
    dice = torch.empty((B,), device=prob.device, dtype=torch.float32)
    focal = torch.empty_like(dice)
    sums = torch.empty((B, 3), device=prob.device, dtype=torch.float32)
    check(_lib.lib().adell_dice_focal_fwd(_ptr(prob), _ptr(target), B, S, smooth, dice_eps, gamma,
                                          float(focal_alpha), focal_eps, _ptr(dice), _ptr(focal),
                                          _ptr(sums),
                                          _ptr(ws), ws.numel() * 4, _stream()))
    return dice, focal, sums


def dice_focal_bwd(prob, target, sums, smooth, dice_eps, gamma, focal_eps, gdice, gfocal,
                   focal_alpha=1.0):
    prob, target = prob.contiguous(), target.contiguous()
    B = prob.shape[0]
    S = prob.numel() // B
    dprob = torch.empty_like(prob)
    check(_lib.lib().adell_dice_focal_bwd(_ptr(prob), _ptr(target), B, S, smooth, dice_eps, gamma,
                                          float(focal_alpha), focal_eps, _ptr(sums), float(gdice),
                                          float(gfocal),
                                          _ptr(dprob), _stream()))
    return dprob


def dice_focal_bwd_dev(prob, target, sums, smooth, dice_eps, gamma, focal_eps, gdice, gfocal,
                       focal_alpha=1.0):
    """gdice / gfocal: per-item upstream gradients as CUDA tensors [B] (or None)."""
    prob, target = prob.contiguous(), target.contiguous()
    B = prob.shape[0]
    S = prob.numel() // B
    dprob = torch.empty_like(prob)
    gd = None if gdice is None else gdice.contiguous().float()
    gf = None if gfocal is None else gfocal.contiguous().float()
    check(_lib.lib().adell_dice_focal_bwd_dev(_ptr(prob), _ptr(target), B, S, smooth, dice_eps,
                                              gamma, float(focal_alpha), focal_eps, _ptr(sums),
                                              _ptr(gd), _ptr(gf), _ptr(dprob), _stream()))
    return dprob


def class_sums_fwd(p, t):
    """p, t: [B, V, C] contiguous -> sums [B, C, 3] = (sum p t, sum p, sum t) over the voxels."""
    _require_cuda(p, t)
    B, V, C = p.shape
    nbytes = _lib.lib().adell_class_sums_workspace(B, V, C)
    ws = _workspace(nbytes, p.device)
    sums = torch.empty((B, C, 3), device=p.device, dtype=torch.float32)
    check(_lib.lib().adell_class_sums_fwd(_ptr(p), _ptr(t), B, V, C, _ptr(sums), _ptr(ws),
                                          ws.numel() * 4, _stream()))
    return sums


def class_sums_bwd(t, gsums):
    """dp [B, V, C] = gsums[..., 0] * t + gsums[..., 1]."""
    _require_cuda(t, gsums)
    B, V, C = t.shape
    dp = torch.empty_like(t)
    g = gsums.contiguous()
    check(_lib.lib().adell_class_sums_bwd(_ptr(t), _ptr(g), B, V, C, _ptr(dp), _stream()))
    return dp


# Bumped whenever a HIP kernel rewrites parameters in place (torch's version
# counters do not see those writes); functional._packed keys its cache on it.
WEIGHT_EPOCH = 0


def _weights_changed():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def sgd_step(param, grad, buf, lr, momentum, weight_decay, nesterov, first, grad_scale=1.0):
    _require_cuda(param, grad, buf)
    check(_lib.lib().adell_sgd_step(_ptr(param), _ptr(grad), _ptr(buf), param.numel(), lr,
                                    momentum, weight_decay, int(nesterov), int(first),
                                    grad_scale, _stream()))
    _weights_changed()


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step,
               grad_scale=1.0, decoupled=True):
    """torch.optim.AdamW (decoupled decay) or torch.optim.Adam (decay added to the gradient)."""
    _require_cuda(param, grad, exp_avg, exp_avg_sq)
    fn = _lib.lib().adell_adamw_step if decoupled else _lib.lib().adell_adam_step
    check(fn(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr, beta1,
             beta2, eps, weight_decay, int(step), grad_scale, _stream()))
    _weights_changed()


def ema_update(shadow, param, decay):
    _require_cuda(shadow, param)
    check(_lib.lib().adell_ema_update(_ptr(shadow), _ptr(param), shadow.numel(), decay, _stream()))
    _weights_changed()


# ---- token-sequence ops (ViT encoder of UNETR) ------------------------------------------
def layernorm_fwd(x, gamma, beta, eps):
    """LayerNorm over the last dim of a contiguous tensor; returns (y, mean, rstd)."""
    _require_cuda(x, gamma, beta)
    x = x.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty((rows,), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    check(_lib.lib().adell_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                                         _ptr(rstd), rows, C, float(eps), _stream()))
    return y, mean, rstd


def layernorm_bwd(x, dy, gamma, mean, rstd, want_affine):
    x, dy = x.contiguous(), dy.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    dg = db = ws = None
    if want_affine:
        dg = torch.empty((C,), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        ws = _workspace(_lib.lib().adell_layernorm_bwd_workspace(rows, C), x.device)
    check(_lib.lib().adell_layernorm_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(mean), _ptr(rstd),
                                         _ptr(dx), _ptr(dg), _ptr(db), rows, C, _ptr(ws),
                                         0 if ws is None else ws.numel() * 4, _stream()))
    return dx, dg, db


def add_bcast(a, b):
    """a + b where b is broadcast over a's leading dims (b.numel() divides a.numel())."""
    _require_cuda(a, b)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(_lib.lib().adell_add_bcast(_ptr(a), _ptr(b), _ptr(out), a.numel(), b.numel(), _stream()))
    return out


def sum_bcast(g, period_shape):
    _require_cuda(g)
    g = g.contiguous()
    db = torch.empty(period_shape, device=g.device, dtype=torch.float32)
    period = db.numel()
    rows = g.numel() // period
    if rows >= 64 and period < 2 ** 31:
        # column sums of the [rows][period] view: chunked partials + fixed-order fold
        nbytes = _lib.lib().adell_bias_grad_workspace(rows, period)
        ws = _workspace(nbytes, g.device)
        check(_lib.lib().adell_bias_grad(_ptr(g), rows, period, _ptr(db), _ptr(ws),
                                         ws.numel() * 4, _stream()))
        return db
    check(_lib.lib().adell_sum_bcast(_ptr(g), _ptr(db), g.numel(), period, _stream()))
    return db


def attention_fwd(q, k, v, bias, scale, drop_p=0.0, seed=0, offset=0):
    """q,k: [BH,T,A]; v: [BH,T,Dv]; bias: [nbias,T,T] or None. Returns (out, lse)."""
    _require_cuda(q, k, v, bias)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    BH, T, A = q.shape
    Dv = v.shape[-1]
    out = torch.empty((BH, T, Dv), device=q.device, dtype=torch.float32)
    lse = torch.empty((BH, T), device=q.device, dtype=torch.float32)
    nb = 0
    if bias is not None:
        bias = bias.contiguous()
        nb = bias.numel() // (T * T)
    check(_lib.lib().adell_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(bias), nb, BH, T, A, Dv,
                                         float(scale), float(drop_p),
                                         int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF,
                                         _ptr(out), _ptr(lse), _stream()))
    return out, lse


def attention_bwd(q, k, v, bias, out, dout, lse, scale, drop_p=0.0, seed=0, offset=0):
    BH, T, A = q.shape
    Dv = v.shape[-1]
    dout = dout.contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    if bias is not None:
        bias = bias.contiguous()
    nb = 0 if bias is None else bias.numel() // (T * T)
    check(_lib.lib().adell_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(bias), nb, _ptr(out),
                                         _ptr(dout), _ptr(lse), BH, T, A, Dv, float(scale),
                                         float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                         int(offset) & 0xFFFFFFFF,
                                         _ptr(dq), _ptr(dk), _ptr(dv), _stream()))
    return dq, dk, dv


# ---- data movement (U-Net++ dense links) ----------------------------------------------------
def cat_channels(tensors):
    """Channel concat of NDHWC tensors [N,Ci,D,H,W] -> [N,sum Ci,D,H,W]."""
    _require_cuda(*tensors)
    tensors = [ndhwc(t) for t in tensors]
    N, _, D, H, W = tensors[0].shape
    Ct = sum(t.shape[1] for t in tensors)
    out = new_act(N, Ct, D, H, W, tensors[0].device)
    off = 0
    for t in tensors:
        assert tuple(t.shape[2:]) == (D, H, W) and t.shape[0] == N
        check(_lib.lib().adell_copy_channels(_ptr(out), _ptr(t), N * D * H * W, Ct, t.shape[1], off,
                                             0, _stream()))
        off += t.shape[1]
    return out


def split_channels(full, sizes):
    full = ndhwc(full)
    N, Ct, D, H, W = full.shape
    outs, off = [], 0
    for c in sizes:
        t = new_act(N, c, D, H, W, full.device)
        check(_lib.lib().adell_copy_channels(_ptr(full), _ptr(t), N * D * H * W, Ct, c, off, 1,
                                             _stream()))
        outs.append(t)
        off += c
    return outs


def interp_nearest(x, size, backward_from=None):
    """Forward: x [N,C,Di,Hi,Wi] -> [N,C,*size]. With backward_from=(Di,Hi,Wi): x is the
    output gradient and the result the input gradient."""
    _require_cuda(x)
    x = ndhwc(x)
    N, C = x.shape[:2]
    if backward_from is None:
        Di, Hi, Wi = x.shape[2:]
        Do, Ho, Wo = size
        y = new_act(N, C, Do, Ho, Wo, x.device)
        check(_lib.lib().adell_interp_nearest_fwd(_ptr(x), _ptr(y), N, C, Di, Hi, Wi, Do, Ho, Wo,
                                                  _stream()))
        return y
    Di, Hi, Wi = backward_from
    Do, Ho, Wo = x.shape[2:]
    dx = new_act(N, C, Di, Hi, Wi, x.device)
    check(_lib.lib().adell_interp_nearest_bwd(_ptr(x), _ptr(dx), N, C, Di, Hi, Wi, Do, Ho, Wo,
                                              _stream()))
    return dx


def interp_linear(x, scales, backward_from=None, size=None):
    """torch.nn.Upsample(scale_factor=scales, mode="trilinear", align_corners=False) on
    [N,C,Di,Hi,Wi]; with size=(Do,Ho,Wo): F.interpolate(size=..., align_corners=True) instead;
    with backward_from=(Di,Hi,Wi): x is the output gradient, result = dX."""
    _require_cuda(x)
    x = ndhwc(x)
    N, C = x.shape[:2]
    align = int(size is not None)
    sd, sh, sw = (1.0, 1.0, 1.0) if align else (float(s) for s in scales)
    if backward_from is None:
        Di, Hi, Wi = x.shape[2:]
        Do, Ho, Wo = size if align else (int(math.floor(n * s))
                                         for n, s in zip((Di, Hi, Wi), (sd, sh, sw)))
        y = new_act(N, C, Do, Ho, Wo, x.device)
        check(_lib.lib().adell_interp_linear_fwd(_ptr(x), _ptr(y), N, C, Di, Hi, Wi, Do, Ho, Wo,
                                                 sd, sh, sw, align, _stream()))
        return y
    Di, Hi, Wi = backward_from
    Do, Ho, Wo = x.shape[2:]
    dx = new_act(N, C, Di, Hi, Wi, x.device)
    check(_lib.lib().adell_interp_linear_bwd(_ptr(x), _ptr(dx), N, C, Di, Hi, Wi, Do, Ho, Wo,
                                             sd, sh, sw, align, _stream()))
    return dx


def fold_x_taps(x, K, P, Cp=16):
    """[N,Cin,D,H,W] -> [N,Cp,D,H,W+2P-K+1] with channel kx*Cin+ci = x[..., ox-P+kx] (zero fill)."""
    _require_cuda(x)
    x = ndhwc(x)
    N, Cin, D, H, W = x.shape
    Wo = W + 2 * P - K + 1
    out = new_act(N, Cp, D, H, Wo, x.device)
    check(_lib.lib().adell_fold_x_taps(_ptr(x), _ptr(out), N, D, H, W, Cin, K, P, Cp, _stream()))
    return out


def scale_bc(x, s):
    """x [N,C,D,H,W] (NDHWC memory) times s [N,C]."""
    _require_cuda(x, s)
    x = ndhwc(x)
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    y = new_act(N, C, *x.shape[2:], x.device)
    sc = s.contiguous()
    check(_lib.lib().adell_scale_bc(_ptr(x), _ptr(sc), _ptr(y), N, V, C, _stream()))
    return y


def scale_bc_dscale(x, dy):
    """ds[n,c] = sum over voxels of dy * x."""
    x, dy = ndhwc(x), ndhwc(dy)
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    ws = _workspace(4 * _lib.lib().adell_scale_bc_dscale_workspace_floats(N, V, C), x.device)
    ds = torch.empty((N, C), device=x.device, dtype=torch.float32)
    check(_lib.lib().adell_scale_bc_dscale(_ptr(x), _ptr(dy), _ptr(ds), N, V, C, _ptr(ws),
                                           _stream()))
    return ds


def cse_apply(x, s, c, inv=None, acc=None):
    """y = acc + x * (s[n, v] + c[n, ch]) * inv[n]; x/acc [N,C,D,H,W] (NDHWC memory), s [N,1,D,H,W],
    c [N,C], inv [N] or None, acc or None."""
    _require_cuda(x, s, c, inv, acc)
    x = ndhwc(x)
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    s, c = s.contiguous(), c.contiguous()
    if s.numel() != N * V or c.numel() != N * C:
        raise AdellHipError(f"cse_apply: gates {tuple(s.shape)}, {tuple(c.shape)} do not fit "
                            f"x {tuple(x.shape)}")
    acc = None if acc is None else ndhwc(acc)
    inv = None if inv is None else inv.contiguous()
    y = new_act(N, C, *x.shape[2:], x.device)
    check(_lib.lib().adell_cse_apply(_ptr(x), _ptr(s), _ptr(c), _ptr(inv), _ptr(acc), _ptr(y), N, V,
                                     C, _stream()))
    return y


def cse_apply_bwd(x, dy, s, c, inv=None):
    """(dx, ds [N,1,D,H,W], dc [N,C]) of cse_apply."""
    _require_cuda(x, dy, s, c, inv)
    x, dy = ndhwc(x), ndhwc(dy)
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    s, c = s.contiguous(), c.contiguous()
    inv = None if inv is None else inv.contiguous()
    ws = _workspace(4 * _lib.lib().adell_cse_apply_bwd_workspace_floats(N, V, C), x.device)
    dx = new_act(N, C, *x.shape[2:], x.device)
    ds = torch.empty((N, 1, *x.shape[2:]), device=x.device, dtype=torch.float32)
    dc = torch.empty((N, C), device=x.device, dtype=torch.float32)
    check(_lib.lib().adell_cse_apply_bwd(_ptr(x), _ptr(dy), _ptr(s), _ptr(c), _ptr(inv), _ptr(dx),
                                         _ptr(ds), _ptr(dc), N, V, C, _ptr(ws), _stream()))
    return dx, ds, dc


def bcast_nc(g, shape, scale=1.0):
    """out[n, c, ...] = g[n, c] * scale, out of logical shape [N, C, D, H, W] (NDHWC memory)."""
    _require_cuda(g)
    N, C = shape[:2]
    V = int(math.prod(shape[2:]))
    out = new_act(N, C, *shape[2:], g.device)
    gc = g.contiguous()
    check(_lib.lib().adell_bcast_nc(_ptr(gc), _ptr(out), N, V, C, float(scale), _stream()))
    return out


def maxpool3d_fwd(x, kernel, stride, padding):
    _require_cuda(x)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    d = make_conv_desc(N, (D, H, W), C, 0, C, kernel, stride, padding)
    y = new_act(N, C, d.Do, d.Ho, d.Wo, x.device)
    idx = torch.empty((N, d.Do, d.Ho, d.Wo, C), device=x.device, dtype=torch.int32)
    check(_lib.lib().adell_maxpool3d_fwd(ctypes.byref(d), _ptr(x), _ptr(y), _ptr(idx), _stream()))
    return y, idx


def maxpool3d_bwd(dy, idx, in_shape, kernel, stride, padding):
    dy = ndhwc(dy)
    N, C, D, H, W = in_shape
    d = make_conv_desc(N, (D, H, W), C, 0, C, kernel, stride, padding)
    dx = new_act(N, C, D, H, W, dy.device)
    check(_lib.lib().adell_maxpool3d_bwd(ctypes.byref(d), _ptr(dy), _ptr(idx), _ptr(dx), _stream()))
    return dx


# ---- ConvNeXt / VICReg -------------------------------------------------------------------------
def dwconv3d_fwd(x, w, bias):
    _require_cuda(x, w, bias)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    kd, kh, kw = w.shape[2:]
    y = new_act(N, C, D, H, W, x.device)
    wc = w.contiguous()
    check(_lib.lib().adell_dwconv3d_fwd(N, C, D, H, W, kd, kh, kw, _ptr(x), _ptr(wc),
                                        _ptr(bias), _ptr(y), _stream()))
    return y


def dwconv3d_bwd_data(dy, w):
    dy = ndhwc(dy)
    N, C, D, H, W = dy.shape
    kd, kh, kw = w.shape[2:]
    dx = new_act(N, C, D, H, W, dy.device)
    wc = w.contiguous()
    check(_lib.lib().adell_dwconv3d_bwd_data(N, C, D, H, W, kd, kh, kw, _ptr(dy),
                                             _ptr(wc), _ptr(dx), _stream()))
    return dx


def dwconv3d_bwd_weight(x, dy, kshape, want_db):
    x, dy = ndhwc(x), ndhwc(dy)
    N, C, D, H, W = x.shape
    kd, kh, kw = kshape
    dw = torch.empty((C, 1, kd, kh, kw), device=x.device, dtype=torch.float32)
    db = torch.empty((C,), device=x.device, dtype=torch.float32) if want_db else None
    nws = _lib.lib().adell_dwconv3d_bwd_weight_workspace_floats(N, C, D, H, W, kd, kh, kw)
    ws = torch.empty((nws,), device=x.device, dtype=torch.float32) if nws else None
    check(_lib.lib().adell_dwconv3d_bwd_weight(N, C, D, H, W, kd, kh, kw, _ptr(x), _ptr(dy),
                                               _ptr(dw), _ptr(db), _ptr(ws), _stream()))
    return dw, db


def multi_copy(table, rows, dst):
    """table: int64 device tensor [rows, 3] of (src pointer, dst offset, numel <= 16384)."""
    _require_cuda(dst)
    if not table.is_cuda or table.dtype != torch.int64:
        raise _lib.AdellHipError("multi_copy: the table must be an int64 CUDA tensor")
    check(_lib.lib().adell_multi_copy(_ptr(table), int(rows), _ptr(dst), _stream()))


# ---- shifted-window (SWIN) token path -----------------------------------------------------
def gather_nd(x, dims, axes, out=None):
    """Flat contiguous gather of ``x`` (csrc/window.hip). ``dims``: [(size, axis, mult)] of the
    output, outermost first; ``axes``: [(extent, stride, shift)] of the input, in elements.
    ``out``: optional contiguous destination of exactly prod(sizes) elements."""
    import ctypes

    _require_cuda(x)
    nd, na = len(dims), len(axes)
    total = 1
    for d in dims:
        total *= int(d[0])
    if out is None:
        out = torch.empty((total,), device=x.device, dtype=torch.float32)
    else:
        _require_cuda(out)
        if out.numel() != total or not out.is_contiguous():
            raise AdellHipError("gather_nd: out must be contiguous with prod(sizes) elements")
    IntA, LongD, LongA = ctypes.c_int * nd, ctypes.c_long * nd, ctypes.c_long * na
    check(_lib.lib().adell_gather_nd(
        _ptr(x), _ptr(out), nd, IntA(*[int(d[0]) for d in dims]), IntA(*[int(d[1]) for d in dims]),
        LongD(*[int(d[2]) for d in dims]), na, LongA(*[int(a[0]) for a in axes]),
        LongA(*[int(a[1]) for a in axes]), LongA(*[int(a[2]) for a in axes]), _stream()))
    return out


def attention_strided_ok(T, A, Dv):
    return bool(_lib.lib().adell_attention_strided_ok(int(T), int(A), int(Dv)))


def attention_fwd_strided(q, k, v, out, strides, bias, B, H, T, A, Dv, scale, drop_p=0.0, seed=0,
                          offset=0):
    """q, k, v, out: tensors whose data pointers are the first rows of sequence 0 (flat views
    into packed buffers are fine); strides: 12 element strides, (item, head, row) for each of
    q, k, v, out. Writes ``out``; returns lse [B*H, T]."""
    _require_cuda(q, k, v, out, bias)
    bias = None if bias is None else bias.contiguous()
    lse = torch.empty((B * H, T), device=q.device, dtype=torch.float32)
    nb = 0 if bias is None else bias.numel() // (T * T)
    st = (ctypes.c_long * 12)(*[int(x) for x in strides])
    check(_lib.lib().adell_attention_fwd_strided(
        _ptr(q), _ptr(k), _ptr(v), _ptr(bias), nb, B, H, T, A, Dv, st, float(scale), float(drop_p),
        int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF, _ptr(out), _ptr(lse), _stream()))
    return lse


def attention_bwd_strided(q, k, v, out, dout, lse, dq, dk, dv, strides, bias, B, H, T, A, Dv, scale,
                          drop_p=0.0, seed=0, offset=0):
    """strides: 24 element strides, (item, head, row) for q, k, v, out, dout, dq, dk, dv."""
    _require_cuda(q, k, v, out, dout, lse, dq, dk, dv, bias)
    bias = None if bias is None else bias.contiguous()
    nb = 0 if bias is None else bias.numel() // (T * T)
    st = (ctypes.c_long * 24)(*[int(x) for x in strides])
    check(_lib.lib().adell_attention_bwd_strided(
        _ptr(q), _ptr(k), _ptr(v), _ptr(bias), nb, _ptr(out), _ptr(dout), _ptr(lse), B, H, T, A, Dv,
        st, float(scale), float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF,
        _ptr(dq), _ptr(dk), _ptr(dv), _stream()))


def layernorm_rows_fwd(x, rows, C, inner, so, si, gamma, beta, eps):
    """LayerNorm of ``rows`` rows of C <= 512 values read at (r//inner)*so + (r%inner)*si."""
    _require_cuda(x, gamma, beta)
    y = torch.empty((rows, C), device=x.device, dtype=torch.float32)
    mean = torch.empty((rows,), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    check(_lib.lib().adell_layernorm_rows_fwd(_ptr(x), rows, C, inner, so, si, _ptr(gamma),
                                              _ptr(beta), float(eps), _ptr(y), _ptr(mean),
                                              _ptr(rstd), _stream()))
    return y, mean, rstd


def layernorm_rows_bwd(x, dy, gamma, mean, rstd, rows, C, inner, so, si, dx, dso, dsi,
                       want_affine):
    """dx is written INTO ``dx`` at the strided row positions (dso, dsi)."""
    _require_cuda(x, dy, dx)
    dg = db = ws = None
    nbytes = 0
    if want_affine:
        dg = torch.empty((C,), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        nbytes = _lib.lib().adell_layernorm_rows_bwd_workspace(rows, C)
        ws = _workspace(nbytes, x.device)
    check(_lib.lib().adell_layernorm_rows_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(mean),
                                              _ptr(rstd), rows, C, inner, so, si, _ptr(dx), dso,
                                              dsi, _ptr(dg), _ptr(db), _ptr(ws),
                                              0 if ws is None else ws.numel() * 4, _stream()))
    return dg, db


def winattn_fwd(q, k, v, v_ts, v_hs, rel, mask, W, H, T, A, Dv, scale, drop_p, seed, offset):
    _require_cuda(q, k, v, rel, mask)
    out = torch.empty((W * T, H, Dv), device=q.device, dtype=torch.float32)
    lse = torch.empty((W * H * T,), device=q.device, dtype=torch.float32)
    n_mask = 0 if mask is None else mask.shape[0]
    check(_lib.lib().adell_winattn_fwd(_ptr(q), _ptr(k), _ptr(v), v_ts, v_hs, _ptr(rel),
                                       _ptr(mask), n_mask, W, H, T, A, Dv, float(scale),
                                       float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                       int(offset) & 0xFFFFFFFF, _ptr(out), _ptr(lse), _stream()))
    return out, lse


def winattn_bwd(q, k, v, v_ts, v_hs, rel, mask, o, dout, lse, W, H, T, A, Dv, scale, drop_p,
                seed, offset, dq, dk, dv, want_ds):
    _require_cuda(q, k, v, o, dout, lse, dq, dk, dv)
    n_mask = 0 if mask is None else mask.shape[0]
    ds = torch.empty((W, H * T * T), device=q.device, dtype=torch.float32) if want_ds else None
    check(_lib.lib().adell_winattn_bwd(_ptr(q), _ptr(k), _ptr(v), v_ts, v_hs, _ptr(rel),
                                       _ptr(mask), n_mask, _ptr(o), _ptr(dout), _ptr(lse), W, H, T,
                                       A, Dv, float(scale), float(drop_p),
                                       int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF,
                                       _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ds), _stream()))
    return ds


def gemm(M, N, K, A, lda, a_kc, B, ldb, b_kc, out=None, bias=None, residual=None):
    """out[M, N] = A x B (+ bias) (+ residual) on the fp32 MFMA GEMM (csrc/gemm.hip).
    a_kc: A(m,k) = A[m*lda + k] else A[k*lda + m]; b_kc: B(k,n) = B[n*ldb + k] else B[k*ldb + n]."""
    _require_cuda(A, B, bias, residual)
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    nws = _lib.lib().adell_gemm_f32_workspace_floats(M, N, K)
    ws = _workspace(nws * 4, A.device) if nws else None
    ldr = 0 if residual is None else N

    def run():
        check(_lib.lib().adell_gemm_f32(M, N, K, _ptr(A), lda, int(a_kc), _ptr(B), ldb, int(b_kc),
                                        _ptr(out), N, _ptr(bias), _ptr(residual), ldr, _ptr(ws),
                                        _stream()))
    _timed("adell_gemm_f32_kernel", 2.0 * M * N * K, run,
           lambda: f"{M}x{N}x{K} {'kc' if a_kc else 'outer'}/{'kc' if b_kc else 'outer'}",
           4.0 * (M * K + K * N + M * N * (2 if residual is not None else 1)))
    return out


def absmax_word(x):
    """Device word (int32 tensor holding float bits) with the absmax of a dense fp32 tensor: the
    operand scale of ``gemm_f16x3``."""
    _require_cuda(x)
    x = x.contiguous()
    word = torch.zeros(1, device=x.device, dtype=torch.int32)
    check(_lib.lib().adell_absmax_f32(_ptr(x), x.numel(), _ptr(word), _stream()))
    return word


def gemm_f16x3_ok(M, N, K, A, lda, a_kc, B, ldb, b_kc):
    if not FLAGS["gemm_f16x3"] or K < FLAGS["gemm_f16x3_min_k"]:
        return False
    if min(M, N) < FLAGS["gemm_f16x3_min_mn"]:
        # below the threshold only the streaming regime: >= 32 on both output sides and >= 64 k rows
        # along M or K (524 288 x 32 x 128: 178 -> 66 us on the rows kernel; its dW 188 -> 103 us)
        if FLAGS["gemm_f16x3_min_mn"] != 64 or min(M, N) < 32 or max(M, K) < 65536:
            return False
    # a handful of features over >= 64 k rows: the thread-per-output kernel of adell_gemm_f32
    # (csrc/gemm.hip, rows_small) streams them; 128-wide MFMA tiles run 2 097 152 x 32 x 8 at 1.3 TB/s
    if a_kc and K <= 64 and N <= 64 and N * K <= 512 and M >= 65536:
        return False
    return bool(_lib.lib().adell_gemm_f16x3_applicable(M, N, K, _ptr(A), lda, int(a_kc), _ptr(B), ldb,
                                                       int(b_kc)))


def gemm_f16x3(M, N, K, A, lda, a_kc, B, ldb, b_kc, a_amax=None, b_amax=None, out=None, bias=None,
               residual=None):
    """``gemm`` on the f16 MFMA with the error-compensated split (csrc/gemm_f16x3.hip);
    ``a_amax`` / ``b_amax``: None (scales chosen inside the kernel) or the ``absmax_word`` of the
    two operand tensors."""
    _require_cuda(A, B, bias, residual)
    if (a_amax is None) != (b_amax is None) or (a_amax is not None
                                                and not (a_amax.is_cuda and b_amax.is_cuda)):
        raise _lib.AdellHipError("gemm_f16x3: both absmax words (device tensors) or neither")
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    nws = _lib.lib().adell_gemm_f16x3_workspace_floats(M, N, K)
    ws = _workspace(nws * 4, A.device) if nws else None
    ldr = 0 if residual is None else N

    def run():
        check(_lib.lib().adell_gemm_f16x3(M, N, K, _ptr(A), lda, int(a_kc), _ptr(B), ldb, int(b_kc),
                                          _ptr(out), N, _ptr(bias), _ptr(residual), ldr,
                                          _ptr(a_amax), _ptr(b_amax), _ptr(ws), _stream()))
    # algorithmic bytes: each operand and the output once (+ the residual)
    _timed("adell_gemm_f16x3_kernel", 2.0 * M * N * K, run,
           lambda: f"{M}x{N}x{K} {'kc' if a_kc else 'outer'}/{'kc' if b_kc else 'outer'}",
           4.0 * (M * K + K * N + M * N * (2 if residual is not None else 1)))
    return out


def gemm_f16x3_act(M, N, K, A, lda, a_kc, B, ldb, b_kc, act, act_p=0.0, bias=None, residual=None,
                   want_act=False, dact_in=None):
    """``gemm_f16x3`` with an activation in the epilogue (csrc/gemm_f16x3.hip,
    adell_gemm_f16x3_act): ``want_act`` -> returns (C, act(C)); ``dact_in`` [M, N] (a saved
    pre-activation) -> returns (C * act'(dact_in), None)."""
    _require_cuda(A, B, bias, residual, dact_in)
    out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    act_out = torch.empty_like(out) if want_act else None
    nws = _lib.lib().adell_gemm_f16x3_workspace_floats(M, N, K)
    ws = _workspace(nws * 4, A.device) if nws else None
    ldr = 0 if residual is None else N

    def run():
        check(_lib.lib().adell_gemm_f16x3_act(M, N, K, _ptr(A), lda, int(a_kc), _ptr(B), ldb,
                                              int(b_kc), _ptr(out), N, _ptr(bias), _ptr(residual),
                                              ldr, None, None, _ptr(ws), _lib.ACT_IDS[act],
                                              float(act_p), _ptr(act_out), _ptr(dact_in), _stream()))
    extra = (1 if residual is not None else 0) + (1 if want_act else 0) + (1 if dact_in is not None else 0)
    _timed("adell_gemm_f16x3_kernel", 2.0 * M * N * K, run,
           lambda: f"{M}x{N}x{K} {'kc' if a_kc else 'outer'}/{'kc' if b_kc else 'outer'} act",
           4.0 * (M * K + K * N + M * N * (1 + extra)))
    return out, act_out


def vicreg_fwd(x1, x2, min_var, eps):
    _require_cuda(x1, x2)
    x1, x2 = x1.contiguous(), x2.contiguous()
    B, D = x1.shape
    scratch = torch.empty(_lib.lib().adell_vicreg_scratch_floats(B, D), device=x1.device,
                          dtype=torch.float32)
    out = torch.empty(3, device=x1.device, dtype=torch.float32)
    check(_lib.lib().adell_vicreg_fwd(_ptr(x1), _ptr(x2), B, D, float(min_var), float(eps),
                                      _ptr(scratch), _ptr(out), _stream()))
    return out, scratch


def vicreg_bwd(x1, x2, scratch, min_var, eps, g, need1, need2):
    B, D = x1.shape
    if not (need1 or need2):
        return None, None
    g = g.contiguous().float()
    dx1 = torch.empty_like(x1) if need1 else None
    dx2 = torch.empty_like(x2) if need2 else None
    check(_lib.lib().adell_vicreg_bwd(_ptr(x1), _ptr(x2), B, D, float(min_var), float(eps),
                                      _ptr(scratch), _ptr(g), _ptr(dx1), _ptr(dx2), _stream()))
    return dx1, dx2


PAIR_LOSS_KINDS = {"simsiam": 0, "byol": 1, "ntxent": 2}


def pair_loss_fwd(x1, x2, kind, temperature=1.0, apply_relu=False):
    """(loss [1], scratch) of a cosine-similarity loss between two [B, D] embedding batches."""
    _require_cuda(x1, x2)
    x1, x2 = x1.contiguous(), x2.contiguous()
    if x1.dim() != 2 or x1.shape != x2.shape:
        raise AdellHipError(f"pair_loss: two [B, D] tensors of one shape expected, got "
                            f"{tuple(x1.shape)} and {tuple(x2.shape)}")
    B, D = x1.shape
    scratch = torch.empty(max(_lib.lib().adell_pair_loss_scratch_floats(B, D), 1),
                          device=x1.device, dtype=torch.float32)
    loss = torch.empty(1, device=x1.device, dtype=torch.float32)
    check(_lib.lib().adell_pair_loss_fwd(_ptr(x1), _ptr(x2), B, D, PAIR_LOSS_KINDS[kind],
                                         float(temperature), int(bool(apply_relu)),
                                         _ptr(scratch), _ptr(loss), _stream()))
    return loss, scratch


def pair_loss_bwd(x1, x2, kind, temperature, apply_relu, scratch, g, need1, need2):
    if not (need1 or need2):
        return None, None
    B, D = x1.shape
    g = g.reshape(1).contiguous().float()
    dx1 = torch.empty_like(x1) if need1 else None
    dx2 = torch.empty_like(x2) if need2 else None
    check(_lib.lib().adell_pair_loss_bwd(_ptr(x1), _ptr(x2), B, D, PAIR_LOSS_KINDS[kind],
                                         float(temperature), int(bool(apply_relu)),
                                         _ptr(scratch), _ptr(g), _ptr(dx1), _ptr(dx2), _stream()))
    return dx1, dx2


def loco_loss_fwd(f1, f2, temperature, eps):
    """Per-item local contrastive loss [B] of two NDHWC feature maps [B, C, *spatial]
    (semi_supervised_segmentation/losses.py:498-526)."""
    _require_cuda(f1, f2)
    f1, f2 = ndhwc(f1), ndhwc(f2)
    if f1.shape != f2.shape:
        raise AdellHipError(f"loco_loss: feature shapes differ ({tuple(f1.shape)} vs "
                            f"{tuple(f2.shape)})")
    B, C = f1.shape[0], f1.shape[1]
    S = f1.numel() // (B * C)
    nbytes = _lib.lib().adell_loco_loss_workspace(B, S, C)
    check(min(nbytes, 0))
    ws = _workspace(nbytes, f1.device)
    loss = torch.empty(B, device=f1.device, dtype=torch.float32)
    check(_lib.lib().adell_loco_loss_fwd(_ptr(f1), _ptr(f2), B, S, C, float(temperature),
                                         float(eps), _ptr(loss), _ptr(ws), ws.numel() * 4,
                                         _stream()))
    return loss


def loco_loss_bwd(f1, f2, gloss, temperature, eps, need1, need2):
    if not (need1 or need2):
        return None, None
    f1, f2 = ndhwc(f1), ndhwc(f2)
    B, C = f1.shape[0], f1.shape[1]
    S = f1.numel() // (B * C)
    gloss = gloss.contiguous().float()
    df1 = new_act(*f1.shape, f1.device) if need1 else None
    df2 = new_act(*f2.shape, f2.device) if need2 else None
    check(_lib.lib().adell_loco_loss_bwd(_ptr(f1), _ptr(f2), _ptr(gloss), B, S, C,
                                         float(temperature), float(eps), _ptr(df1), _ptr(df2),
                                         _stream()))
    return df1, df2


# ---- class-axis softmax head, channel max pooling -------------------------------------------------
def channel_softmax_fwd(x):
    """softmax over dim 1 of a logical [N, C, *spatial] tensor (NDHWC memory), C <= 32."""
    _require_cuda(x)
    x = ndhwc(x)
    C = x.shape[1]
    y = new_act(*x.shape, x.device)
    check(_lib.lib().adell_channel_softmax_fwd(_ptr(x), _ptr(y), x.numel() // C, C, _stream()))
    return y


def channel_softmax_bwd(y, dy):
    _require_cuda(y, dy)
    y, dy = ndhwc(y), ndhwc(dy)
    C = y.shape[1]
    dx = new_act(*y.shape, y.device)
    check(_lib.lib().adell_channel_softmax_bwd(_ptr(y), _ptr(dy), _ptr(dx), y.numel() // C, C,
                                               _stream()))
    return dx


def channel_max_fwd(x):
    """[N, C, D, H, W] -> ([N, C] maxima over the voxels, [N, C] int32 voxel indices)."""
    _require_cuda(x)
    x = ndhwc(x)
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    out = torch.empty((N, C), device=x.device, dtype=torch.float32)
    arg = torch.empty((N, C), device=x.device, dtype=torch.int32)
    check(_lib.lib().adell_channel_max_fwd(_ptr(x), _ptr(out), _ptr(arg), N, V, C, _stream()))
    return out, arg


def channel_max_bwd(dout, arg, shape):
    _require_cuda(dout)
    if not arg.is_cuda or arg.dtype != torch.int32:
        raise _lib.AdellHipError("channel_max_bwd: arg must be the int32 CUDA tensor of the forward")
    N, C = shape[:2]
    V = int(math.prod(shape[2:]))
    dx = new_act(N, C, *shape[2:], dout.device)
    dc = dout.contiguous()
    check(_lib.lib().adell_channel_max_bwd(_ptr(dc), _ptr(arg), _ptr(dx), N, V, C, _stream()))
    return dx


# ---- element-wise segmentation losses (loss_factory members beyond dice + focal) ---------------
SEG_LOSS_KINDS = {"binary_cross_entropy": 0, "cat_cross_entropy": 1, "mc_focal": 2, "mc_dice": 3}


def seg_loss_fwd(kind, p, t, cw, eps, scale, ls, gamma, smooth, w_pos):
    """p, t: [B, V, C] contiguous (NDHWC order). Returns (loss [B], sums [B, C, 2])."""
    _require_cuda(p, t, cw)
    B, V, C = p.shape
    ws = _workspace(_lib.lib().adell_seg_loss_workspace(B, V, C), p.device)
    loss = torch.empty((B,), device=p.device, dtype=torch.float32)
    sums = torch.empty((B, C, 2), device=p.device, dtype=torch.float32)
    check(_lib.lib().adell_seg_loss_fwd(kind, _ptr(p), _ptr(t), _ptr(cw), B, V, C, eps, scale, ls,
                                        gamma, smooth, w_pos, _ptr(loss), _ptr(sums), _ptr(ws),
                                        ws.numel() * 4, _stream()))
    return loss, sums


def seg_loss_bwd(kind, p, t, cw, eps, scale, ls, gamma, smooth, w_pos, sums, gout):
    _require_cuda(p, t, cw, sums, gout)
    B, V, C = p.shape
    dp = torch.empty_like(p)
    gc = gout.contiguous()
    check(_lib.lib().adell_seg_loss_bwd(kind, _ptr(p), _ptr(t), _ptr(cw), B, V, C, eps, scale, ls,
                                        gamma, smooth, w_pos, _ptr(sums), _ptr(gc), _ptr(dp),
                                        _stream()))
    return dp


OPTIM_KINDS = {"adamax": 0, "adagrad": 1, "nadam": 2, "radam": 3, "rmsprop": 4}


def optim_step(kind, param, grad, s1, s2, weight_decay, eps, grad_scale, c5):
    """One fused step of adamax / adagrad / nadam / radam / rmsprop on flat fp32 buffers."""
    _require_cuda(param, grad, s1, s2)
    arr = (ctypes.c_float * 5)(*[float(v) for v in (list(c5) + [0.0] * 5)[:5]])
    check(_lib.lib().adell_optim_step(OPTIM_KINDS[kind], _ptr(param), _ptr(grad), _ptr(s1),
                                      _ptr(s2), param.numel(), weight_decay, eps, grad_scale, arr,
                                      _stream()))
    _weights_changed()


# ---- device-side batch augmentation (csrc/augment.hip) -------------------------------------------
def item_stats(x):
    """[N, 4] = (min, max, mean, population std) of every batch item of ``x`` (any layout: the
    statistics are over all elements of the item)."""
    _require_cuda(x)
    N = x.shape[0]
    per = x.numel() // N
    assert x.is_contiguous() or ndhwc(x).data_ptr() == x.data_ptr()
    out = torch.empty((N, 4), device=x.device, dtype=torch.float32)
    ws = _workspace(_lib.lib().adell_item_stats_workspace(N, per), x.device)
    check(_lib.lib().adell_item_stats(_ptr(x), N, per, _ptr(out), _ptr(ws), ws.numel() * 4, _stream()))
    return out


def aug_intensity(x, params, seed=0, rng_offset=0):
    """Gamma contrast -> std shift -> Rician noise of every item in one pass; ``params`` [N, 8]
    device rows {min, range, gamma, shift, noise std, 0, 0, 0} (adell_aug_intensity)."""
    _require_cuda(x, params)
    N = x.shape[0]
    out = torch.empty_like(x)
    check(_lib.lib().adell_aug_intensity(_ptr(x), _ptr(out), N, x.numel() // N, _ptr(params),
                                         int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_offset) & 0xFFFFFFFF,
                                         _stream()))
    return out


def affine_sample(x, theta, linear=True, pad_mode="reflection"):
    """Affine resampling of [N, C, D, H, W] (NDHWC memory) volumes about their centres; ``theta``
    [N, 12] device rows of the 3 x 4 voxel-space matrices (adell_affine_sample)."""
    _require_cuda(x, theta)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    out = new_act(N, C, D, H, W, x.device)
    check(_lib.lib().adell_affine_sample(_ptr(x), _ptr(out), N, D, H, W, C, _ptr(theta),
                                         1 if linear else 0,
                                         {"zeros": 0, "border": 1, "reflection": 2}[pad_mode],
                                         _stream()))
    return out


def axis_filter(x, taps, axis):
    """1-D filter along spatial axis 0 / 1 / 2 (zero padding); ``taps`` [N, 2 R + 1] device rows."""
    _require_cuda(x, taps)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    out = new_act(N, C, D, H, W, x.device)
    check(_lib.lib().adell_axis_filter(_ptr(x), _ptr(out), N, D, H, W, C, int(axis), _ptr(taps),
                                       (taps.shape[1] - 1) // 2, _stream()))
    return out


def bias_field(x, coef):
    """x * exp(Legendre polynomial field); ``coef`` [N, 64] device rows (dense 4 x 4 x 4 cube)."""
    _require_cuda(x, coef)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    out = new_act(N, C, D, H, W, x.device)
    check(_lib.lib().adell_bias_field(_ptr(x), _ptr(out), N, D, H, W, C, _ptr(coef), _stream()))
    return out


def axis_lut_sample(x, lut, linear=True):
    """Resampling through per-axis coordinate tables ``lut`` [N, D + H + W] (border padding)."""
    _require_cuda(x, lut)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    out = new_act(N, C, D, H, W, x.device)
    check(_lib.lib().adell_axis_lut_sample(_ptr(x), _ptr(out), N, D, H, W, C, _ptr(lut),
                                           1 if linear else 0, _stream()))
    return out


def gibbs_lowpass(x, radius):
    """k-space low-pass per item: spectrum zeroed outside ``radius`` [N] (device) about the centre of
    the shifted spectrum."""
    _require_cuda(x, radius)
    x = ndhwc(x)
    N, C, D, H, W = x.shape
    out = new_act(N, C, D, H, W, x.device)
    nbytes = _lib.lib().adell_gibbs_workspace(N, D, H, W, C)
    if nbytes < 0:
        check(int(nbytes))
    ws = _workspace(nbytes, x.device)
    check(_lib.lib().adell_gibbs_lowpass(_ptr(x), _ptr(out), N, D, H, W, C, _ptr(radius), _ptr(ws),
                                         ws.numel() * 4, _stream()))
    return out


def resize_linear(x, size):
    """F.interpolate(x, size=size, mode="trilinear", align_corners=False) on [N, C, D, H, W]."""
    _require_cuda(x)
    x = ndhwc(x)
    N, C, Di, Hi, Wi = x.shape
    Do, Ho, Wo = (int(v) for v in size)
    y = new_act(N, C, Do, Ho, Wo, x.device)
    check(_lib.lib().adell_interp_linear_fwd(_ptr(x), _ptr(y), N, C, Di, Hi, Wi, Do, Ho, Wo,
                                             Do / Di, Ho / Hi, Wo / Wi, 0, _stream()))
    return y
