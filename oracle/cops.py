"""ctypes front-end of oracle/c/adell_oracle.c (TEST INFRASTRUCTURE ONLY).

All arrays are numpy float32 in torch's canonical layouts (NCDHW activations,
[Cout,Cin,kD,kH,kW] conv weights, [Cin,Cout,kD,kH,kW] transposed-conv weights).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libadell_oracle.so")
_lib = None

ACT = {"identity": 0, "swish": 1, "silu": 1, "relu": 2, "leaky_relu": 3, "prelu": 4,
       "gelu": 5, "sigmoid": 6, "tanh": 7, "elu": 8}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(i) for i in v)


def conv3d(x, w, b=None, stride=1, padding=0):
    x, w, b = _f32(x), _f32(w), _f32(b)
    s, p = _triple(stride), _triple(padding)
    N, Cin, D, H, W = x.shape
    Cout, _, KD, KH, KW = w.shape
    out = [(d + 2 * pp - k) // ss + 1 for d, pp, k, ss in zip((D, H, W), p, (KD, KH, KW), s)]
    y = np.empty((N, Cout, *out), np.float32)
    lib().oracle_conv3d(_p(x), _p(w), _p(b), _p(y), N, Cin, D, H, W, Cout, KD, KH, KW, *s, *p)
    return y


def conv3d_bwd(x, w, dy, stride=1, padding=0, need_db=True):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    s, p = _triple(stride), _triple(padding)
    N, Cin, D, H, W = x.shape
    Cout, _, KD, KH, KW = w.shape
    dx, dw = np.empty_like(x), np.empty_like(w)
    db = np.empty((Cout,), np.float32) if need_db else None
    lib().oracle_conv3d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, Cin, D, H, W,
                            Cout, KD, KH, KW, *s, *p)
    return dx, dw, db


def conv_transpose3d(x, w, b=None, stride=2, padding=0):
    x, w, b = _f32(x), _f32(w), _f32(b)
    s, p = _triple(stride), _triple(padding)
    N, Cin, D, H, W = x.shape
    _, Cout, KD, KH, KW = w.shape
    out = [(d - 1) * ss - 2 * pp + k for d, pp, k, ss in zip((D, H, W), p, (KD, KH, KW), s)]
    y = np.empty((N, Cout, *out), np.float32)
    lib().oracle_conv_transpose3d(_p(x), _p(w), _p(b), _p(y), N, Cin, D, H, W, Cout, KD, KH,
                                  KW, *s, *p)
    return y


def conv_transpose3d_bwd(x, w, dy, stride=2, padding=0):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    s, p = _triple(stride), _triple(padding)
    N, Cin, D, H, W = x.shape
    _, Cout, KD, KH, KW = w.shape
    dx, dw = np.empty_like(x), np.empty_like(w)
    db = np.empty((Cout,), np.float32)
    lib().oracle_conv_transpose3d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, Cin,
                                      D, H, W, Cout, KD, KH, KW, *s, *p)
    return dx, dw, db


def norm_act(x, norm=True, eps=1e-5, act="swish", act_p=0.0):
    x = _f32(x)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    out = np.empty_like(x)
    lib().oracle_norm_act(_p(x), _p(out), N, C, ctypes.c_long(S), int(bool(norm)),
                          ctypes.c_float(eps), ACT[act], ctypes.c_float(act_p))
    return out


def norm_act_bwd(x, dout, norm=True, eps=1e-5, act="swish", act_p=0.0):
    x, dout = _f32(x), _f32(dout)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    dx = np.empty_like(x)
    lib().oracle_norm_act_bwd(_p(x), _p(dout), _p(dx), N, C, ctypes.c_long(S),
                              int(bool(norm)), ctypes.c_float(eps), ACT[act],
                              ctypes.c_float(act_p))
    return dx


def dice_focal(p, t, smooth=1e-5, dice_eps=1e-6, gamma=1.0, focal_eps=1e-6,
               grad=False, gscale_dice=1.0, gscale_focal=1.0):
    p, t = _f32(p), _f32(t)
    B = p.shape[0]
    S = int(np.prod(p.shape[1:]))
    dice, focal = np.empty((B,), np.float32), np.empty((B,), np.float32)
    dp = np.empty_like(p) if grad else None
    lib().oracle_dice_focal(_p(p), _p(t), B, ctypes.c_long(S), ctypes.c_float(smooth),
                            ctypes.c_float(dice_eps), ctypes.c_float(gamma),
                            ctypes.c_float(focal_eps), _p(dice), _p(focal), _p(dp),
                            ctypes.c_float(gscale_dice), ctypes.c_float(gscale_focal))
    return (dice, focal, dp) if grad else (dice, focal)


def sgd_nesterov(p, g, buf, lr, momentum=0.99, wd=0.0, nesterov=True, first=False):
    assert p.dtype == np.float32 and g.dtype == np.float32 and buf.dtype == np.float32
    lib().oracle_sgd_nesterov(_p(p), _p(g), _p(buf), ctypes.c_size_t(p.size),
                              ctypes.c_float(lr), ctypes.c_float(momentum),
                              ctypes.c_float(wd), int(nesterov), int(first))
