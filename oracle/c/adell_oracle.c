/*
 * adell_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, obviously-correct restatement (nested loops, fp64 accumulation) of
 * the torch operators on the adell_mri U-Net hot path, in torch's canonical
 * NCDHW layout. It exists so that the HIP kernels in adell_mri_amd/csrc can be
 * checked op-by-op on the GPU box, where neither the reference nor (for these
 * checks) torch's own CPU kernels are the thing being trusted. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path never does.
 *
 * Semantics restated (reference call sites; the arithmetic is torch's):
 *   conv3d            torch.nn.Conv3d        adell_mri/modules/segmentation/unet.py:260-273
 *   conv_transpose3d  torch.nn.ConvTranspose3d  unet.py:445-458
 *   instance_norm     torch.nn.InstanceNorm3d (affine=False, eps 1e-5, biased var)  adn_fn.py:22-26
 *   activations       adell_mri/modules/activations.py:6-31
 *   dice / focal      adell_mri/modules/segmentation/losses.py:14-54,112-164,251-292
 *   sgd (nesterov)    torch.optim.SGD as configured at segmentation/pl.py:563-569
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX5(n, c, d, h, w, C, D, H, W) \
  (((((size_t)(n) * (C) + (c)) * (D) + (d)) * (H) + (h)) * (W) + (w))

/* y[N][Cout][Do][Ho][Wo] = conv3d(x[N][Cin][D][H][W], w[Cout][Cin][K..]) + b */
void oracle_conv3d(const float* x, const float* w, const float* b, float* y, int N,
                   int Cin, int D, int H, int W, int Cout, int KD, int KH, int KW,
                   int SD, int SH, int SW, int PD, int PH, int PW) {
  const int Do = (D + 2 * PD - KD) / SD + 1, Ho = (H + 2 * PH - KH) / SH + 1,
            Wo = (W + 2 * PW - KW) / SW + 1;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int od = 0; od < Do; ++od)
        for (int oh = 0; oh < Ho; ++oh)
          for (int ow = 0; ow < Wo; ++ow) {
            double acc = b ? (double)b[co] : 0.0;
            for (int ci = 0; ci < Cin; ++ci)
              for (int kd = 0; kd < KD; ++kd) {
                const int id = od * SD - PD + kd;
                if (id < 0 || id >= D) continue;
                for (int kh = 0; kh < KH; ++kh) {
                  const int ih = oh * SH - PH + kh;
                  if (ih < 0 || ih >= H) continue;
                  for (int kw = 0; kw < KW; ++kw) {
                    const int iw = ow * SW - PW + kw;
                    if (iw < 0 || iw >= W) continue;
                    acc += (double)x[IDX5(n, ci, id, ih, iw, Cin, D, H, W)] *
                           (double)w[IDX5(co, ci, kd, kh, kw, Cin, KD, KH, KW)];
                  }
                }
              }
            y[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)] = (float)acc;
          }
}

/* gradients of oracle_conv3d; any of dx / dw / db may be NULL */
void oracle_conv3d_bwd(const float* x, const float* w, const float* dy, float* dx,
                       float* dw, float* db, int N, int Cin, int D, int H, int W,
                       int Cout, int KD, int KH, int KW, int SD, int SH, int SW,
                       int PD, int PH, int PW) {
  const int Do = (D + 2 * PD - KD) / SD + 1, Ho = (H + 2 * PH - KH) / SH + 1,
            Wo = (W + 2 * PW - KW) / SW + 1;
  const size_t nx = (size_t)N * Cin * D * H * W;
  const size_t nw = (size_t)Cout * Cin * KD * KH * KW;
  double* dxa = dx ? (double*)calloc(nx, sizeof(double)) : NULL;
  double* dwa = dw ? (double*)calloc(nw, sizeof(double)) : NULL;
  double* dba = db ? (double*)calloc((size_t)Cout, sizeof(double)) : NULL;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int od = 0; od < Do; ++od)
        for (int oh = 0; oh < Ho; ++oh)
          for (int ow = 0; ow < Wo; ++ow) {
            const double g = (double)dy[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)];
            if (dba) dba[co] += g;
            for (int ci = 0; ci < Cin; ++ci)
              for (int kd = 0; kd < KD; ++kd) {
                const int id = od * SD - PD + kd;
                if (id < 0 || id >= D) continue;
                for (int kh = 0; kh < KH; ++kh) {
                  const int ih = oh * SH - PH + kh;
                  if (ih < 0 || ih >= H) continue;
                  for (int kw = 0; kw < KW; ++kw) {
                    const int iw = ow * SW - PW + kw;
                    if (iw < 0 || iw >= W) continue;
                    const size_t xi = IDX5(n, ci, id, ih, iw, Cin, D, H, W);
                    const size_t wi = IDX5(co, ci, kd, kh, kw, Cin, KD, KH, KW);
                    if (dxa) dxa[xi] += g * (double)w[wi];
                    if (dwa) dwa[wi] += g * (double)x[xi];
                  }
                }
              }
          }
  if (dx) { for (size_t i = 0; i < nx; ++i) dx[i] = (float)dxa[i]; free(dxa); }
  if (dw) { for (size_t i = 0; i < nw; ++i) dw[i] = (float)dwa[i]; free(dwa); }
  if (db) { for (int i = 0; i < Cout; ++i) db[i] = (float)dba[i]; free(dba); }
}

/* y[N][Cout][Do..] = conv_transpose3d(x[N][Cin][D..], w[Cin][Cout][K..]) + b,
 * Do = (D-1)*S - 2P + K */
void oracle_conv_transpose3d(const float* x, const float* w, const float* b, float* y,
                             int N, int Cin, int D, int H, int W, int Cout, int KD,
                             int KH, int KW, int SD, int SH, int SW, int PD, int PH,
                             int PW) {
  const int Do = (D - 1) * SD - 2 * PD + KD, Ho = (H - 1) * SH - 2 * PH + KH,
            Wo = (W - 1) * SW - 2 * PW + KW;
  const size_t ny = (size_t)N * Cout * Do * Ho * Wo;
  double* ya = (double*)calloc(ny, sizeof(double));
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int id = 0; id < D; ++id)
        for (int ih = 0; ih < H; ++ih)
          for (int iw = 0; iw < W; ++iw) {
            const double xv = (double)x[IDX5(n, ci, id, ih, iw, Cin, D, H, W)];
            for (int co = 0; co < Cout; ++co)
              for (int kd = 0; kd < KD; ++kd) {
                const int od = id * SD - PD + kd;
                if (od < 0 || od >= Do) continue;
                for (int kh = 0; kh < KH; ++kh) {
                  const int oh = ih * SH - PH + kh;
                  if (oh < 0 || oh >= Ho) continue;
                  for (int kw = 0; kw < KW; ++kw) {
                    const int ow = iw * SW - PW + kw;
                    if (ow < 0 || ow >= Wo) continue;
                    ya[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)] +=
                        xv * (double)w[IDX5(ci, co, kd, kh, kw, Cout, KD, KH, KW)];
                  }
                }
              }
          }
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co) {
      const size_t base = ((size_t)n * Cout + co) * Do * Ho * Wo;
      for (size_t i = 0; i < (size_t)Do * Ho * Wo; ++i)
        y[base + i] = (float)(ya[base + i] + (b ? (double)b[co] : 0.0));
    }
  free(ya);
}

void oracle_conv_transpose3d_bwd(const float* x, const float* w, const float* dy,
                                 float* dx, float* dw, float* db, int N, int Cin,
                                 int D, int H, int W, int Cout, int KD, int KH, int KW,
                                 int SD, int SH, int SW, int PD, int PH, int PW) {
  const int Do = (D - 1) * SD - 2 * PD + KD, Ho = (H - 1) * SH - 2 * PH + KH,
            Wo = (W - 1) * SW - 2 * PW + KW;
  const size_t nx = (size_t)N * Cin * D * H * W;
  const size_t nw = (size_t)Cin * Cout * KD * KH * KW;
  double* dxa = dx ? (double*)calloc(nx, sizeof(double)) : NULL;
  double* dwa = dw ? (double*)calloc(nw, sizeof(double)) : NULL;
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int id = 0; id < D; ++id)
        for (int ih = 0; ih < H; ++ih)
          for (int iw = 0; iw < W; ++iw) {
            const size_t xi = IDX5(n, ci, id, ih, iw, Cin, D, H, W);
            for (int co = 0; co < Cout; ++co)
              for (int kd = 0; kd < KD; ++kd) {
                const int od = id * SD - PD + kd;
                if (od < 0 || od >= Do) continue;
                for (int kh = 0; kh < KH; ++kh) {
                  const int oh = ih * SH - PH + kh;
                  if (oh < 0 || oh >= Ho) continue;
                  for (int kw = 0; kw < KW; ++kw) {
                    const int ow = iw * SW - PW + kw;
                    if (ow < 0 || ow >= Wo) continue;
                    const double g =
                        (double)dy[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)];
                    const size_t wi = IDX5(ci, co, kd, kh, kw, Cout, KD, KH, KW);
                    if (dxa) dxa[xi] += g * (double)w[wi];
                    if (dwa) dwa[wi] += g * (double)x[xi];
                  }
                }
              }
          }
  if (db) {
    for (int co = 0; co < Cout; ++co) {
      double s = 0.0;
      for (int n = 0; n < N; ++n) {
        const size_t base = ((size_t)n * Cout + co) * Do * Ho * Wo;
        for (size_t i = 0; i < (size_t)Do * Ho * Wo; ++i) s += (double)dy[base + i];
      }
      db[co] = (float)s;
    }
  }
  if (dx) { for (size_t i = 0; i < nx; ++i) dx[i] = (float)dxa[i]; free(dxa); }
  if (dw) { for (size_t i = 0; i < nw; ++i) dw[i] = (float)dwa[i]; free(dwa); }
}

/* activation ids follow include/adell_hip.h */
static double act_fwd(int act, double x, double p) {
  switch (act) {
    case 1: return x / (1.0 + exp(-x));
    case 2: return x > 0 ? x : 0.0;
    case 3: case 4: return x > 0 ? x : p * x;
    case 5: return 0.5 * x * (1.0 + erf(x * 0.70710678118654752));
    case 6: return 1.0 / (1.0 + exp(-x));
    case 7: return tanh(x);
    case 8: return x > 0 ? x : p * (exp(x) - 1.0);
    default: return x;
  }
}
static double act_grad(int act, double x, double p) {
  switch (act) {
    case 1: { double s = 1.0 / (1.0 + exp(-x)); return s * (1.0 + x * (1.0 - s)); }
    case 2: return x > 0 ? 1.0 : 0.0;
    case 3: case 4: return x > 0 ? 1.0 : p;
    case 5: return 0.5 * (1.0 + erf(x * 0.70710678118654752)) +
                   x * 0.39894228040143268 * exp(-0.5 * x * x);
    case 6: { double s = 1.0 / (1.0 + exp(-x)); return s * (1.0 - s); }
    case 7: { double t = tanh(x); return 1.0 - t * t; }
    case 8: return x > 0 ? 1.0 : p * exp(x);
    default: return 1.0;
  }
}

/* out = act(instance_norm(x)) over x[N][C][S] (S = D*H*W); norm=0 skips the norm */
void oracle_norm_act(const float* x, float* out, int N, int C, long S, int norm,
                     float eps, int act, float act_p) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((size_t)n * C + c) * S;
      float* o = out + ((size_t)n * C + c) * S;
      double m = 0.0, r = 1.0;
      if (norm) {
        double s1 = 0.0;
        for (long i = 0; i < S; ++i) s1 += p[i];
        m = s1 / (double)S;
        double s2 = 0.0;
        for (long i = 0; i < S; ++i) s2 += (p[i] - m) * (p[i] - m);
        r = 1.0 / sqrt(s2 / (double)S + (double)eps);
      }
      for (long i = 0; i < S; ++i) o[i] = (float)act_fwd(act, (p[i] - m) * r, act_p);
    }
}

/* dx of oracle_norm_act given dout */
void oracle_norm_act_bwd(const float* x, const float* dout, float* dx, int N, int C,
                         long S, int norm, float eps, int act, float act_p) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float* p = x + ((size_t)n * C + c) * S;
      const float* g = dout + ((size_t)n * C + c) * S;
      float* o = dx + ((size_t)n * C + c) * S;
      double m = 0.0, r = 1.0;
      if (norm) {
        double s1 = 0.0;
        for (long i = 0; i < S; ++i) s1 += p[i];
        m = s1 / (double)S;
        double s2 = 0.0;
        for (long i = 0; i < S; ++i) s2 += (p[i] - m) * (p[i] - m);
        r = 1.0 / sqrt(s2 / (double)S + (double)eps);
      }
      double a1 = 0.0, a2 = 0.0;
      for (long i = 0; i < S; ++i) {
        const double h = (p[i] - m) * r;
        const double dh = (double)g[i] * act_grad(act, h, act_p);
        a1 += dh;
        a2 += dh * h;
      }
      a1 /= (double)S;
      a2 /= (double)S;
      for (long i = 0; i < S; ++i) {
        const double h = (p[i] - m) * r;
        const double dh = (double)g[i] * act_grad(act, h, act_p);
        o[i] = (float)(norm ? r * (dh - a1 - h * a2) : dh);
      }
    }
}

/* Binary generalised dice + binary focal loss on probabilities p[B][S] and
 * targets t[B][S] (single foreground channel), restating losses.py:14-54,
 * 251-292 (dice: weight 1, scale 1) and :112-164 (focal: alpha 1, threshold
 * 0.5, scale 1, no label smoothing). Writes per-item losses dice[B], focal[B];
 * when dp != NULL also d(mean_b dice + mean_b focal)/dp scaled by gscale. */
void oracle_dice_focal(const float* p, const float* t, int B, long S, float smooth,
                       float dice_eps, float gamma, float focal_eps, float* dice,
                       float* focal, float* dp, float gscale_dice,
                       float gscale_focal) {
  for (int b = 0; b < B; ++b) {
    const float* pp = p + (size_t)b * S;
    const float* tt = t + (size_t)b * S;
    double num = 0.0, den = 0.0, fl = 0.0;
    for (long i = 0; i < S; ++i) {
      const double pi = pp[i], ti = tt[i];
      double a = ti * pi;
      if (a < 0) a = 0;
      num += a;
      double d = ti + pi + (double)smooth;
      if (d < dice_eps) d = dice_eps;
      den += d;
      const double pc = pi > focal_eps ? pi : focal_eps;
      const double qi = (1.0 - pc) > focal_eps ? (1.0 - pc) : focal_eps;
      const double tb = ti > 0.5 ? 1.0 : 0.0;
      fl += pow(pc, gamma) * log(pc) * tb + pow(qi, gamma) * log(qi) * (1.0 - tb);
    }
    dice[b] = (float)(1.0 - 2.0 * num / den);
    focal[b] = (float)(-fl / (double)S);
    if (dp) {
      float* g = dp + (size_t)b * S;
      for (long i = 0; i < S; ++i) {
        const double pi = pp[i], ti = tt[i];
        /* dice: d/dp [1 - 2 num/den] */
        const double dnum = (ti * pi > 0) ? ti : 0.0;
        const double dden = (ti + pi + (double)smooth > dice_eps) ? 1.0 : 0.0;
        const double gd = -2.0 * (dnum * den - num * dden) / (den * den);
        /* focal */
        const double tb = ti > 0.5 ? 1.0 : 0.0;
        double gf = 0.0;
        if (pi > focal_eps) {
          const double pc = pi;
          gf += tb * (gamma * pow(pc, gamma - 1.0) * log(pc) + pow(pc, gamma - 1.0));
          const double q = 1.0 - pc;
          if (q > focal_eps)
            gf += (1.0 - tb) *
                  -(gamma * pow(q, gamma - 1.0) * log(q) + pow(q, gamma - 1.0));
        }
        gf = -gf / (double)S;
        g[i] = (float)(gd * gscale_dice + gf * gscale_focal);
      }
    }
  }
}

/* torch.optim.SGD(momentum, nesterov, weight_decay, dampening=0): one step.
 * first != 0 initialises the momentum buffer with the gradient. */
void oracle_sgd_nesterov(float* p, const float* g, float* buf, size_t n, float lr,
                         float momentum, float wd, int nesterov, int first) {
  for (size_t i = 0; i < n; ++i) {
    float d = g[i] + wd * p[i];
    if (momentum != 0.f) {
      buf[i] = first ? d : momentum * buf[i] + d;
      d = nesterov ? d + momentum * buf[i] : buf[i];
    }
    p[i] = p[i] - lr * d;
  }
}
