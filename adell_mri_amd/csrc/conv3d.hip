// Host launchers + C-ABI for the 3D convolution family (see conv_igemm.h for
// the kernel). Everything here is NDHWC fp32 on raw device pointers.
#include "conv_igemm.h"
#include "conv_igemm_f16.h"

#ifdef ADELL_DEBUG
// phase stamps of the f16x3 implicit-GEMM kernel (conv_igemm_f16.h, tools/igemm_stamps.py)
__device__ unsigned long long* adell_g_stamps = nullptr;
extern "C" int adell_debug_set_stamps(void* device_buffer) {
  ADELL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(adell_g_stamps), &device_buffer, sizeof(void*)));
  return ADELL_OK;
}
#endif

// ---------------------------------------------------------------------------
// Tile selection. BM voxels are laid out as a TX x TY x TZ brick (powers of 2).
// ---------------------------------------------------------------------------
struct ConvTile {
  int cfg;  // 0: 256x64, 1: 256x32, 2: 64x64, 3: 128x32
  int BM, BN;
  int lTX, lTY, lTZ;
};

static void adell_shape_brick(int BM, int Wo, int Ho, int Do, int* lx, int* ly,
                              int* lz) {
  // Spread log2(BM) bits over x,y,z (x first) without exceeding the padded
  // extent of each dim; leftover bits go to x.
  int bits = adell_ilog2(BM);
  int cap[3] = {adell_ilog2(Wo), adell_ilog2(Ho), adell_ilog2(Do)};
  int l[3] = {0, 0, 0};
  int pref[3] = {3, 3, 2};  // 8 x 8 x 4
  for (int pass = 0; pass < 2 && bits > 0; ++pass) {
    bool progress = true;
    while (bits > 0 && progress) {
      progress = false;
      for (int d = 0; d < 3 && bits > 0; ++d) {
        const int lim = pass == 0 ? (cap[d] < pref[d] ? cap[d] : pref[d]) : cap[d];
        if (l[d] < lim) {
          ++l[d];
          --bits;
          progress = true;
        }
      }
    }
  }
  l[0] += bits;
  *lx = l[0];
  *ly = l[1];
  *lz = l[2];
}

static ConvTile adell_pick_tile(int N, int Do, int Ho, int Wo, int Cout,
                                int force_cfg) {
  ConvTile t;
  const long vox = (long)N * Do * Ho * Wo;
  const bool wide = Cout > 32;
  const bool big = vox * (wide ? adell_cdiv(Cout, 64) : 1) >= 256L * 256;
  if (force_cfg >= 0)
    t.cfg = force_cfg;
  else if (big)
    t.cfg = wide ? 0 : 1;
  else
    t.cfg = wide ? 2 : 3;
  switch (t.cfg) {
    case 0: t.BM = 256; t.BN = 64; break;
    case 1: t.BM = 256; t.BN = 32; break;
    case 2: t.BM = 64; t.BN = 64; break;
    case 6: t.BM = 64; t.BN = 32; break;   // two-wave blocks (adell_plan_f16: strided layers)
    default: t.BM = 128; t.BN = 32; break;
  }
  adell_shape_brick(t.BM, Wo, Ho, Do, &t.lTX, &t.lTY, &t.lTZ);
  return t;
}

static int g_conv_force_cfg = -1;
extern "C" void adell_debug_force_conv_cfg(int cfg) {
  if (g_conv_force_cfg != cfg) ++g_adell_plan_epoch;
  g_conv_force_cfg = cfg;
}

template <int MT, int NT, int WM, int WN>
static int adell_launch_conv(const ConvArgs& a, dim3 grid, size_t lds,
                             hipStream_t st) {
  static bool attr_done = false;
  auto kern = adell_conv_igemm_kernel<MT, NT, WM, WN>;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kern),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// Generic launcher. `a` must have everything but the tile fields filled in.
static int adell_conv_dispatch(ConvArgs a, int N, hipStream_t st) {
  const ConvTile t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout, g_conv_force_cfg);
  a.lTX = t.lTX;
  a.lTY = t.lTY;
  a.lTZ = t.lTZ;
  const int TX = 1 << t.lTX, TY = 1 << t.lTY, TZ = 1 << t.lTZ;
  a.ntx = adell_cdiv(a.Wo, TX);
  a.nty = adell_cdiv(a.Ho, TY);
  a.ntz = adell_cdiv(a.Do, TZ);
  a.HX = (TX - 1) * a.SW + a.KW;
  a.HY = (TY - 1) * a.SH + a.KH;
  a.HZ = (TZ - 1) * a.SD + a.KD;
  a.VP = a.HX * a.HY * a.HZ;
  if ((a.VP & 1) == 0) a.VP += 1;
  const int ntap = a.KD * a.KH * a.KW;
  size_t lds = ((size_t)8 * a.VP + (size_t)ntap * 8 * t.BN) * sizeof(float);
  const size_t red = (size_t)4 * t.BN * 2 * sizeof(float);
  if (lds < red) lds = red;
  if (lds > 160 * 1024) {
    adell_set_error("conv: LDS need %zu B exceeds 160 KiB (halo %dx%dx%d)", lds,
                    a.HX, a.HY, a.HZ);
    return ADELL_E_UNSUPPORTED;
  }
  a.vecx = (a.C0 % 4 == 0) && (a.C1 % 4 == 0) &&
           (((uintptr_t)a.x0 & 15) == 0) && (((uintptr_t)a.x1 & 15) == 0);
  a.vecw = (a.Cout % 4 == 0) && (((uintptr_t)a.w & 15) == 0);
  const long nsp = (long)a.ntx * a.nty * a.ntz;
  if (nsp > 0x7fffffffL || N > 65535) {
    adell_set_error("conv: grid too large");
    return ADELL_E_UNSUPPORTED;
  }
  ADELL_REQUIRE_ROWS(a.part, a.part_rows, nsp, "conv (fp32 MFMA)");
  dim3 grid((unsigned)nsp, (unsigned)adell_cdiv(a.Cout, t.BN), (unsigned)N);
  switch (t.cfg) {
    case 0: return adell_launch_conv<2, 2, 4, 1>(a, grid, lds, st);
    case 1: return adell_launch_conv<2, 1, 4, 1>(a, grid, lds, st);
    case 2: return adell_launch_conv<1, 1, 2, 2>(a, grid, lds, st);
    default: return adell_launch_conv<1, 1, 4, 1>(a, grid, lds, st);
  }
}

static int adell_check_desc(const adell_conv3d_desc* d) {
  ADELL_REQUIRE(d != nullptr, "conv: null descriptor");
  ADELL_REQUIRE(d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0, "conv: bad input dims");
  ADELL_REQUIRE(d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv: bad channel counts");
  ADELL_REQUIRE(d->KD >= 1 && d->KD <= 7 && d->KH >= 1 && d->KH <= 7 && d->KW >= 1 &&
                    d->KW <= 7,
                "conv: kernel size must be 1..7 per dim");
  ADELL_REQUIRE(d->SD >= 1 && d->SD <= 4 && d->SH >= 1 && d->SH <= 4 && d->SW >= 1 &&
                    d->SW <= 4,
                "conv: stride must be 1..4 per dim");
  ADELL_REQUIRE(d->PD >= 0 && d->PH >= 0 && d->PW >= 0, "conv: negative padding");
  ADELL_REQUIRE(d->Do == (d->D + 2 * d->PD - d->KD) / d->SD + 1 &&
                    d->Ho == (d->H + 2 * d->PH - d->KH) / d->SH + 1 &&
                    d->Wo == (d->W + 2 * d->PW - d->KW) / d->SW + 1,
                "conv: output dims do not match input/kernel/stride/pad");
  ADELL_REQUIRE(d->Do > 0 && d->Ho > 0 && d->Wo > 0, "conv: empty output");
  return ADELL_OK;
}

extern "C" int adell_conv3d_fwd_ntiles(const adell_conv3d_desc* d) {
  if (adell_check_desc(d) != ADELL_OK) return ADELL_E_BADARG;
  const ConvTile t = adell_pick_tile(d->N, d->Do, d->Ho, d->Wo, d->Cout, g_conv_force_cfg);
  return adell_cdiv(d->Wo, 1 << t.lTX) * adell_cdiv(d->Ho, 1 << t.lTY) *
         adell_cdiv(d->Do, 1 << t.lTZ);
}

static int adell_fill_fwd(ConvArgs& a, const adell_conv3d_desc* d, const float* x0,
                          const float* x1, const float* bias, const float* residual, float* y,
                          float* stat_partials, int partial_rows = 0) {
  int rc = adell_check_desc(d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x0 && y, "conv_fwd: null pointer");
  ADELL_REQUIRE(d->C1 == 0 || x1, "conv_fwd: C1 > 0 needs x1");
  a = ConvArgs{};
  a.x0 = x0; a.x1 = x1; a.w = nullptr; a.bias = bias; a.res = residual;
  a.y0 = y; a.y1 = nullptr; a.part = stat_partials; a.part_rows = partial_rows;
  a.D = d->D; a.H = d->H; a.W = d->W;
  a.C0 = d->C0; a.C1 = d->C1; a.Cin = d->C0 + d->C1; a.Cout = d->Cout;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW;
  a.SD = d->SD; a.SH = d->SH; a.SW = d->SW;
  a.PD = d->PD; a.PH = d->PH; a.PW = d->PW;
  a.UPS = a.UPSY = a.UPSZ = 1;
  a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ysplit = d->Cout; a.shuffle = 0; a.Cs = d->Cout;
  return ADELL_OK;
}

extern "C" int adell_conv3d_fwd(const adell_conv3d_desc* d, const float* x0,
                                const float* x1, const float* w_packed,
                                const float* bias, const float* residual,
                                float* y, float* stat_partials, int partial_rows, void* stream) {
  ConvArgs a;
  int rc = adell_fill_fwd(a, d, x0, x1, bias, residual, y, stat_partials, partial_rows);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_packed, "conv_fwd: null weights");
  a.w = w_packed;
  return adell_conv_dispatch(a, d->N, (hipStream_t)stream);
}

// dX = conv_stride1(zero_insert(dY, S), flip(W)^T, pad = K-1-P), written to the
// two sources of the forward's virtual concat (per-axis insertion factors, so anisotropic
// strides such as the (1,2,2) of a 2-D network work); w_packed_bwd is
// [flipped tap][Cout][Cin] (adell_pack_weight mode 1).
static int adell_fill_bwd_data(ConvArgs& a, const adell_conv3d_desc* d, const float* dy,
                               float* dx0, float* dx1) {
  int rc = adell_check_desc(d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(dy && dx0, "conv_bwd_data: null pointer");
  ADELL_REQUIRE(d->C1 == 0 || dx1, "conv_bwd_data: C1 > 0 needs dx1");
  // (pad > k - 1 -- the padded 1x1 convs of the depthwise U-Net blocks, unet.py:292-307 -- makes
  // the padding of this stride-1 conv over dY negative: the halo origin moves INSIDE dY, which the
  // generic halo arithmetic handles as it stands)
  a = ConvArgs{};
  a.x0 = dy; a.x1 = nullptr; a.w = nullptr; a.bias = nullptr; a.res = nullptr;
  a.y0 = dx0; a.y1 = dx1; a.part = nullptr;
  a.D = d->Do; a.H = d->Ho; a.W = d->Wo;
  a.C0 = d->Cout; a.C1 = 0; a.Cin = d->Cout; a.Cout = d->C0 + d->C1;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW;
  a.SD = a.SH = a.SW = 1;
  a.PD = d->KD - 1 - d->PD; a.PH = d->KH - 1 - d->PH; a.PW = d->KW - 1 - d->PW;
  a.UPS = d->SW; a.UPSY = d->SH; a.UPSZ = d->SD;   // per-axis zero insertion
  a.Do = d->D; a.Ho = d->H; a.Wo = d->W;
  a.ysplit = d->C0; a.shuffle = 0; a.Cs = a.Cout;
  return ADELL_OK;
}

extern "C" int adell_conv3d_bwd_data(const adell_conv3d_desc* d, const float* dy,
                                     const float* w_packed_bwd, float* dx0,
                                     float* dx1, void* stream) {
  ConvArgs a;
  int rc = adell_fill_bwd_data(a, d, dy, dx0, dx1);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_packed_bwd, "conv_bwd_data: null weights");
  a.w = w_packed_bwd;
  return adell_conv_dispatch(a, d->N, (hipStream_t)stream);
}

// ConvTranspose3d with kernel = stride = (FD,FH,FW), each 1 or 2, padding 0: a
// per-voxel GEMM [Cin] -> [F*Cout] whose columns scatter to the FD x FH x FW children
// of the voxel. w_packed is [Cin][F][Cout] (adell_pack_weight mode 2).
static int adell_convt_factors_ok(int FD, int FH, int FW) {
  return (FD == 1 || FD == 2) && (FH == 1 || FH == 2) && (FW == 1 || FW == 2);
}

extern "C" int adell_convtranspose3d_fwd(int N, int D, int H, int W, int Cin, int Cout, int FD,
                                         int FH, int FW, const float* x, const float* w_packed,
                                         const float* bias, float* y, void* stream) {
  ADELL_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convT_fwd: bad dims");
  ADELL_REQUIRE(adell_convt_factors_ok(FD, FH, FW), "convT_fwd: kernel=stride must be 1 or 2 per dim");
  ADELL_REQUIRE(x && w_packed && y, "convT_fwd: null pointer");
  ConvArgs a = {};
  a.x0 = x; a.x1 = nullptr; a.w = w_packed; a.bias = bias; a.res = nullptr;
  a.y0 = y; a.y1 = nullptr; a.part = nullptr;
  a.D = D; a.H = H; a.W = W;
  a.C0 = Cin; a.C1 = 0; a.Cin = Cin; a.Cout = FD * FH * FW * Cout;
  a.KD = a.KH = a.KW = 1;
  a.SD = a.SH = a.SW = 1;
  a.PD = a.PH = a.PW = 0;
  a.UPS = a.UPSY = a.UPSZ = 1;
  a.Do = D; a.Ho = H; a.Wo = W;
  a.ysplit = a.Cout; a.Cs = Cout;
  a.shuffle = 8 | (FW - 1) | ((FH - 1) << 1) | ((FD - 1) << 2);  // bit 3 marks "scatter store"
  return adell_conv_dispatch(a, N, (hipStream_t)stream);
}

extern "C" int adell_convtranspose3d_k2s2_fwd(int N, int D, int H, int W, int Cin,
                                              int Cout, const float* x,
                                              const float* w_packed,
                                              const float* bias, float* y,
                                              void* stream) {
  return adell_convtranspose3d_fwd(N, D, H, W, Cin, Cout, 2, 2, 2, x, w_packed, bias, y, stream);
}

// dX of the transposed conv = a kernel = stride = (FD,FH,FW), padding 0 convolution of dY
// ([N,FD*D,FH*H,FW*W,Cout]) with w_packed_bwd [taps][Cout][Cin] (pack mode 3).
extern "C" int adell_convtranspose3d_bwd_data(int N, int D, int H, int W, int Cin, int Cout,
                                              int FD, int FH, int FW, const float* dy,
                                              const float* w_packed_bwd, float* dx,
                                              void* stream) {
  ADELL_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0,
                "convT_bwd_data: bad dims");
  ADELL_REQUIRE(adell_convt_factors_ok(FD, FH, FW), "convT_bwd_data: bad factors");
  ADELL_REQUIRE(dy && w_packed_bwd && dx, "convT_bwd_data: null pointer");
  ConvArgs a = {};
  a.x0 = dy; a.x1 = nullptr; a.w = w_packed_bwd; a.bias = nullptr; a.res = nullptr;
  a.y0 = dx; a.y1 = nullptr; a.part = nullptr;
  a.D = FD * D; a.H = FH * H; a.W = FW * W;
  a.C0 = Cout; a.C1 = 0; a.Cin = Cout; a.Cout = Cin;
  a.KD = a.SD = FD; a.KH = a.SH = FH; a.KW = a.SW = FW;
  a.PD = a.PH = a.PW = 0;
  a.UPS = a.UPSY = a.UPSZ = 1;
  a.Do = D; a.Ho = H; a.Wo = W;
  a.ysplit = a.Cout; a.shuffle = 0; a.Cs = a.Cout;
  return adell_conv_dispatch(a, N, (hipStream_t)stream);
}

extern "C" int adell_convtranspose3d_k2s2_bwd_data(int N, int D, int H, int W,
                                                   int Cin, int Cout,
                                                   const float* dy,
                                                   const float* w_packed_bwd,
                                                   float* dx, void* stream) {
  return adell_convtranspose3d_bwd_data(N, D, H, W, Cin, Cout, 2, 2, 2, dy, w_packed_bwd, dx,
                                        stream);
}

// ---------------------------------------------------------------------------
// Weight repacking (canonical torch layouts -> GEMM-B layouts), run once per
// optimiser step on ~8M floats: HBM-trivial.
// ---------------------------------------------------------------------------
__global__ void adell_pack_weight_kernel(const float* __restrict__ w,
                                         float* __restrict__ out, int mode, int A,
                                         int B, int KD, int KH, int KW) {
  // conv:  w[A=Cout][B=Cin][taps];  convT: w[A=Cin][B=Cout][taps]
  const int taps = KD * KH * KW;
  const long total = (long)A * B * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    long o = i;  // index into `out`
    int tap, ia, ib;
    long src;
    switch (mode) {
      case 0:  // out[tap][Cin][Cout]
        ia = o % A; o /= A; ib = o % B; tap = o / B;
        src = ((long)ia * B + ib) * taps + tap;
        break;
      case 1:  // out[flipped tap][Cout][Cin]
        ib = o % B; o /= B; ia = o % A; tap = o / A;
        src = ((long)ia * B + ib) * taps + (taps - 1 - tap);
        break;
      case 2:  // convT fwd: out[Cin][tap][Cout]
        ib = o % B; o /= B; tap = o % taps; ia = o / taps;
        src = ((long)ia * B + ib) * taps + tap;
        break;
      default:  // 3, convT bwd-data: out[tap][Cout][Cin]
        ia = o % A; o /= A; ib = o % B; tap = o / B;
        src = ((long)ia * B + ib) * taps + tap;
        break;
    }
    out[i] = w[src];
  }
}

extern "C" int adell_pack_weight(const float* w, float* out, int mode, int dim0,
                                 int dim1, int KD, int KH, int KW, void* stream) {
  ADELL_REQUIRE(w && out, "pack_weight: null pointer");
  ADELL_REQUIRE(mode >= 0 && mode <= 3, "pack_weight: mode must be 0..3");
  ADELL_REQUIRE(dim0 > 0 && dim1 > 0 && KD > 0 && KH > 0 && KW > 0, "pack_weight: bad dims");
  const long total = (long)dim0 * dim1 * KD * KH * KW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_pack_weight_kernel, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, w, out, mode, dim0, dim1, KD, KH, KW);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// f16x3 path (conv_igemm_f16.h): fp32 tensors in and out, 3 f16 MFMAs per K-block.
// ---------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int SPEC, int EPI = 0, int ROWS = 0>
static int adell_launch_conv_f16(const ConvArgs& a, const ConvF16Extra& e, dim3 grid, size_t lds,
                                 hipStream_t st) {
  static bool attr_done = false;
  auto kern = adell_conv_igemm_f16_kernel<MT, NT, WM, WN, SPEC, EPI, ROWS>;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, a, e);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

static int adell_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// 16 -> 16 channel layers: the z-marching 16-column kernel (conv_zring16.hip)
extern "C" int adell_conv_zring16_ok(const ConvArgs* a);
extern "C" void adell_conv_zring16_segments(int N, int Do, int Ho, int Wo, int* seglen, int* nseg);
extern "C" int adell_conv_zring16_launch(const ConvArgs* a, const ConvF16Extra* e, int N, int seglen,
                                         int nseg, hipStream_t st);

// Tile plan of the f16x3 kernel: the heuristic brick, or (when its halo does not fit
// LDS, i.e. stride 2) the small-brick configuration of the same channel width.
static int adell_plan_f16(ConvArgs& a, int N, ConvTile* tile, size_t* lds_out) {
  ConvTile t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout, g_conv_force_cfg);
  if (g_conv_force_cfg < 0 && adell_conv_zring16_ok(&a)) {
    // cfg 8: units = 8 x 8 columns x z segments (ntz = segments, HZ = steps per segment); the
    // statistics rows per item are ntx * nty * ntz like every other plan's
    int seglen = 0, nseg = 0;
    adell_conv_zring16_segments(N, a.Do, a.Ho, a.Wo, &seglen, &nseg);
    t.cfg = 8;
    t.BM = 64;
    t.BN = 16;
    t.lTX = 3; t.lTY = 3; t.lTZ = 0;
    a.lTX = 3; a.lTY = 3; a.lTZ = 0;
    a.ntx = adell_cdiv(a.Wo, 8);
    a.nty = adell_cdiv(a.Ho, 8);
    a.ntz = nseg;
    a.HX = 10; a.HY = 10; a.HZ = seglen;
    a.VP = 100;
    a.GKH = 3;
    *tile = t;
    *lds_out = 0;
    return ADELL_OK;
  }
  // kernel == stride > 1 (transposed-conv backward-data): every staged voxel feeds one tap, so
  // the launch is bound by staging, not MFMA. 64-voxel bricks (32 KB of LDS, 4 blocks per CU)
  // overlap the staging of one block with the MFMAs of another (measured, 2 x 64^3 x 32 ->
  // 32: 0.47 -> 0.31 ms against the 256-voxel brick at one block per CU)
  if (g_conv_force_cfg < 0 && a.shuffle == 0 && a.KD == a.SD && a.KH == a.SH && a.KW == a.SW &&
      a.KD * a.KH * a.KW > 1 && a.UPS == 1 && a.UPSY == 1 && a.UPSZ == 1)
    t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout,
                        a.Cout <= 32 ? 6 : 2);
  // strided k > stride layers with <= 32 output channels (the 3^3 stride-2 downsampling convs):
  // the halo of a 128-voxel brick is 17 x 9 x 9 voxels = 88 KB, one block per CU, and nothing
  // overlaps its staging (measured 2 x 128^3 -> 64^3, 32 -> 32: 0.73 ms at 40 TF). 64-voxel bricks
  // on two-wave blocks (46 KB halo) keep two to three blocks per CU in different phases.
  else if (g_conv_force_cfg < 0 && a.shuffle == 0 && a.Cout <= 32 &&
           (a.SD > 1 || a.SH > 1 || a.SW > 1) && a.UPS == 1 && a.UPSY == 1 && a.UPSZ == 1)
    t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout, 6);
  // 5x5x5 stride-1 layers (the k = 5 bottleneck blocks of the ResNet backbone, 64 -> 64 at 65^3): a
  // 256-voxel brick has a 12 x 12 x 8 halo (74 KB) and one block per CU; 128-voxel x 32-column bricks
  // (49 KB halo, two blocks per CU): 1923 -> 1687 us (tools/cfg_exp.py 64 64 65 1 5)
  else if (g_conv_force_cfg < 0 && a.shuffle == 0 && a.KD == 5 && a.KH == 5 && a.KW == 5 &&
           a.SD == 1 && a.SH == 1 && a.SW == 1 && a.UPS == 1 && a.UPSY == 1 && a.UPSZ == 1 &&
           (long)N * a.Do * a.Ho * a.Wo >= 65536)
    t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout, 3);
  // Low-resolution 3x3x3 stride-1 levels with wide outputs (measured with warm clocks,
  // tools/cfg_exp.py, batch 2): the 64-voxel x 64-column bricks of the "small problem" rule make
  // every block stream the whole weight slice of its column tile from L2 (256 -> 256 at 16^3:
  // 0.9 GB of weight reads, 214 us); 256-voxel x 32-column bricks + split-K read a quarter of
  // that (139 us). 8^3 levels: 64-voxel x 32-column two-wave bricks (256 -> 256: 89 -> 68 us).
  bool retiled = false;
  if (g_conv_force_cfg < 0 && a.shuffle == 0 && a.KD == 3 &&
      a.KH == 3 && a.KW == 3 && a.SD == 1 && a.SH == 1 && a.SW == 1 && a.UPS == 1 &&
      a.UPSY == 1 && a.UPSZ == 1) {
    const long vox = (long)N * a.Do * a.Ho * a.Wo;
    int pick = -1;
    if (a.Cout <= 32) {
      // 16^3 / 32^3 levels with <= 32 columns: 8x8x4 bricks (the 8x8x8 bricks of the large
      // layers leave half the chip without a block: 32 -> 32 at 2 x 32^3 30 -> 20 us)
      if (vox >= 4096 && vox < 262144) pick = 1;
    } else if (vox < 4096)
      // (512 -> 512 at 9 x 9 x 33, the bottom level of the ResNet-backbone U-Net: every 64-voxel
      // brick streams 1.8 MB of weights per 32 columns -- 851 us; 256-voxel x 64-column bricks 279 us)
      // (round 5, tools/cfg_exp.py, batch 2 at 8^3: 64-voxel x 64-column four-wave bricks 41 us against
      // 50 us on the two-wave ones for 256 -> 256, 25 / 31 for 128 -> 256, 72 / 86 for 512 -> 256)
      pick = (a.Cin >= 512 && vox >= 2048) ? 0 : (vox >= 512 ? 2 : 6);
    else if (vox < 32768)
      // (512 output channels: 64-column bricks -- 512 -> 512 at 17 x 17 x 65 1381 -> 1283 us,
      // 256 -> 512 693 -> 657 us; 256 outputs stay on 32 columns: 765 vs 778 us. Round 5: the
      // 128-voxel bricks that Cin < 256 took lose to the 256-voxel ones + split-K at 2 x 16^3:
      // 128 -> 128 55 -> 38 us, 64 -> 128 35 -> 29 us, 128 -> 64 35 -> 28 us)
      pick = a.Cout >= 512 ? 0 : 1;
    else if (vox < 262144 && a.Cout <= 64)
      pick = 1;
    if (pick >= 0) {
      t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout, pick);
      retiled = true;
    }
  }
  size_t lds = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    a.lTX = t.lTX; a.lTY = t.lTY; a.lTZ = t.lTZ;
    const int TX = 1 << t.lTX, TY = 1 << t.lTY, TZ = 1 << t.lTZ;
    a.ntx = adell_cdiv(a.Wo, TX);
    a.nty = adell_cdiv(a.Ho, TY);
    a.ntz = adell_cdiv(a.Do, TZ);
    a.HX = (TX - 1) * a.SW + a.KW;
    a.HY = (TY - 1) * a.SH + a.KH;
    a.HZ = (TZ - 1) * a.SD + a.KD;
    a.VP = a.HX * a.HY * a.HZ;
    // weights are staged one kz plane at a time, or row by row when a plane is too big
    a.GKH = ((size_t)a.KH * a.KW * t.BN * 64 <= 40 * 1024) ? a.KH : 1;
    lds = (size_t)a.VP * 64 + (size_t)a.GKH * a.KW * t.BN * 64 + 64;
    const size_t red = (size_t)8 * t.BN * 2 * sizeof(float);
    if (lds < red) lds = red;
    if (lds <= 160 * 1024) break;
    const int next = t.BN == 64 ? 2 : (t.cfg == 3 ? 6 : 3);
    if (t.cfg == next || g_conv_force_cfg >= 0) break;
    t = adell_pick_tile(N, a.Do, a.Ho, a.Wo, a.Cout, next);
  }
  if (lds > 160 * 1024) {
    adell_set_error("conv f16x3: LDS need %zu B exceeds 160 KiB (k=%d stride=%d)", lds, a.KW, a.SW);
    return ADELL_E_UNSUPPORTED;
  }
  // 32-channel tiles of 3x3x3 stride-1 layers: 8x8x8 bricks (cfg 4, SPEC = 3 instance of the
  // kernel): a third less halo per output and half the weight staging of the 8x8x4 brick
  if (t.cfg == 1 && !retiled && a.KD == 3 && a.KH == 3 && a.KW == 3 && a.SD == 1 && a.SH == 1 && a.SW == 1 &&
      a.UPS == 1 && a.UPSY == 1 && a.UPSZ == 1 && a.shuffle == 0 && a.lTX == 3 && a.lTY == 3 &&
      a.lTZ == 2 && a.Do >= 8 && a.C0 % 16 == 0 && a.C1 % 16 == 0 &&
      (size_t)a.D * a.H * a.W * (a.C0 > a.C1 ? a.C0 : a.C1) < ((size_t)1 << 30) &&
      g_conv_force_cfg < 0 && !g_adell_tune.igemm_nospec &&
      !g_adell_tune.igemm_no8) {
    t.cfg = 4;
    t.BM = 512;
    t.lTZ = 3;
    a.lTZ = 3;
    a.ntz = adell_cdiv(a.Do, 8);
    a.HZ = 10;
    a.VP = a.HX * a.HY * a.HZ;
    lds = (size_t)1000 * 64 + (size_t)7 * 32 * 64 + 64;
  }
  // transposed-conv forward (1 tap, F * Cs columns with a pixel-shuffle store): one block takes a
  // 64-voxel brick and up to 256 columns (cfg 5), so the input brick is staged once instead of
  // once per 32 / 64-column tile
  if (a.shuffle != 0 && a.KD == 1 && a.KH == 1 && a.KW == 1 && a.Cout >= 128 &&
      g_conv_force_cfg < 0 && !g_adell_tune.igemm_nospec) {
    t.cfg = 5;
    t.BM = 64;
    t.BN = 256;
    adell_shape_brick(64, a.Wo, a.Ho, a.Do, &t.lTX, &t.lTY, &t.lTZ);
    a.lTX = t.lTX; a.lTY = t.lTY; a.lTZ = t.lTZ;
    a.ntx = adell_cdiv(a.Wo, 1 << t.lTX);
    a.nty = adell_cdiv(a.Ho, 1 << t.lTY);
    a.ntz = adell_cdiv(a.Do, 1 << t.lTZ);
    a.HX = 1 << t.lTX; a.HY = 1 << t.lTY; a.HZ = 1 << t.lTZ;
    a.VP = a.HX * a.HY * a.HZ;
    a.GKH = 1;
    lds = (size_t)a.VP * 64 + (size_t)256 * 64 + 64;
    if (lds < (size_t)1 * 256 * 2 * sizeof(float)) lds = (size_t)256 * 2 * sizeof(float);
  }
  *tile = t;
  *lds_out = lds;
  return ADELL_OK;
}

// ---- split-K for the low-resolution layers --------------------------------------------------
// 8^3 - 16^3 levels have 8 - 64 bricks: a block that walks all 8 - 16 channel chunks of its brick
// takes ~80 us while most of the chip idles. The chunks are shared out over `ksplit` blocks per
// brick instead (blockIdx.z); each writes raw partial outputs to its slab, and this kernel folds
// the slabs in fixed order, adds bias and residual, stores y (or the two halves of a split
// store) and emits the per-tile (sum, sum of squares) partials of the normal epilogue.
struct ConvFoldArgs {
  const float* slabs;
  const float* bias;
  const float* res;
  float* y0;
  float* y1;
  float* part;
  long vox;           // output voxels per batch item
  long slab;          // floats per slab
  int ksplit, Cout, ysplit, ntiles;
};

__global__ __launch_bounds__(256) void adell_conv_splitk_fold_kernel(ConvFoldArgs a) {
  __shared__ float4 sred[256][2];
  const int tid = threadIdx.x;
  const int cq = a.Cout >> 2;              // quads per row; 256 % cq == 0 (checked on the host)
  const int q = tid % cq, r0 = tid / cq, rstep = 256 / cq;
  const int nb = blockIdx.y, tile = blockIdx.x;
  const long vpt = (a.vox + a.ntiles - 1) / a.ntiles;
  const long v0 = (long)tile * vpt;
  const long v1 = (v0 + vpt) < a.vox ? (v0 + vpt) : a.vox;
  const int c = 4 * q;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + c);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (long v = v0 + r0; v < v1; v += rstep) {
    const size_t row = (size_t)nb * a.vox + v;
    float4 t = bv;
    for (int k = 0; k < a.ksplit; ++k) {
      const float4 u = *reinterpret_cast<const float4*>(a.slabs + (size_t)k * a.slab + row * a.Cout + c);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    if (a.res) {
      const float4 u = *reinterpret_cast<const float4*>(a.res + row * a.Cout + c);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float* o = c < a.ysplit ? a.y0 + row * a.ysplit + c
                            : a.y1 + row * (a.Cout - a.ysplit) + (c - a.ysplit);
    *reinterpret_cast<float4*>(o) = t;
    s1.x += t.x; s1.y += t.y; s1.z += t.z; s1.w += t.w;
    s2.x += t.x * t.x; s2.y += t.y * t.y; s2.z += t.z * t.z; s2.w += t.w * t.w;
  }
  if (a.part) {
    sred[tid][0] = s1;
    sred[tid][1] = s2;
    __syncthreads();
    if (tid < cq) {
      float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
      for (int r = 0; r < rstep; ++r) {
        const float4 u = sred[r * cq + tid][0], w = sred[r * cq + tid][1];
        t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
        t2.x += w.x; t2.y += w.y; t2.z += w.z; t2.w += w.w;
      }
      float* p = a.part + (((size_t)nb * a.ntiles + tile) * a.Cout + 4 * tid) * 2;
      p[0] = t1.x; p[1] = t2.x; p[2] = t1.y; p[3] = t2.y;
      p[4] = t1.z; p[5] = t2.z; p[6] = t1.w; p[7] = t2.w;
    }
  }
}

// Voxel ranges (= blocks per batch item, = statistics partial rows per item) of the fold over the K
// shares: the conv that needs split-K has few bricks (16 per item at 16^3), and a fold on that many
// blocks streamed its 4 x 4 MB of slabs at a twentieth of the chip's bandwidth (20-53 us per call,
// 24 calls per step) -- the fold gets its own, finer partition: ~512 blocks in all.
static int adell_fold_tiles(long vox, int N, int Cout, long bricks) {
  const int rstep = 256 / (Cout / 4);          // voxel rows a block covers per pass
  long tiles = adell_cdiv(512, N);
  const long most = adell_cdiv(vox, rstep);
  if (tiles > most) tiles = most;
  if (tiles < 1) tiles = 1;
  return (int)tiles;
}

// number of K shares for this problem (1 = no split) given the planned tile
static int adell_splitk_shares(const ConvArgs& a, const ConvTile& t, int N) {
  if (g_adell_tune.no_splitk) return 1;
  const int nchunk = adell_cdiv(a.Cin, 16);
  const long blocks = (long)a.ntx * a.nty * a.ntz * adell_cdiv(a.Cout, t.BN) * N;
  const int cq = a.Cout / 4;
  if (a.shuffle || a.Cout % 4 || cq > 256 || 256 % cq || a.ysplit % 4 || nchunk < 4 || blocks > 256)
    return 1;
  int ks = (int)(512 / blocks);   // measured: a target of 1024 blocks costs more slab traffic than it hides
  if (ks > nchunk) ks = nchunk;
  if (ks < 2) return 1;
  const int cpk = adell_cdiv(nchunk, ks);
  return adell_cdiv(nchunk, cpk);
}

// adn: 0 plain; 1 launch the EPI = 1 instance (fused norm / dropout / activation backward in the
// epilogue: ConvF16Extra::adn), failing when the plan does not allow it; -1 only answer whether it
// would (returns the number of bricks per batch item = rows of the partial-sum buffer, 0 = no).
static int adell_conv_dispatch_f16(ConvArgs a, ConvF16Extra e, int N, hipStream_t st,
                                   void* ws = nullptr, size_t ws_bytes = 0, int adn = 0) {
  ConvTile t;
  size_t lds;
  int rc = adell_plan_f16(a, N, &t, &lds);
  if (rc != ADELL_OK) return adn < 0 ? 0 : rc;
  // (the f16x3 kernel's 16-byte halo loads address a batch item with 32-bit byte offsets)
  a.vecx = (a.C0 % 4 == 0) && (a.C1 % 4 == 0) && (((uintptr_t)a.x0 & 15) == 0) &&
           (((uintptr_t)a.x1 & 15) == 0) &&
           (size_t)a.D * a.H * a.W * (a.C0 > a.C1 ? a.C0 : a.C1) < ((size_t)1 << 30);
  a.vecw = 1;
  e.dbg = g_adell_tune.igemm_dbg;
  const long nsp = (long)a.ntx * a.nty * a.ntz;
  if (nsp > 0x0fffffffL || N > 65535) {
    adell_set_error("conv: grid too large");
    return ADELL_E_UNSUPPORTED;
  }
  {
    // the epilogue addresses rows inside a brick with 32-bit element offsets
    const int fx = (a.shuffle & 1) + 1, fy = ((a.shuffle >> 1) & 1) + 1,
              fz = ((a.shuffle >> 2) & 1) + 1;
    const size_t span = ((size_t)fz << a.lTZ) * ((size_t)fy * a.Ho) * ((size_t)fx * a.Wo) *
                        (size_t)(a.shuffle ? a.Cs : a.Cout);
    if (span >= ((size_t)1 << 32)) {
      adell_set_error("conv f16x3: a brick of %d output planes spans %zu elements (>= 2^32)",
                      1 << a.lTZ, span);
      return ADELL_E_UNSUPPORTED;
    }
  }
  // split-K when the caller supplied a workspace and the problem is small (see the fold kernel)
  ConvArgs full = a;
  int shares = ws ? adell_splitk_shares(a, t, N) : 1;
  const long slab = (long)N * a.Do * a.Ho * a.Wo * a.Cout;
  if (shares > 1 && (size_t)shares * slab * sizeof(float) > ws_bytes) {
    // (the statistics rows of the split form are the fold's, adell_conv3d_fwd_ntiles_f16x3_ws:
    // falling back to one share here would write a different number of rows)
    adell_set_error("conv f16x3: split-K workspace too small (%zu bytes, adell_conv3d_splitk_workspace)",
                    ws_bytes);
    return ADELL_E_BADARG;
  }
  a.ksplit = shares;
  a.slab = slab;
  if (shares > 1) a.y0 = (float*)ws;
  // blocks are dealt to the 8 XCDs in contiguous ranges (see the kernel): pad the grid
  dim3 grid((unsigned)(8 * ((nsp + 7) / 8)), (unsigned)adell_cdiv(a.Cout, t.BN),
            (unsigned)(N * shares));
  const bool spec = a.KD == 3 && a.KH == 3 && a.KW == 3 && a.SD == 1 && a.SH == 1 && a.SW == 1 &&
                    a.UPS == 1 && a.UPSY == 1 && a.UPSZ == 1 && a.lTX == 3 && a.lTY == 3 &&
                    a.lTZ == 2 && a.shuffle == 0 && a.vecx && a.GKH == 3 && t.cfg <= 1 &&
                    a.C0 % 16 == 0 && a.C1 % 16 == 0 &&
                    (size_t)a.D * a.H * a.W * (a.C0 > a.C1 ? a.C0 : a.C1) < ((size_t)1 << 30) &&
                    !g_adell_tune.igemm_nospec;
  // split-row sources are staged by the specialised instances only (no split-K: a share would have
  // to know the rows' exponent of chunks it does not own... it could; it is simply not built)
  const bool rows_ok = shares == 1 && ((t.cfg <= 1 && spec) || t.cfg == 4 || t.cfg == 8);
  if (adn == -2) return rows_ok ? 1 : 0;
  if ((e.xs0 != nullptr || e.xs1 != nullptr) && !rows_ok) {
    adell_set_error("conv f16x3: this problem's launch plan does not take split-row sources "
                    "(adell_conv3d_f16x3_rows_ok)");
    return ADELL_E_UNSUPPORTED;
  }
  if (adn >= 0)
    ADELL_REQUIRE_ROWS(a.part, a.part_rows,
                       shares > 1 ? adell_fold_tiles((long)a.Do * a.Ho * a.Wo, N, a.Cout, nsp) : nsp,
                       "conv f16x3");
  if (adn != 0) {
    // the fused epilogue lives in the interior-brick path of the specialised instances: every brick
    // whole, every 32-column sub-tile whole and on one side of ysplit, no split-K, no
    // wave-specialised form
    const int bz = t.cfg == 4 ? 8 : 4;
    const bool ok = ((t.cfg <= 1 && spec) || t.cfg == 4) && shares == 1 && (a.shuffle & 16) == 0 &&
                    a.Wo % 8 == 0 && a.Ho % 8 == 0 && a.Do % bz == 0 && a.Cout % t.BN == 0 &&
                    a.ysplit % 32 == 0 && a.bias == nullptr && nsp * (long)a.Cout < (1L << 30);
    if (adn < 0) return ok ? (int)nsp : 0;
    if (!ok) {
      adell_set_error("conv_bwd_data_f16x3_adn: this problem does not take the fused epilogue");
      return ADELL_E_UNSUPPORTED;
    }
    switch (t.cfg) {
      case 0: return adell_launch_conv_f16<2, 2, 4, 1, 1, 1>(a, e, grid, lds, st);
      case 4: return adell_launch_conv_f16<4, 1, 4, 1, 3, 1>(a, e, grid, lds, st);
      default: return adell_launch_conv_f16<2, 1, 4, 1, 1, 1>(a, e, grid, lds, st);
    }
  }
  if (t.cfg == 8) return adell_conv_zring16_launch(&a, &e, N, a.HZ, a.ntz, st);
  int rc2 = ADELL_OK;
  if (e.xs0 != nullptr || e.xs1 != nullptr) {
    // split-row sources (rows_ok above): the instance without the fp32 staging path when every
    // source is rows, else the one that decides per source
    const bool all_rows = e.xs0 != nullptr && (a.C1 == 0 || e.xs1 != nullptr);
    switch (t.cfg) {
      case 0:
        return all_rows ? adell_launch_conv_f16<2, 2, 4, 1, 1, 0, 2>(a, e, grid, lds, st)
                        : adell_launch_conv_f16<2, 2, 4, 1, 1, 0, 1>(a, e, grid, lds, st);
      case 4:
        return all_rows ? adell_launch_conv_f16<4, 1, 4, 1, 3, 0, 2>(a, e, grid, lds, st)
                        : adell_launch_conv_f16<4, 1, 4, 1, 3, 0, 1>(a, e, grid, lds, st);
      default:
        return all_rows ? adell_launch_conv_f16<2, 1, 4, 1, 1, 0, 2>(a, e, grid, lds, st)
                        : adell_launch_conv_f16<2, 1, 4, 1, 1, 0, 1>(a, e, grid, lds, st);
    }
  }
  switch (t.cfg) {
    case 0:
      rc2 = spec ? adell_launch_conv_f16<2, 2, 4, 1, 1>(a, e, grid, lds, st)
                  : adell_launch_conv_f16<2, 2, 4, 1, 0>(a, e, grid, lds, st);
      break;
    case 5:   // transposed-conv forward: 64 voxels x 256 columns per block
      rc2 = adell_launch_conv_f16<2, 2, 1, 4, 0>(a, e, grid, lds, st);
      break;
    case 4:   // 8x8x8 bricks, four m-tiles per wave (adell_plan_f16)
      if (!a.vecx) {
        adell_set_error("conv f16x3: input pointers must be 16-byte aligned");
        return ADELL_E_BADARG;
      }
      rc2 = adell_launch_conv_f16<4, 1, 4, 1, 3>(a, e, grid, lds, st);
      break;
    case 1:
      rc2 = spec ? adell_launch_conv_f16<2, 1, 4, 1, 1>(a, e, grid, lds, st)
                  : adell_launch_conv_f16<2, 1, 4, 1, 0>(a, e, grid, lds, st);
      break;
    case 2: rc2 = adell_launch_conv_f16<1, 1, 2, 2, 0>(a, e, grid, lds, st); break;
    case 6: rc2 = adell_launch_conv_f16<1, 1, 2, 1, 0>(a, e, grid, lds, st); break;
    default: rc2 = adell_launch_conv_f16<1, 1, 4, 1, 0>(a, e, grid, lds, st); break;
  }
  if (rc2 != ADELL_OK || shares == 1) return rc2;
  ConvFoldArgs f = {};
  f.slabs = (const float*)ws; f.bias = full.bias; f.res = full.res; f.y0 = full.y0; f.y1 = full.y1;
  f.part = full.part; f.vox = (long)a.Do * a.Ho * a.Wo; f.slab = slab; f.ksplit = shares;
  f.Cout = a.Cout; f.ysplit = full.ysplit; f.ntiles = adell_fold_tiles(f.vox, N, a.Cout, nsp);
  hipLaunchKernelGGL(adell_conv_splitk_fold_kernel, dim3((unsigned)f.ntiles, (unsigned)N), dim3(256),
                     0, st, f);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_conv3d_fwd_ntiles_f16x3(const adell_conv3d_desc* d) {
  ConvArgs a;
  float dummy;
  if (adell_fill_fwd(a, d, &dummy, &dummy, nullptr, nullptr, &dummy, nullptr) != ADELL_OK)
    return ADELL_E_BADARG;
  ConvTile t;
  size_t lds;
  if (adell_plan_f16(a, d->N, &t, &lds) != ADELL_OK) return ADELL_E_UNSUPPORTED;
  return a.ntx * a.nty * a.ntz;
}

// Statistics partial rows per batch item that adell_conv3d_fwd_f16x3_ws writes (called with the
// workspace adell_conv3d_splitk_workspace asks for): the fold's voxel ranges when the layer runs
// split-K, else the bricks.
extern "C" int adell_conv3d_fwd_ntiles_f16x3_ws(const adell_conv3d_desc* d) {
  ConvArgs a;
  float dummy;
  if (adell_fill_fwd(a, d, &dummy, &dummy, nullptr, nullptr, &dummy, nullptr) != ADELL_OK)
    return ADELL_E_BADARG;
  ConvTile t;
  size_t lds;
  if (adell_plan_f16(a, d->N, &t, &lds) != ADELL_OK) return ADELL_E_UNSUPPORTED;
  if (adell_splitk_shares(a, t, d->N) > 1)
    return adell_fold_tiles((long)a.Do * a.Ho * a.Wo, d->N, a.Cout, (long)a.ntx * a.nty * a.ntz);
  return a.ntx * a.nty * a.ntz;
}

extern "C" long adell_pack_weight_f16x3_bytes(int mode, int dim0, int dim1, int taps) {
  if (mode < 0 || mode > 1 || dim0 <= 0 || dim1 <= 0 || taps <= 0) return ADELL_E_BADARG;
  const long N = mode == 0 ? dim0 : dim1, K = mode == 0 ? dim1 : dim0;
  return (long)taps * N * ((K + 15) / 16) * 64;
}

// w: canonical conv weight [Cout=dim0][Cin=dim1][taps]. out: split fp16 tiles
// (adell_pack_weight_f16x3_bytes). wscale: one float per GEMM column (mode 0: Cout,
// mode 1: Cin): the factor that undoes that column's power-of-two scale.
extern "C" int adell_pack_weight_f16x3(const float* w, void* out, float* wscale, int mode,
                                       int dim0, int dim1, int KD, int KH, int KW,
                                       void* stream) {
  ADELL_REQUIRE(w && out && wscale, "pack_weight_f16x3: null pointer");
  ADELL_REQUIRE(mode == 0 || mode == 1, "pack_weight_f16x3: mode must be 0 (fwd) or 1 (bwd-data)");
  ADELL_REQUIRE(dim0 > 0 && dim1 > 0 && KD > 0 && KH > 0 && KW > 0, "pack_weight_f16x3: bad dims");
  const int N = mode == 0 ? dim0 : dim1;
  hipLaunchKernelGGL(adell_pack_weight_f16_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, w,
                     (_Float16*)out, wscale, mode, dim0, dim1, KD * KH * KW);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// table: DEVICE array of `entries` rows of 8 int64 {w, out, wscale, mode, dim0, dim1, taps,
// first block}; total_blocks = sum of the GEMM-column counts.
extern "C" int adell_pack_weight_f16x3_multi(const long* table, int entries, long total_blocks,
                                             void* stream) {
  ADELL_REQUIRE(table && entries > 0 && total_blocks > 0 && total_blocks < 0x7fffffffL,
                "pack_weight_f16x3_multi: bad arguments");
  hipLaunchKernelGGL(adell_pack_weight_f16_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0,
                     (hipStream_t)stream, table, entries);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_conv3d_fwd_f16x3(const adell_conv3d_desc* d, const float* x0,
                                      const float* x1, const void* w_split,
                                      const float* wscale, const float* bias,
                                      const float* residual, float* y, float* stat_partials,
                                      int partial_rows, uint32_t* in_absmax, void* stream) {
  ConvArgs a;
  int rc = adell_fill_fwd(a, d, x0, x1, bias, residual, y, stat_partials, partial_rows);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split && wscale, "conv_fwd_f16x3: null weights");
  ConvF16Extra e = {(const _Float16*)w_split, wscale, in_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream);
}

// Workspace that lets the forward / backward-data calls below split K on small problems
// (0: this problem never splits).
extern "C" long adell_conv3d_splitk_workspace(const adell_conv3d_desc* d, int backward_data) {
  if (!d) return ADELL_E_BADARG;
  ConvArgs a;
  float dummy;
  int rc = backward_data ? adell_fill_bwd_data(a, d, &dummy, &dummy, d->C1 > 0 ? &dummy : nullptr)
                         : adell_fill_fwd(a, d, &dummy, d->C1 > 0 ? &dummy : nullptr, nullptr,
                                          nullptr, &dummy, nullptr);
  if (rc != ADELL_OK) return 0;
  ConvTile t;
  size_t lds;
  if (adell_plan_f16(a, d->N, &t, &lds) != ADELL_OK) return 0;
  const int shares = adell_splitk_shares(a, t, d->N);
  if (shares <= 1) return 0;
  return (long)sizeof(float) * shares * d->N * a.Do * a.Ho * a.Wo * a.Cout;
}

extern "C" int adell_conv3d_fwd_f16x3_ws(const adell_conv3d_desc* d, const float* x0,
                                         const float* x1, const void* w_split,
                                         const float* wscale, const float* bias,
                                         const float* residual, float* y, float* stat_partials,
                                         int partial_rows, uint32_t* in_absmax, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  ConvArgs a;
  int rc = adell_fill_fwd(a, d, x0, x1, bias, residual, y, stat_partials, partial_rows);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split && wscale, "conv_fwd_f16x3: null weights");
  ConvF16Extra e = {(const _Float16*)w_split, wscale, in_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream, workspace, workspace_bytes);
}

extern "C" int adell_conv3d_bwd_data_f16x3_ws(const adell_conv3d_desc* d, const float* dy,
                                              const void* w_split_bwd, const float* wscale,
                                              float* dx0, float* dx1, uint32_t* dy_absmax,
                                              void* workspace, size_t workspace_bytes,
                                              void* stream) {
  ConvArgs a;
  int rc = adell_fill_bwd_data(a, d, dy, dx0, dx1);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split_bwd && wscale, "conv_bwd_data_f16x3: null weights");
  ConvF16Extra e = {(const _Float16*)w_split_bwd, wscale, dy_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream, workspace, workspace_bytes);
}

// The same with `add0` ([N][D][H][W][C0], one destination only) added to dx0 in the epilogue: the
// gradient a residual link sends straight to the block input (`op(X) + X`, res_blocks.py:192),
// which autograd would otherwise add in a separate full-size pass.
extern "C" int adell_conv3d_bwd_data_f16x3_add(const adell_conv3d_desc* d, const float* dy,
                                               const void* w_split_bwd, const float* wscale,
                                               const float* add0, float* dx0,
                                               uint32_t* dy_absmax, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  ConvArgs a;
  int rc = adell_fill_bwd_data(a, d, dy, dx0, nullptr);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split_bwd && wscale && add0, "conv_bwd_data_f16x3_add: null pointer");
  ADELL_REQUIRE(d->C1 == 0, "conv_bwd_data_f16x3_add: one destination only");
  a.res = add0;
  ConvF16Extra e = {(const _Float16*)w_split_bwd, wscale, dy_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream, workspace, workspace_bytes);
}

// Backward-data whose destination(s) are gradients with respect to the OUTPUT of a fused
// norm -> dropout -> activation site (adn_fn.py:140-152): the epilogue applies the site's
// activation / dropout derivative and reduces the two per-channel sums the normalisation's backward
// needs, so the site's own backward is left with ONE elementwise pass (adell_norm_act_bwd_from_dt).
// site0 / site1 (either may be null: plain destination) describe the sites behind dx0 / dx1;
// partials: [N][ntiles][C0 + C1][2] floats, ntiles = adell_conv3d_bwd_data_f16x3_adn_ntiles(d)
// (0: this problem does not take the fused epilogue -- small / ragged / split-K launches).
extern "C" int adell_conv3d_bwd_data_f16x3_adn_ntiles(const adell_conv3d_desc* d) {
  if (!d) return 0;
  ConvArgs a;
  alignas(16) static float dummy[4];   // (the plan looks at pointer alignment)
  if (adell_fill_bwd_data(a, d, dummy, dummy, d->C1 > 0 ? dummy : nullptr) != ADELL_OK) return 0;
  ConvF16Extra e = {};
  // the plan of the real call (which passes the split-K workspace when there is one)
  static float ws_probe;
  const long wsb = adell_conv3d_splitk_workspace(d, 1);
  return adell_conv_dispatch_f16(a, e, d->N, nullptr, wsb > 0 ? &ws_probe : nullptr,
                                 wsb > 0 ? (size_t)wsb : 0, -1);
}

extern "C" int adell_conv3d_bwd_data_f16x3_adn(const adell_conv3d_desc* d, const float* dy,
                                               const void* w_split_bwd, const float* wscale,
                                               const float* add0, float* dx0, float* dx1,
                                               uint32_t* dy_absmax, const adell_adn_site* site0,
                                               const adell_adn_site* site1, float* partials,
                                               int partial_rows, void* stream) {
  ConvArgs a;
  int rc = adell_fill_bwd_data(a, d, dy, dx0, dx1);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split_bwd && wscale && partials, "conv_bwd_data_f16x3_adn: null pointer");
  ADELL_REQUIRE(site0 || site1, "conv_bwd_data_f16x3_adn: no site");
  ADELL_REQUIRE(!add0 || d->C1 == 0, "conv_bwd_data_f16x3_adn: add0 needs one destination");
  a.res = add0;
  a.part = partials;
  a.part_rows = partial_rows;
  ConvF16Extra e = {(const _Float16*)w_split_bwd, wscale, dy_absmax, nullptr, nullptr, nullptr};
  const adell_adn_site* sites[2] = {site0, site1};
  const long V = (long)a.Do * a.Ho * a.Wo;
  for (int k = 0; k < 2; ++k) {
    const adell_adn_site* s = sites[k];
    if (!s) continue;
    const int C = k == 0 ? a.ysplit : a.Cout - a.ysplit;
    ADELL_REQUIRE(C > 0 && s->y && s->mean && s->rstd, "conv_bwd_data_f16x3_adn: bad site %d", k);
    ADELL_REQUIRE(s->drop_p >= 0.f && s->drop_p < 1.f && (s->drop_p == 0.f || s->keep_mask),
                  "conv_bwd_data_f16x3_adn: dropout needs the forward's keep mask");
    ADELL_REQUIRE(s->act == ADELL_ACT_IDENTITY || s->act == ADELL_ACT_SILU ||
                      s->act == ADELL_ACT_RELU || s->act == ADELL_ACT_LEAKY_RELU,
                  "conv_bwd_data_f16x3_adn: activation %d has no fused derivative", s->act);
    e.adn[k].y = s->y; e.adn[k].mean = s->mean; e.adn[k].rstd = s->rstd;
    e.adn[k].mask = s->drop_p > 0.f ? (const unsigned*)s->keep_mask : nullptr;
    e.adn[k].keep_scale = 1.0f / (1.0f - s->drop_p);
    e.adn[k].act_p = s->act_p;
    e.adn[k].act = s->act;
    e.adn[k].groups = (int)((V * C + 255) / 256);
  }
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream, nullptr, 0, 1);
}

// Backward-data of a stride-2 conv by parity classes. dX[2i + p] (p in {0,1}^3) only receives the
// taps t with t = p + P (mod 2) per axis, so each of the 8 classes is a stride-1 conv of dY with a
// (1..2)^3 sub-kernel whose outputs land on a stride-2 lattice of dX (the pixel-shuffle store of
// the transposed conv with one sub-position): exactly the useful MFMA work, where the
// zero-insertion formulation multiplies 7 zeros out of 8. w_split[c] / wscale[c]: the class's
// sub-kernel w[:, :, t0z::2, t0y::2, t0x::2] packed with mode 1; c = 4 pz + 2 py + px.
// Needs even input dims, one destination (C1 = 0), k = 3.
static int adell_bwd_data_s2_classes(const adell_conv3d_desc* d, const float* dy,
                                     const void* const* w_split, const float* const* wscale,
                                     const float* add0, float* dx, uint32_t* dy_absmax,
                                     void* stream) {
  int rc = adell_check_desc(d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(dy && w_split && wscale && dx, "conv_bwd_data_s2: null pointer");
  ADELL_REQUIRE(d->C1 == 0 && d->SD == 2 && d->SH == 2 && d->SW == 2 && d->KD == 3 && d->KH == 3 &&
                    d->KW == 3 && d->D % 2 == 0 && d->H % 2 == 0 && d->W % 2 == 0 &&
                    d->PD <= 2 && d->PH <= 2 && d->PW <= 2,
                "conv_bwd_data_s2: stride 2, k = 3, even input dims, one destination only");
  for (int c = 0; c < 8; ++c) {
    const int pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
    const int p3[3] = {pz, py, px}, P3[3] = {d->PD, d->PH, d->PW};
    int n3[3], pad3[3];
    for (int ax = 0; ax < 3; ++ax) {
      const int t0 = (p3[ax] + P3[ax]) & 1;
      n3[ax] = (3 - t0 + 1) / 2;                 // taps t0, t0 + 2, ... below 3
      const int cc = (p3[ax] + P3[ax]) / 2;      // dX[2i + p] = sum_s dY[i + cc - s] Wsub[s]
      pad3[ax] = n3[ax] - 1 - cc;
      ADELL_REQUIRE(pad3[ax] >= 0, "conv_bwd_data_s2: padding > 1 not supported for this class");
    }
    ADELL_REQUIRE(w_split[c] && wscale[c], "conv_bwd_data_s2: null class weights");
    ConvArgs a = {};
    a.x0 = dy;
    const size_t class_origin = ((size_t)(pz * d->H + py) * d->W + px) * d->C0;
    a.y0 = dx + class_origin;
    a.res = add0 ? add0 + class_origin : nullptr;   // dX-shaped: read where the class stores
    a.D = d->Do; a.H = d->Ho; a.W = d->Wo;
    a.C0 = d->Cout; a.C1 = 0; a.Cin = d->Cout; a.Cout = d->C0;
    a.KD = n3[0]; a.KH = n3[1]; a.KW = n3[2];
    a.SD = a.SH = a.SW = 1;
    a.PD = pad3[0]; a.PH = pad3[1]; a.PW = pad3[2];
    a.UPS = a.UPSY = a.UPSZ = 1;
    a.Do = d->D / 2; a.Ho = d->H / 2; a.Wo = d->W / 2;
    a.ysplit = a.Cout; a.Cs = a.Cout;
    a.shuffle = 8 | 7 | (add0 ? 16 : 0);         // rows on the stride-2 lattice, sub-position 0
    ConvF16Extra e = {(const _Float16*)w_split[c], wscale[c], c == 0 ? dy_absmax : nullptr,
                      nullptr, nullptr, nullptr};
    rc = adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream);
    if (rc != ADELL_OK) return rc;
  }
  return ADELL_OK;
}

extern "C" int adell_conv3d_bwd_data_s2_f16x3(const adell_conv3d_desc* d, const float* dy,
                                              const void* const* w_split,
                                              const float* const* wscale, float* dx,
                                              uint32_t* dy_absmax, void* stream) {
  return adell_bwd_data_s2_classes(d, dy, w_split, wscale, nullptr, dx, dy_absmax, stream);
}

// ... + add0 (dX-shaped: the gradient another consumer of the same input parked, see
// functional.GradCarry) in the epilogue of every class launch.
extern "C" int adell_conv3d_bwd_data_s2_f16x3_add(const adell_conv3d_desc* d, const float* dy,
                                                  const void* const* w_split,
                                                  const float* const* wscale, const float* add0,
                                                  float* dx, uint32_t* dy_absmax, void* stream) {
  ADELL_REQUIRE(add0, "conv_bwd_data_s2_add: null add0");
  return adell_bwd_data_s2_classes(d, dy, w_split, wscale, add0, dx, dy_absmax, stream);
}

// ConvTranspose3d (kernel = stride = factors) on the f16x3 kernel. Forward: w_split = the
// VIRTUAL 1x1x1 conv weight V[(f, co)][ci] = w[ci][co][f] packed with mode 0 (wscale has
// F*Cout entries); backward-data: the torch weight [Cin][Cout][taps] read as a conv weight with
// Cin outputs and Cout inputs, packed with mode 0 as well.
extern "C" int adell_convtranspose3d_fwd_f16x3(int N, int D, int H, int W, int Cin, int Cout,
                                               int FD, int FH, int FW, const float* x,
                                               const void* w_split, const float* wscale,
                                               const float* bias, float* y, uint32_t* in_absmax,
                                               void* stream) {
  ADELL_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convT_fwd: bad dims");
  ADELL_REQUIRE(adell_convt_factors_ok(FD, FH, FW), "convT_fwd: kernel=stride must be 1 or 2 per dim");
  ADELL_REQUIRE(x && w_split && wscale && y, "convT_fwd_f16x3: null pointer");
  ConvArgs a = {};
  a.x0 = x; a.bias = bias; a.y0 = y;
  a.D = D; a.H = H; a.W = W;
  a.C0 = Cin; a.C1 = 0; a.Cin = Cin; a.Cout = FD * FH * FW * Cout;
  a.KD = a.KH = a.KW = 1;
  a.SD = a.SH = a.SW = 1;
  a.UPS = a.UPSY = a.UPSZ = 1;
  a.Do = D; a.Ho = H; a.Wo = W;
  a.ysplit = a.Cout; a.Cs = Cout;
  a.shuffle = 8 | (FW - 1) | ((FH - 1) << 1) | ((FD - 1) << 2);
  ConvF16Extra e = {(const _Float16*)w_split, wscale, in_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, N, (hipStream_t)stream);
}

extern "C" int adell_convtranspose3d_bwd_data_f16x3(int N, int D, int H, int W, int Cin, int Cout,
                                                    int FD, int FH, int FW, const float* dy,
                                                    const void* w_split_bwd, const float* wscale,
                                                    float* dx, uint32_t* dy_absmax, void* stream) {
  ADELL_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0,
                "convT_bwd_data: bad dims");
  ADELL_REQUIRE(adell_convt_factors_ok(FD, FH, FW), "convT_bwd_data: bad factors");
  ADELL_REQUIRE(dy && w_split_bwd && wscale && dx, "convT_bwd_data_f16x3: null pointer");
  ConvArgs a = {};
  a.x0 = dy; a.y0 = dx;
  a.D = FD * D; a.H = FH * H; a.W = FW * W;
  a.C0 = Cout; a.C1 = 0; a.Cin = Cout; a.Cout = Cin;
  a.KD = a.SD = FD; a.KH = a.SH = FH; a.KW = a.SW = FW;
  a.UPS = a.UPSY = a.UPSZ = 1;
  a.Do = D; a.Ho = H; a.Wo = W;
  a.ysplit = a.Cout; a.shuffle = 0; a.Cs = a.Cout;
  ConvF16Extra e = {(const _Float16*)w_split_bwd, wscale, dy_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, N, (hipStream_t)stream);
}

extern "C" int adell_conv3d_bwd_data_f16x3(const adell_conv3d_desc* d, const float* dy,
                                           const void* w_split_bwd, const float* wscale,
                                           float* dx0, float* dx1, uint32_t* dy_absmax,
                                           void* stream) {
  ConvArgs a;
  int rc = adell_fill_bwd_data(a, d, dy, dx0, dx1);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split_bwd && wscale, "conv_bwd_data_f16x3: null weights");
  ConvF16Extra e = {(const _Float16*)w_split_bwd, wscale, dy_absmax, nullptr, nullptr, nullptr};
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream);
}


// ---------------------------------------------------------------------------
// Split-row sources (round 4): the forward of a 3x3x3 stride-1 layer whose input(s) the producer
// already wrote as the 64-byte hi | lo rows of the kernels' LDS image (adell_norm_act_fwd_split;
// ConvF16Extra::xs0 / xs1). xk0 / xk1 non-null: that source is rows with exponents
// xk[N][C / 16]; null: fp32 as in adell_conv3d_fwd_f16x3. _rows_ok: 1 when the forward plan of
// `d` stages rows (the specialised instances; never split-K).
// ---------------------------------------------------------------------------
extern "C" int adell_conv3d_f16x3_rows_ok(const adell_conv3d_desc* d) {
  ConvArgs a;
  alignas(16) static float dummy[4];
  if (adell_fill_fwd(a, d, dummy, d && d->C1 > 0 ? dummy : nullptr, nullptr, nullptr, dummy,
                     nullptr) != ADELL_OK)
    return 0;
  ConvF16Extra e = {};
  // (the plan of adell_conv3d_fwd_f16x3_ws, which the caller would otherwise use: a layer that
  // runs split-K there is too small to gain from rows)
  static float ws_probe;
  const long wsb = adell_conv3d_splitk_workspace(d, 0);
  return adell_conv_dispatch_f16(a, e, d->N, nullptr, wsb > 0 ? &ws_probe : nullptr,
                                 wsb > 0 ? (size_t)wsb : 0, -2);
}

extern "C" int adell_conv3d_fwd_f16x3_rows(const adell_conv3d_desc* d, const void* x0,
                                           const int* xk0, const void* x1, const int* xk1,
                                           const void* w_split, const float* wscale,
                                           const float* bias, const float* residual, float* y,
                                           float* stat_partials, int partial_rows,
                                           uint32_t* in_absmax, void* stream) {
  ConvArgs a;
  int rc = adell_fill_fwd(a, d, (const float*)x0, (const float*)x1, bias, residual, y,
                          stat_partials, partial_rows);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(w_split && wscale, "conv_fwd_f16x3_rows: null weights");
  ADELL_REQUIRE(xk0 || xk1, "conv_fwd_f16x3_rows: no split-row source (use adell_conv3d_fwd_f16x3)");
  ADELL_REQUIRE(!xk1 || x1, "conv_fwd_f16x3_rows: xk1 without x1");
  ConvF16Extra e = {(const _Float16*)w_split, wscale, in_absmax,
                    xk0 ? (const char*)x0 : nullptr, xk1 ? (const char*)x1 : nullptr, xk0, xk1};
  return adell_conv_dispatch_f16(a, e, d->N, (hipStream_t)stream);
}
