// Shared helpers for the adell HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/adell_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ADELL_WAVE 64
// Pointer into GLOBAL memory (address space 1). A pointer rebuilt from integers (e.g. after
// readfirstlane) is a generic pointer: accessed with flat_load / flat_store, which count on lgkmcnt
// as well as vmcnt, so every LDS wait behind one becomes a wait for the memory operation too.
#define ADELL_GLOBAL __attribute__((address_space(1)))

void adell_set_error(const char* fmt, ...);

// Launch-plan switches (A/B tests of kernel instances). Read from the environment ONCE when the
// library is loaded (ADELL_IGEMM_NOSPEC, ADELL_IGEMM_NO8, ADELL_NO_SPLITK, ADELL_NO_WGRAD_TINY,
// ADELL_WGRAD_NOZRING, ADELL_ZR_MINSEG, ADELL_IGEMM_WS, ADELL_WS_MIN_ITEMS); afterwards only adell_set_tuning() changes them, so
// the per-launch host path never calls getenv().
struct AdellTuning {
  // launch-plan switches that select another BUILT path (the parity tests compare the two sides):
  int igemm_nospec;               // 3^3 stride-1 convs on the generic f16x3 instance, not the specialised ones
  int igemm_no8;                  // 32-column layers on 8x8x4 bricks, not the 8x8x8-brick instance
  int no_splitk;                  // 8^3 .. 16^3 levels without split-K
  int wgrad_nozring;              // weight gradients on the per-plane kernel, not the z-ring (nor the stride-2 sub-lattice kernel)
  int attn_nomfma;                // attention: vector-ALU kernels even for MFMA-eligible head dims
  int dw_nomfma;                  // depthwise 7^3: vector-ALU kernels only (exact fp32 FMAs) instead of the f16x3 Toeplitz MFMA form
  int dw_wgrad_nomfma;            // depthwise 7^3 weight gradient: vector-ALU tile kernel instead of the MFMA form
  int wgrad_no16;                 // z-ring weight gradient: 32 x 32 tiles even for 16-channel layers
  int igemm_no16;                 // forward / backward-data of 16 -> 16 layers: not the z-ring 16-column kernel
  int gemm_norows;                // f16x3 GEMM: never the streaming kernel for many-row Linear layers (gemm_rows.hip)
  int igemm_dbg, zr_dbg;          // timing experiments: always 0 unless built with -DADELL_DEBUG
};
extern AdellTuning g_adell_tune;
// Bumped by every adell_set_tuning / adell_debug_force_conv_cfg that changes the launch plan:
// anything a caller sized from an *_ntiles / *_workspace query is valid for one epoch only.
extern long g_adell_plan_epoch;
// A statistics / partial-sum buffer the caller sized for `have` rows per batch item, against the
// `need` rows the plan of THIS launch writes (the row count is also the buffer's stride, so more
// rows are as wrong as fewer): refuse before anything is launched.
#define ADELL_REQUIRE_ROWS(ptr, have, need, what)                                              \
  ADELL_REQUIRE((ptr) == nullptr || (long)(have) == (long)(need),                              \
                "%s: partial-sum buffer sized for %ld rows per item, the current launch plan "  \
                "writes %ld (query the matching *_ntiles again after adell_set_tuning)",        \
                what, (long)(have), (long)(need))
// Kernel-side timing experiments make results WRONG; they exist only in -DADELL_DEBUG builds.
#ifdef ADELL_DEBUG
#define ADELL_DBG(bits) (bits)
#else
#define ADELL_DBG(bits) 0
#endif

#define ADELL_CHECK_HIP(expr)                                                  \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      adell_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                      hipGetErrorString(_e));                                  \
      return ADELL_E_HIP;                                                      \
    }                                                                          \
  } while (0)

#define ADELL_REQUIRE(cond, ...)                                               \
  do {                                                                         \
    if (!(cond)) {                                                             \
      adell_set_error(__VA_ARGS__);                                            \
      return ADELL_E_BADARG;                                                   \
    }                                                                          \
  } while (0)

static inline int adell_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int adell_ilog2(int v) {
  int r = 0;
  while ((1 << r) < v) ++r;
  return r;
}
static inline bool adell_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float adell_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double adell_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Activation ids shared with the host side (adell_hip.h: ADELL_ACT_*).
// v_exp_f32 + v_rcp_f32: ~1 ulp each (2^-23-level relative error, far inside the 1e-4 parity
// bar) instead of the ~35-instruction IEEE expf + division, which made the elementwise
// kernels VALU-bound at ~2.5 TB/s.
__device__ __forceinline__ float adell_sigmoidf(float x) {
  return __frcp_rn(1.0f + __expf(-x));
}

// ---------------------------------------------------------------------------
// Activations
// ---------------------------------------------------------------------------
__device__ __forceinline__ float adell_act_fwd(int act, float x, float p) {
  switch (act) {
    case ADELL_ACT_SILU: return x * adell_sigmoidf(x);
    case ADELL_ACT_RELU: return x > 0.f ? x : 0.f;
    case ADELL_ACT_LEAKY_RELU: return x > 0.f ? x : p * x;
    case ADELL_ACT_PRELU: return x > 0.f ? x : p * x;
    case ADELL_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    case ADELL_ACT_SIGMOID: return adell_sigmoidf(x);
    case ADELL_ACT_TANH: return tanhf(x);
    case ADELL_ACT_ELU: return x > 0.f ? x : p * (expf(x) - 1.0f);
    default: return x;
  }
}
// d act(x) / dx
__device__ __forceinline__ float adell_act_grad(int act, float x, float p) {
  switch (act) {
    case ADELL_ACT_SILU: {
      const float s = adell_sigmoidf(x);
      return s * (1.0f + x * (1.0f - s));
    }
    case ADELL_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ADELL_ACT_LEAKY_RELU: return x > 0.f ? 1.f : p;
    case ADELL_ACT_PRELU: return x > 0.f ? 1.f : p;
    case ADELL_ACT_GELU: {
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
      const float pdf = 0.39894228040143268f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case ADELL_ACT_SIGMOID: {
      const float s = adell_sigmoidf(x);
      return s * (1.0f - s);
    }
    case ADELL_ACT_TANH: {
      const float t = tanhf(x);
      return 1.0f - t * t;
    }
    case ADELL_ACT_ELU: return x > 0.f ? 1.f : p * expf(x);
    default: return 1.f;
  }
}

// Replay counter of the dropout offsets. A kernel's (seed, offset) words are launch arguments: in a
// captured HIP graph they are frozen, and every replay would draw the masks of the captured step. Each
// translation unit with a dropout kernel therefore keeps ONE device word that is ADDED to the offset
// argument (0 unless a graph advances it: eager launches are unchanged), and the graph ends with the
// one-thread kernel that advances it by the number of offsets a step draws -- replay r of a step
// captured with offsets c .. c + K - 1 uses c + r K .. , exactly what eager step r would have drawn
// (adell_rng_advance, include/adell_hip.h; trainer.StepRunner.enable_graph).
#define ADELL_RNG_STEP_DEFINE(tu)                                                                \
  static __device__ unsigned g_adell_rng_step = 0;                                                \
  __global__ void adell_rng_advance_kernel_##tu(unsigned delta, int set) {                        \
    g_adell_rng_step = set ? delta : g_adell_rng_step + delta;                                    \
  }                                                                                               \
  int adell_rng_advance_##tu(unsigned delta, int set, hipStream_t st) {                           \
    hipLaunchKernelGGL(adell_rng_advance_kernel_##tu, dim3(1), dim3(1), 0, st, delta, set);      \
    return hipGetLastError() == hipSuccess ? ADELL_OK : ADELL_E_HIP;                              \
  }

// Philox-4x32 counter RNG, 7 rounds (Salmon et al., SC'11: the fewest rounds of this generator that
// pass BigCrush; 10 is the library default with a safety margin): one call -> 4 uniform 32-bit
// words. The dropout mask of element e of a tensor is a pure function of (seed, offset, e), so
// forward and backward regenerate it instead of storing it. The 32-bit multiplies are quarter-rate
// instructions: at 10 rounds the generator was 640 of the ~880 vector-ALU cycles the fused norm ->
// dropout -> activation forward spends per 64 float4, i.e. that HBM-bound-looking pass was bound by
// its RNG (4.1 TB/s in the step, 5.0 alone, 5.8 without dropout).
#ifndef ADELL_PHILOX_ROUNDS
#define ADELL_PHILOX_ROUNDS 7
#endif
__device__ __forceinline__ uint4 adell_philox4(uint32_t c0, uint32_t c1,
                                               uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < ADELL_PHILOX_ROUNDS; ++r) {
    // one 32 x 32 -> 64 bit multiply per word pair (v_mad_u64_u32) instead of a mul_hi and a mul_lo:
    // both are quarter-rate instructions, and the generator is what these passes spend their
    // vector-ALU time on
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return make_uint4(c0, c1, c2, c3);
}
