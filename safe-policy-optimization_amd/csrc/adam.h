// Adam arithmetic shared by the persistent update kernels (update.hip) and the wide-network path (ma_net.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace spo {

// Adam on one scalar.  Same update as torch.optim.Adam's single-tensor path
// (exp_avg.lerp_, exp_avg_sq.mul_.addcmul_, denom = sqrt(v)/sqrt(bc2) + eps, addcdiv_) with the
// square root and the two divisions done by v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of IEEE
// sequences: the parameter step changes by <= 3e-7 relative, i.e. < 1e-10 absolute per step.
struct AdamOut { float p, m, v; };
__device__ __forceinline__ AdamOut adam1(float p, float g, float m, float v, float b1, float b2, float eps,
                                         float step_size, float inv_bc2_sqrt) {
  AdamOut o;
  o.m = fmaf(1.f - b1, g - m, m);
  o.v = fmaf(v, b2, ((1.f - b2) * g) * g);
  const float denom = fmaf(__builtin_amdgcn_sqrtf(o.v), inv_bc2_sqrt, eps);
  o.p = fmaf(-step_size, o.m * __builtin_amdgcn_rcpf(denom), p);
  return o;
}
// The two per-step scalars of Adam's bias correction: step_size = lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t).  The powers are
// carried in double (one v_mul_f64 each per step); rounds 1-2 also formed the quotient and the square root in double -- ~40
// double-precision VALU instructions per helper wave and step (IEEE division and sqrt sequences at half / quarter rate),
// ~400 cycles of SIMD time that nothing overlaps (tools/mfma_valu_overlap.hip).  Now: 1 - beta^t is formed in double (exact
// to 1e-16) and rounded to fp32 once, the reciprocal and the reciprocal square root come from v_rcp_f32 / v_rsq_f32 with
// one Newton step and a residual correction: <= ~1.5e-7 relative on either scalar, half the error budget of the
// v_sqrt_f32 / v_rcp_f32 inside adam1 itself (tests: drift envelopes at full size, first steps at 1e-5).
#ifndef SPO_ADAM_SCALARS_F64
#define SPO_ADAM_SCALARS_F64 0       // 1: the double-precision quotient / square root of rounds 1-2 (A/B knob)
#endif
__device__ __forceinline__ void adam_scalars(float lr, double pw1, double pw2, float& step_size, float& inv_bc2s) {
#if SPO_ADAM_SCALARS_F64
  step_size = (float)((double)lr / (1.0 - pw1));
  inv_bc2s = (float)(1.0 / sqrt(1.0 - pw2));
#else
  const float bc1 = (float)(1.0 - pw1), bc2 = (float)(1.0 - pw2);
  float r = __builtin_amdgcn_rcpf(bc1);
  r = fmaf(fmaf(-bc1, r, 1.f), r, r);                 // Newton step on the reciprocal
  float qv = lr * r;
  qv = fmaf(fmaf(-bc1, qv, lr), r, qv);               // residual correction of the quotient
  step_size = qv;
  float y = __builtin_amdgcn_rsqf(bc2);
  y = fmaf(y, fmaf(-(bc2 * y), 0.5f * y, 0.5f), y);   // y (1.5 - 0.5 x y^2)
  inv_bc2s = y;
#endif
}

}  // namespace spo
