// Plane-parameter head shared by the LPG kernels (lpg.hip, lpg_chain.hip): raw 1x1-conv outputs -> sigmoid ->
// (theta, phi, dist) -> unit normal -> F.normalize (pytorch/bts.py:110-122, 222-226).
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float lpg_offset(int r, int k) { return ((float)r - (float)(k - 1) * 0.5f) / (float)k; }

struct Plane {
    float n1, n2, n3, n4;        // normalised plane
    float s0, s1, s2;            // sigmoids
    float st, ct, sp, cp;        // sin/cos theta, phi
    float m1, m2, m3, inv_norm;  // un-normalised normal and 1/max(norm, 1e-12)
};

__device__ __forceinline__ Plane plane_from_raw(float r0, float r1, float r2, float max_depth) {
    Plane p;
    p.s0 = act_sigmoid(r0); p.s1 = act_sigmoid(r1); p.s2 = act_sigmoid(r2);
    const float theta = __fdiv_rn(__fmul_rn(p.s0, 3.14159274101257324f), 3.0f);   // sigmoid * math.pi / 3 (bts.py:113)
    const float phi = __fmul_rn(__fmul_rn(p.s1, 3.14159274101257324f), 2.0f);     // sigmoid * math.pi * 2 (bts.py:114)
    sincosf(theta, &p.st, &p.ct);
    sincosf(phi, &p.sp, &p.cp);
    p.m1 = __fmul_rn(p.st, p.cp); p.m2 = __fmul_rn(p.st, p.sp); p.m3 = p.ct;     // bts.py:116-118
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(p.m1, p.m1), __fmul_rn(p.m2, p.m2)), __fmul_rn(p.m3, p.m3)));
    const float d = fmaxf(nrm, 1e-12f);                                           // F.normalize eps (bts.py:224)
    p.inv_norm = 1.f / d;
    p.n1 = p.m1 / d; p.n2 = p.m2 / d; p.n3 = p.m3 / d;
    p.n4 = __fmul_rn(p.s2, max_depth);                                            // bts.py:115
    return p;
}

}  // namespace
