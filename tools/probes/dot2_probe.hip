// Probe (gfx950): semantics of __builtin_amdgcn_fdot2_f32_bf16 (v_dot2c_f32_bf16) -- which halves multiply, is the addend kept,
// and do DEPENDENT chains (acc = dot2(x, w, acc) back to back, as a convolution inner loop issues them) give the right sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dot2_probe.hip -o tools/probes/dot2_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* o) {
    const int i = threadIdx.x;
    o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, a[i]), __builtin_bit_cast(bf16x2v_t, b[i]), c[i], false);
}
__device__ __forceinline__ float dot2s(uint32_t a, uint32_t b, float acc) {      // operands as scalar dwords: see conv_c1.hip
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, a), __builtin_bit_cast(bf16x2v_t, b), acc, false);
}
__global__ void chain_scalar_args(const u32x4* x, const u32x4* w, int n, float* o) {
    const int l = threadIdx.x;
    float acc = 0.f;
    for (int v = 0; v < n; ++v) {
        const u32x4 xv = x[v * 64 + l], wv = w[v];
        acc = dot2s(xv.x, wv.x, acc); acc = dot2s(xv.y, wv.y, acc); acc = dot2s(xv.z, wv.z, acc); acc = dot2s(xv.w, wv.w, acc);
    }
    o[l] = acc;
}
// n vectors of 8 bf16 per lane, one dependent chain / four independent chains (bit_cast applied directly to the vector elements)
__global__ void chain(const u32x4* x, const u32x4* w, int n, float* o1, float* o4) {
    const int l = threadIdx.x;
    float acc = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    for (int v = 0; v < n; ++v) {
        const u32x4 xv = x[v * 64 + l], wv = w[v];
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.x), __builtin_bit_cast(bf16x2v_t, wv.x), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.y), __builtin_bit_cast(bf16x2v_t, wv.y), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.z), __builtin_bit_cast(bf16x2v_t, wv.z), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.w), __builtin_bit_cast(bf16x2v_t, wv.w), acc, false);
        p0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.x), __builtin_bit_cast(bf16x2v_t, wv.x), p0, false);
        p1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.y), __builtin_bit_cast(bf16x2v_t, wv.y), p1, false);
        p2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.z), __builtin_bit_cast(bf16x2v_t, wv.z), p2, false);
        p3 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, xv.w), __builtin_bit_cast(bf16x2v_t, wv.w), p3, false);
    }
    o1[l] = acc;
    o4[l] = (p0 + p1) + (p2 + p3);
}
static uint32_t pk(float lo, float hi) {
    uint32_t l, h; memcpy(&l, &lo, 4); memcpy(&h, &hi, 4);
    return (l >> 16) | (h & 0xffff0000u);
}
static float bf(uint32_t w, int hi) { uint32_t u = hi ? (w & 0xffff0000u) : (w << 16); float f; memcpy(&f, &u, 4); return f; }
int main() {
    uint32_t ha[4] = {pk(1, 0), pk(0, 1), pk(2, 3), pk(1, 1)}, hb[4] = {pk(5, 7), pk(5, 7), pk(10, 100), pk(0.5f, 0.25f)};
    float hc[4] = {0, 0, 1000, -1}, ho[4];
    uint32_t *a, *b; float *c, *o;
    (void)hipMalloc(&a, 16); (void)hipMalloc(&b, 16); (void)hipMalloc(&c, 16); (void)hipMalloc(&o, 16);
    (void)hipMemcpy(a, ha, 16, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 16, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, a, b, c, o);
    (void)hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
    printf("{\"dot2\": [%g, %g, %g, %g], \"expected\": [5, 7, 1320, -0.25]}\n", ho[0], ho[1], ho[2], ho[3]);
    const int n = 36;
    static uint32_t hx[36 * 64 * 4], hw[36 * 4];
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : hx) v = pk(rnd(), rnd());
    for (auto& v : hw) v = pk(rnd(), rnd());
    uint32_t *dx, *dw; float *d1, *d4;
    (void)hipMalloc(&dx, sizeof hx); (void)hipMalloc(&dw, sizeof hw); (void)hipMalloc(&d1, 256); (void)hipMalloc(&d4, 256);
    (void)hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, (const u32x4*)dx, (const u32x4*)dw, n, d1, d4);
    float r1[64], r4[64], rs[64];
    (void)hipMemcpy(r1, d1, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(r4, d4, 256, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(chain_scalar_args, dim3(1), dim3(64), 0, 0, (const u32x4*)dx, (const u32x4*)dw, n, d1);
    (void)hipMemcpy(rs, d1, 256, hipMemcpyDeviceToHost);
    double e1 = 0, e4 = 0, es = 0, mag = 0;
    for (int l = 0; l < 64; ++l) {
        double ref = 0;
        for (int v = 0; v < n; ++v)
            for (int d = 0; d < 4; ++d) {
                const uint32_t xv = hx[(v * 64 + l) * 4 + d], wv = hw[v * 4 + d];
                ref += (double)bf(xv, 0) * bf(wv, 0) + (double)bf(xv, 1) * bf(wv, 1);
            }
        e1 = fmax(e1, fabs(r1[l] - ref)); e4 = fmax(e4, fabs(r4[l] - ref)); es = fmax(es, fabs(rs[l] - ref)); mag = fmax(mag, fabs(ref));
    }
    printf("{\"vector_element_bitcast_dependent_max_err\": %.3g, \"vector_element_bitcast_independent_max_err\": %.3g, "
           "\"scalar_dword_args_max_err\": %.3g, \"max_ref\": %.3g}\n", e1, e4, es, mag);
    return 0;
}
