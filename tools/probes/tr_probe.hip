// Microprobe (gfx950): semantics and LDS-bank cost of ds_read_b64_tr_b16 for candidate wgrad LDS layouts.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o tools/probes/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ u32x2 tr_read(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// pattern 0: addr = lane*8 (linear).  pattern 1: lane i of each 16-group -> row (i>>2) at 128 B stride, col group (i&3)*8 B,
// groups at +32 B.  pattern 2: same with rows (i&3), col group (i>>2).
__global__ void sem_kernel(uint16_t* out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) ((volatile uint16_t*)lds)[i] = (uint16_t)i;   // volatile: the asm read is invisible to DSE
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds;
    uint32_t addr;
    if (pattern == 0) addr = l * 8;
    else if (pattern == 1) addr = (i >> 2) * 128 + (i & 3) * 8 + g * 32;
    else addr = (i & 3) * 128 + (i >> 2) * 8 + g * 32;
    u32x2 v = tr_read(base + addr);
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}

// timing: 4 waves per block, every CU busy; each wave issues NREP * 16 tr reads with the layout's addresses; reports cycles.
// layout 0: [px][64ch] rows of 128 B, no swizzle; lane -> key (per sem), 16-lane group g: cols 16*(g&1), keys 8*(g>>1)+{0..3}
// layout 1: same + 64-B XOR swizzle keyed on (row>>1)&1
// layout 2: rows of 256 B (128 ch), no swizzle
// layout 3: rows of 256 B, 16-B-piece XOR swizzle keyed on row&7 (piece ^= (row&7)<<1 keeps 32-B pairs)
// layout 4: plain ds_read_b128 row-per-lane with the conv kernel's swizzle (reference cost)
__global__ void time_kernel(uint32_t* cyc, int layout, int rowmap) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    const int l = threadIdx.x & 63, i = l & 15, g = (l >> 4);
    const int key = rowmap ? (i & 3) : (i >> 2), cg = rowmap ? (i >> 2) : (i & 3);
    const int row = 8 * (g >> 1) + key;            // pixel row within a 16-row k-step
    const int col16 = (g & 1);                      // which 16-channel group of the 32 the MFMA tile covers
    uint32_t addr;
    if (layout == 0) addr = row * 128 + col16 * 32 + cg * 8;
    else if (layout == 1) addr = row * 128 + ((col16 * 32 + cg * 8) ^ (((row >> 1) & 1) << 6));
    else if (layout == 2) addr = row * 256 + col16 * 32 + cg * 8;
    else if (layout == 3) addr = row * 256 + ((col16 * 32 + cg * 8) ^ ((row & 7) << 5));
    else addr = (l & 31) * 128 + (((l >> 5) ^ (((l & 31) >> 1) & 7)) << 4);
    addr += (threadIdx.x >> 6) * 16384;
    u32x2 acc = {0, 0};
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < 256; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (layout == 4) {
                uint32_t a, b, c, d;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(__attribute__((ext_vector_type(4))) uint32_t*)&a) : "v"(addr), "n"(0));
                (void)b; (void)c; (void)d;
            } else {
                u32x2 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
                acc.x ^= 0;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = (uint32_t)(t1 - t0) + acc.x;
}

int main() {
    uint16_t* d; hipMalloc(&d, 256 * 2 * 3);
    uint16_t h[256];
    for (int p = 0; p < 3; ++p) {
        sem_kernel<<<1, 64>>>(d, p);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d:\n", p);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %5d(src lane-slot %2d elem %d)", h[l * 4 + j], (h[l * 4 + j] / 4) , h[l * 4 + j] % 4);
            printf("\n");
            if (p > 0 && l == 15) { printf("  ...\n"); l = 47; }
        }
    }
    return 0;
}
