#include <hip/hip_runtime.h>
#include <stdint.h>
extern __shared__ __attribute__((aligned(16))) float dyn[];
__device__ __forceinline__ void dma_x4(const void* gbase, uint32_t voff, void* lds_wave_base) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    uint32_t saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "s"(la), "v"(voff), "s"(gbase) : "memory");
}
__global__ void k(const float* __restrict__ g, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < 4; i += 4) dma_x4(g, (uint32_t)(i * 1024 + lane * 16), dyn + i * 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = dyn[i];
}
int main() {
    float *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, g, o);
    float r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) if (r[i] != h[i]) { if (bad < 5) printf("mismatch %d: %f\n", i, r[i]); ++bad; }
    printf("x4 dma bad=%d\n", bad); return bad != 0;
}
