// Thin device-side vocabulary used by every kernel, so that the SAME kernel source compiles
//   (a) with hipcc for gfx950 (the product), and
//   (b) with g++ -DPRL_EMU against tests/emu/prl_emu.h (a fiber-based SIMT emulator used only by the CPU test-suite).
// Kernels use 1-D grids / blocks, dynamic LDS only (16-byte aligned carve-outs, cdna guide G17) and wave64 cross-lane ops.
#pragma once
#include "prl_defs.h"

#if defined(PRL_EMU)
#include "prl_emu.h"
#else
#include <hip/hip_runtime.h>

#define PRL_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), __VA_ARGS__)

PRL_DEV PRL_INLINE unsigned prl_tid() { return threadIdx.x; }
PRL_DEV PRL_INLINE unsigned prl_bid() { return blockIdx.x; }
PRL_DEV PRL_INLINE unsigned prl_nthreads() { return blockDim.x; }
PRL_DEV PRL_INLINE unsigned prl_nblocks() { return gridDim.x; }
PRL_DEV PRL_INLINE unsigned prl_lane() { return threadIdx.x & 63u; }
PRL_DEV PRL_INLINE void prl_sync() { __syncthreads(); }
PRL_DEV PRL_INLINE char* prl_smem() {
    extern __shared__ __attribute__((aligned(16))) char prl_dyn_smem[];
    return prl_dyn_smem;
}
// wave64 cross-lane primitives
PRL_DEV PRL_INLINE float prl_shfl_up(float v, unsigned delta) { return __shfl_up(v, delta, 64); }
PRL_DEV PRL_INLINE float prl_shfl(float v, int src_lane) { return __shfl(v, src_lane, 64); }
PRL_DEV PRL_INLINE int prl_shfl_i(int v, int src_lane) { return __shfl(v, src_lane, 64); }
PRL_DEV PRL_INLINE int prl_shfl_up_i(int v, unsigned delta) { return __shfl_up(v, delta, 64); }
PRL_DEV PRL_INLINE unsigned long long prl_ballot(int pred) { return __ballot(pred); }
#endif

// lane-local helpers on 64-bit masks
PRL_HD PRL_INLINE int prl_popc64(unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
PRL_HD PRL_INLINE int prl_msb64(unsigned long long x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)x);
#else
    return 63 - __builtin_clzll(x);
#endif
}
PRL_HD PRL_INLINE int prl_lsb64(unsigned long long x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
