// LDS / cross-lane micro-benchmark for gfx950: per-CU cost of the instructions the fused board kernel leans on.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_ubench.hip -o scripts/ubench/lds_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define ITERS 2000
#define UNROLL 8
enum { OP_READ_RANDOM, OP_READ_LINEAR, OP_WRITE_RANDOM, OP_WRITE_LINEAR, OP_BPERMUTE, OP_ATOMIC_ADD_RANDOM, OP_READ_B128, OP_DPP_SCAN, OP_READ_U16, OP_READLANE, OP_COUNT };
const char* NAMES[] = {"ds_read_b32 random", "ds_read_b32 linear", "ds_write_b32 random", "ds_write_b32 linear", "ds_bpermute_b32", "ds_add_f32 random (2 per addr)", "ds_read_b128 linear", "6-op DPP scan (VALU)", "ds_read_u16 linear", "v_readlane + add"};

__device__ __forceinline__ float scan(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return v;
}

template <int OP>
__global__ void __launch_bounds__(512) k(const int* __restrict__ idx, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 16384 floats
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 512) lds[i] = (float)i;
    __syncthreads();
    int a[UNROLL];
    for (int u = 0; u < UNROLL; ++u) a[u] = idx[(blockIdx.x * 512 + tid) * UNROLL + u] & 16383;
    float acc = 0.f;
    float4 acc4 = make_float4(0, 0, 0, 0);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == OP_READ_RANDOM) acc += lds[a[u]];
            if (OP == OP_READ_LINEAR) acc += lds[(tid + u * 512 + it) & 16383];
            if (OP == OP_WRITE_RANDOM) lds[a[u]] = acc + it;
            if (OP == OP_WRITE_LINEAR) lds[(tid + u * 512 + it) & 16383] = acc + it;
            if (OP == OP_BPERMUTE) acc += __int_as_float(__builtin_amdgcn_ds_bpermute((a[u] & 63) << 2, __float_as_int(acc + u)));
            if (OP == OP_ATOMIC_ADD_RANDOM) atomicAdd(&lds[(a[u] >> 1)], 1.0f);
            if (OP == OP_READ_B128) { float4 v = *(float4*)&lds[((tid + u * 512 + it) & 4095) * 4]; acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w; }
            if (OP == OP_DPP_SCAN) acc = scan(acc + u);
            if (OP == OP_READ_U16) acc += ((unsigned short*)lds)[(tid + u * 512 + it) & 32767];
            if (OP == OP_READLANE) acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 45));
        }
    }
    out[blockIdx.x * 512 + tid] = acc + acc4.x + acc4.y + acc4.z + acc4.w + lds[tid];
}

template <int OP>
void run(int wg_per_cu, const int* d_idx, float* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(512), 65536, 0, d_idx, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(512), 65536, 0, d_idx, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_cu = (double)wg_per_cu * 8 * ITERS * UNROLL;
    printf("%-34s wg/cu=%d  %.3f ms  -> %.2f ns per wave-instruction per CU (%.1f cycles @2.4GHz)\n", NAMES[OP], wg_per_cu, ms, ms * 1e6 / wave_instr_per_cu,
           ms * 1e6 / wave_instr_per_cu * 2.4);
}

int main() {
    int n = 256 * 2 * 512 * UNROLL;
    int* h = (int*)malloc(n * sizeof(int));
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = rand();
    int* d_idx; float* d_out;
    hipMalloc(&d_idx, n * sizeof(int)); hipMalloc(&d_out, 256 * 2 * 512 * sizeof(float));
    hipMemcpy(d_idx, h, n * sizeof(int), hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<OP_READ_RANDOM>(w, d_idx, d_out); run<OP_READ_LINEAR>(w, d_idx, d_out); run<OP_WRITE_RANDOM>(w, d_idx, d_out);
        run<OP_WRITE_LINEAR>(w, d_idx, d_out); run<OP_BPERMUTE>(w, d_idx, d_out); run<OP_ATOMIC_ADD_RANDOM>(w, d_idx, d_out);
        run<OP_READ_B128>(w, d_idx, d_out); run<OP_DPP_SCAN>(w, d_idx, d_out); run<OP_READ_U16>(w, d_idx, d_out); run<OP_READLANE>(w, d_idx, d_out);
    }
    return 0;
}
