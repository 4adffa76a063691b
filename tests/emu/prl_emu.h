// SIMT emulator for the CPU test-suite (TEST INFRASTRUCTURE -- never part of the product).
//
// Runs the package's HIP kernel SOURCES (pokerrl_amd/csrc/*.hip compiled with g++ -x c++ -DPRL_EMU) on the host so that
// kernel logic (indexing, barriers, wave-level scans) can be checked against the oracle in the GPU-less CI container.
// One OS thread; every GPU thread is a ucontext fiber; blocks run one after another; `prl_sync()` and the wave64
// cross-lane ops are rendezvous points of the fibers of a block / wave. It models lock-step wave64 semantics only --
// no memory model, no performance. libpokerrl_hip.so is never built this way and pokerrl_amd/ never loads this build.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

namespace prl_emu {
struct Ctx {
    unsigned tid, bid, bdim, gdim;
    char* smem;
    unsigned bid_y;
};
extern thread_local Ctx* g_ctx;
void launch(const std::function<void()>& body, unsigned grid, unsigned block, size_t smem_bytes);
void launch2(const std::function<void()>& body, unsigned grid_x, unsigned grid_y, unsigned block, size_t smem_bytes);
void block_barrier();
uint64_t wave_exchange(uint64_t my_value, int src_lane);  // returns the value contributed by src_lane (0 if it exited)
uint64_t wave_ballot(int pred);
}  // namespace prl_emu

#define PRL_LAUNCH(kernel, grid, block, smem, stream, ...) \
    prl_emu::launch([=]() { kernel(__VA_ARGS__); }, (unsigned)(grid), (unsigned)(block), (size_t)(smem))

// 1 x grid_y workgroups: the kernel picks its work item with prl_bid_y(); prl_bid() is 0 and prl_nblocks() is 1
#define PRL_LAUNCH_Y(kernel, grid_y, block, smem, stream, ...) \
    prl_emu::launch2([=]() { kernel(__VA_ARGS__); }, 1u, (unsigned)(grid_y), (unsigned)(block), (size_t)(smem))
inline unsigned prl_bid_y() { return prl_emu::g_ctx->bid_y; }
inline unsigned prl_tid() { return prl_emu::g_ctx->tid; }
inline unsigned prl_bid() { return prl_emu::g_ctx->bid; }
inline unsigned prl_nthreads() { return prl_emu::g_ctx->bdim; }
inline unsigned prl_nblocks() { return prl_emu::g_ctx->gdim; }
inline unsigned prl_lane() { return prl_emu::g_ctx->tid & 63u; }
inline void prl_sync() { prl_emu::block_barrier(); }
inline void prl_sync_lds() { prl_emu::block_barrier(); }
// the emulator's "DMA" is an immediate copy (lane l writes dword l of the wave's destination row)
inline void prl_lds_dma_dword(const void* gbase, uint32_t byte_off, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + 4 * (prl_emu::g_ctx->tid & 63u), (const char*)gbase + byte_off, 4);
}
inline void prl_lds_dma_x4(const void* gbase, uint32_t byte_off, void* lds_wave_base) {
    memcpy((char*)lds_wave_base + 16 * (prl_emu::g_ctx->tid & 63u), (const char*)gbase + byte_off, 16);
}
// LDS addresses of the emulator: byte offsets into the workgroup's shared memory
inline uint32_t prl_lds_addr(const void* lds_ptr) { return (uint32_t)((const char*)lds_ptr - prl_emu::g_ctx->smem); }
inline void prl_lds_dma_x4_a(const void* gbase, uint32_t byte_off, uint32_t lds_wave_addr) {
    memcpy(prl_emu::g_ctx->smem + lds_wave_addr + 16 * (prl_emu::g_ctx->tid & 63u), (const char*)gbase + byte_off, 16);
}
inline void prl_dma_wait() {}
inline int prl_wave_uniform(int v) { return v; }
inline int prl_opaque_scalar(int v) { return v; }
inline int prl_opaque_lane(int v) { return v; }
inline char* prl_smem() { return prl_emu::g_ctx->smem; }

inline float prl_shfl(float v, int src_lane) {
    uint32_t u;
    memcpy(&u, &v, 4);
    uint32_t r = (uint32_t)prl_emu::wave_exchange(u, src_lane & 63);
    float f;
    memcpy(&f, &r, 4);
    return f;
}
inline float prl_shfl_up(float v, unsigned delta) {
    int lane = (int)prl_lane();
    int src = lane - (int)delta;
    float r = prl_shfl(v, src < 0 ? lane : src);
    return src < 0 ? v : r;  // HIP semantics: lanes below delta keep their own value
}
inline int prl_shfl_i(int v, int src_lane) { return (int)(uint32_t)prl_emu::wave_exchange((uint32_t)v, src_lane & 63); }
inline int prl_shfl_up_i(int v, unsigned delta) {
    int lane = (int)prl_lane();
    int src = lane - (int)delta;
    int r = prl_shfl_i(v, src < 0 ? lane : src);
    return src < 0 ? v : r;
}
inline unsigned long long prl_ballot(int pred) { return prl_emu::wave_ballot(pred); }
// DPP moves (see prl_device.h): lanes without a valid source receive 0
template <int D>
inline float prl_dpp_row_shr(float v) {
    int lane = (int)prl_lane();
    float r = prl_shfl(v, (lane & 15) >= D ? lane - D : lane);
    return (lane & 15) >= D ? r : 0.f;
}
template <int N>
inline float prl_dpp_row_share(float v) { return prl_shfl(v, ((int)prl_lane() & 48) + N); }  // every lane <- lane N of its row of 16
inline float prl_dpp_row_bcast15(float v) {
    int lane = (int)prl_lane();
    int row = lane >> 4;
    float r = prl_shfl(v, (row & 1) ? row * 16 - 1 : lane);
    return (row & 1) ? r : 0.f;
}
inline float prl_dpp_row_bcast31(float v) {
    int lane = (int)prl_lane();
    float r = prl_shfl(v, lane >= 32 ? 31 : lane);
    return lane >= 32 ? r : 0.f;
}
inline int prl_dpp_wave_shr1_i(int v, int fill) {
    int lane = (int)prl_lane();
    int r = prl_shfl_i(v, lane > 0 ? lane - 1 : 0);
    return lane > 0 ? r : fill;
}
inline float prl_readlane(float v, int lane) { return prl_shfl(v, lane); }
inline void prl_use(float&) {}

// ---- the sliver of the HIP runtime API the C-ABI layer uses, mapped onto the host heap ---------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct prl_emu_event* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 2; return 0; }  // two "CUs": persistent kernels walk several items per workgroup
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
struct prl_emu_event { double t; };
double prl_emu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new prl_emu_event{0}; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = prl_emu_now_ms(); return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }  // launches are synchronous here
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }
