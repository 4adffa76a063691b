// Thin device-side vocabulary used by every kernel, so that the SAME kernel source compiles
//   (a) with hipcc for gfx950 (the product), and
//   (b) with g++ -DPRL_EMU against tests/emu/prl_emu.h (a fiber-based SIMT emulator used only by the CPU test-suite).
// Kernels use 1-D grids / blocks, dynamic LDS only (16-byte aligned carve-outs, cdna guide G17) and wave64 cross-lane ops.
#pragma once
#include "prl_defs.h"

#if defined(PRL_EMU)
#include "prl_emu.h"
#define PRL_LAUNCH_BOUNDS(n)
inline void prl_atomic_add_u64(unsigned long long* p, unsigned long long v) { *p += v; }  // the emulator runs one fiber at a time
inline void prl_lds_add_i(int* p, int v) { *p += v; }
inline void prl_lds_min_u(unsigned* p, unsigned v) { if (v < *p) *p = v; }
inline int prl_atomic_add_i(int* p, int v) { int o = *p; *p += v; return o; }
inline unsigned long long prl_atomic_cas_u64(unsigned long long* p, unsigned long long expect, unsigned long long v) { const unsigned long long o = *p; if (o == expect) *p = v; return o; }
#else
#include <hip/hip_runtime.h>

#define PRL_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), __VA_ARGS__)

// 1 x grid_y workgroups: the kernel picks its work item with prl_bid_y(); prl_bid() is 0 and prl_nblocks() is 1, so device
// functions written for a grid-stride launch cover all of their items inside the one workgroup
#define PRL_LAUNCH_Y(kernel, grid_y, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(1u, (unsigned)(grid_y)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), __VA_ARGS__)
#define PRL_LAUNCH_BOUNDS(n) __launch_bounds__(n)
PRL_DEV PRL_INLINE void prl_atomic_add_u64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
PRL_DEV PRL_INLINE void prl_lds_add_i(int* p, int v) { atomicAdd(p, v); }  // integer add on an LDS word (order-free)
PRL_DEV PRL_INLINE void prl_lds_min_u(unsigned* p, unsigned v) { atomicMin(p, v); }  // unsigned minimum on an LDS word (order-free)
PRL_DEV PRL_INLINE int prl_atomic_add_i(int* p, int v) { return atomicAdd(p, v); }  // returns the value before the add
PRL_DEV PRL_INLINE unsigned long long prl_atomic_cas_u64(unsigned long long* p, unsigned long long expect, unsigned long long v) { return atomicCAS(p, expect, v); }  // returns the value before
PRL_DEV PRL_INLINE unsigned prl_tid() { return threadIdx.x; }
PRL_DEV PRL_INLINE unsigned prl_bid() { return blockIdx.x; }
PRL_DEV PRL_INLINE unsigned prl_bid_y() { return blockIdx.y; }
PRL_DEV PRL_INLINE unsigned prl_nthreads() { return blockDim.x; }
PRL_DEV PRL_INLINE unsigned prl_nblocks() { return gridDim.x; }
PRL_DEV PRL_INLINE unsigned prl_lane() { return threadIdx.x & 63u; }
PRL_DEV PRL_INLINE void prl_sync() { __syncthreads(); }
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter (loads, stores
// and LDS DMA share vmcnt on gfx9), which would serialise the asynchronous prefetch below with every phase boundary.
PRL_DEV PRL_INLINE void prl_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// Asynchronous global -> LDS copy (gfx950 LDS DMA, no VGPR round trip), 16 bytes or 4 bytes per lane:
// LDS[lds_wave_base + SIZE * lane ...] <- *(gbase + byte_off). gbase and lds_wave_base are wave-uniform; the 16-byte form
// needs 16-byte aligned addresses on both sides. Not tracked by the compiler's wait-count insertion: the consumer calls
// prl_dma_wait() (then a barrier) before reading the destination. The vector-memory address unit handles 64 lanes in
// ~16 clocks per instruction whatever the width, so the 16-byte form moves 4x the data per issue slot.
PRL_DEV PRL_INLINE void prl_lds_dma_x4(const void* gbase, uint32_t byte_off, void* lds_wave_base) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "s"(la), "v"(byte_off), "s"(gbase) : "memory");
}
// the same with the LDS destination as a 32-bit LDS address (prl_lds_addr): scalar arithmetic on it stays scalar, and there is no
// generic-to-LDS pointer conversion (with its null test) per instruction
PRL_DEV PRL_INLINE uint32_t prl_lds_addr(const void* lds_ptr) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) const char*)lds_ptr);
}
PRL_DEV PRL_INLINE void prl_lds_dma_x4_a(const void* gbase, uint32_t byte_off, uint32_t lds_wave_addr) {
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "s"(lds_wave_addr), "v"(byte_off), "s"(gbase) : "memory");
}
PRL_DEV PRL_INLINE void prl_lds_dma_dword(const void* gbase, uint32_t byte_off, void* lds_wave_base) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "s"(la), "v"(byte_off), "s"(gbase) : "memory");
}
PRL_DEV PRL_INLINE void prl_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
PRL_DEV PRL_INLINE int prl_opaque_scalar(int v) { asm volatile("" : "+s"(v)); return v; }  // hides a wave-uniform value's history from the optimiser
// the same for a per-lane value: what is computed from the result cannot be hoisted out of a loop or shared with other uses (the compiler
// otherwise precomputes every loop-invariant LDS address of the pass before the loop and SPILLS them: 24 scratch reloads per instance, each
// with a full s_waitcnt vmcnt(0), on the 27-node shape; profiles/r05_experiments.txt)
PRL_DEV PRL_INLINE int prl_opaque_lane(int v) { asm volatile("" : "+v"(v)); return v; }
// a USE of a loaded value, in the compiler's eyes: its wait for the load is placed here and not where the value is used next. (Values loaded before a
// loop and used at the top of its body otherwise count as pending at the loop header, and the wait inserted there is sized for the entry path: it then
// also waits for everything the previous iteration left in flight -- on gfx9 that includes its stores.)
PRL_DEV PRL_INLINE void prl_use(float& x) { asm volatile("" : "+v"(x)); }
PRL_DEV PRL_INLINE int prl_wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }  // v is the same in every lane: keep it scalar
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
// DPP cross-lane moves (gfx9 data-parallel primitives: VALU latency instead of an LDS-crossbar round trip).
// Lanes without a valid source receive 0.
template <int D>
PRL_DEV PRL_INLINE float prl_dpp_row_shr(float v) {  // lane l <- lane l - D inside its row of 16 lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xF, 0xF, false));
}
template <int N>
PRL_DEV PRL_INLINE float prl_dpp_row_share(float v) {  // every lane <- lane N of its own row of 16 (gfx90a+: row_newbcast / row_share): VALU, no LDS crossbar
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + N, 0xF, 0xF, false));
}
PRL_DEV PRL_INLINE float prl_dpp_row_bcast15(float v) {  // rows 1 and 3 <- lane 15 of the previous row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
}
PRL_DEV PRL_INLINE float prl_dpp_row_bcast31(float v) {  // lanes 32..63 <- lane 31
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
}
PRL_DEV PRL_INLINE int prl_dpp_wave_shr1_i(int v, int fill) {  // lane l <- lane l - 1, lane 0 <- fill
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xF, 0xF, false);
}
PRL_DEV PRL_INLINE float prl_readlane(float v, int lane) {  // wave-uniform lane
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
#endif

// Canonical wave-64 inclusive scan (DESIGN.md "summation order"): Hillis-Steele inside each row of 16 lanes
// (d = 1, 2, 4, 8), then rows 1 and 3 add lane 15 of the row below, then lanes 32..63 add lane 31.
// Six DPP adds on gfx950; the oracle (oracle/prl_oracle.c: scan64) replays exactly this association.
PRL_DEV PRL_INLINE float prl_wave_scan_canonical(float v) {
    v = v + prl_dpp_row_shr<1>(v);
    v = v + prl_dpp_row_shr<2>(v);
    v = v + prl_dpp_row_shr<4>(v);
    v = v + prl_dpp_row_shr<8>(v);
#if defined(PRL_EMU)
    v = v + prl_dpp_row_bcast15(v);
    v = v + prl_dpp_row_bcast31(v);
#else
    // The two masked steps as ONE instruction each: the add itself carries the DPP modifier and leaves the rows outside
    // row_mask untouched (v instead of v + 0.0: the same bits for every value but -0.0, which a sum of reach
    // probabilities never is). The compiler only fuses a mov_dpp into a float add when all rows are enabled and would
    // emit mov 0 / mov_dpp / add here. "s_nop 1": a DPP operand written by the preceding VALU instruction needs two wait
    // states, which the hazard recogniser does not insert inside inline assembly.
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
#endif
    return v;
}

// Canonical per-card list scan, "row16 order" (DESIGN.md "summation order"): a list of n <= 64 values, zero-padded to 16 * E
// entries with E = ceil(n / 16); lane i of a ROW of 16 lanes owns entries i*E .. i*E + E-1:
//   local inclusive prefix l_k = l_(k-1) + x_(i*E+k) (l_0 = x_(i*E)), lane total t_i = l_(E-1);
//   Hillis-Steele inclusive scan of t over the 16 lanes of the row (d = 1, 2, 4, 8: t_i += t_(i-d), lanes i < d add 0);
//   carry_i = the scanned t_(i-1) (0 for lane 0);  Q_incl[i*E + k] = carry_i + l_k.
// One wave scans four lists at once, and on a 5-card board (46 entries, E = 3) twelve waves scan the lists of all 47 live
// cards in one go. The oracle (oracle/prl_oracle.c: card_scan) replays exactly this association.
PRL_DEV PRL_INLINE float prl_row16_scan(float t) {  // inclusive, every lane of the wave must call it
    t = t + prl_dpp_row_shr<1>(t);
    t = t + prl_dpp_row_shr<2>(t);
    t = t + prl_dpp_row_shr<4>(t);
    t = t + prl_dpp_row_shr<8>(t);
    return t;
}

// N independent canonical scans at once. Issued one by one, every masked step above pays its own "s_nop 1"; side by side the
// steps of the other vectors ARE the wait states, so a group of 8 or 9 vectors needs a single s_nop (an s_nop costs the
// wave an issue turn like any other instruction, and this kernel family is bound by issue turns per wave).
#if !defined(PRL_EMU)
#define PRL_B15(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define PRL_B31(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
PRL_DEV PRL_INLINE void prl_scan_tail8(float* v) {
    asm("s_nop 1\n\t" PRL_B15(0) PRL_B15(1) PRL_B15(2) PRL_B15(3) PRL_B15(4) PRL_B15(5) PRL_B15(6) PRL_B15(7)
        PRL_B31(0) PRL_B31(1) PRL_B31(2) PRL_B31(3) PRL_B31(4) PRL_B31(5) PRL_B31(6) PRL_B31(7)
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
PRL_DEV PRL_INLINE void prl_scan_tail9(float* v) {
    asm("s_nop 1\n\t" PRL_B15(0) PRL_B15(1) PRL_B15(2) PRL_B15(3) PRL_B15(4) PRL_B15(5) PRL_B15(6) PRL_B15(7) PRL_B15(8)
        PRL_B31(0) PRL_B31(1) PRL_B31(2) PRL_B31(3) PRL_B31(4) PRL_B31(5) PRL_B31(6) PRL_B31(7) PRL_B31(8)
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
}
#undef PRL_B15
#undef PRL_B31
#endif
template <int N>
PRL_DEV PRL_INLINE void prl_wave_scan_canonical_n(float (&v)[N]) {
#if defined(PRL_EMU)
    for (int i = 0; i < N; ++i) v[i] = prl_wave_scan_canonical(v[i]);
#else
    static_assert(N == 9 || N == 17, "group sizes are spelled out for the two callers");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float x = v[i];
        x = x + prl_dpp_row_shr<1>(x);
        x = x + prl_dpp_row_shr<2>(x);
        x = x + prl_dpp_row_shr<4>(x);
        x = x + prl_dpp_row_shr<8>(x);
        v[i] = x;
    }
    prl_scan_tail9(v);
    if constexpr (N == 17) prl_scan_tail8(v + 9);
#endif
}

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
