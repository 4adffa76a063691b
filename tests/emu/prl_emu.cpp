// Fiber-based SIMT emulator, see prl_emu.h. TEST INFRASTRUCTURE ONLY.
#include "prl_emu.h"

#include <stdio.h>
#include <time.h>
#include <ucontext.h>

#include <vector>

namespace prl_emu {

thread_local Ctx* g_ctx = nullptr;

namespace {
enum State { RUNNABLE, AT_BARRIER, AT_WAVE, DONE };

struct Fiber {
    ucontext_t uc;
    Ctx ctx;
    State state;
    uint64_t wave_in;
    int wave_src;
    uint64_t wave_out;
    int wave_kind;  // 0 exchange, 1 ballot
};

constexpr size_t STACK_BYTES = 96 * 1024;
thread_local std::vector<char> g_stacks;
thread_local std::vector<Fiber> g_fibers;
thread_local ucontext_t g_sched;
thread_local const std::function<void()>* g_body = nullptr;
thread_local Fiber* g_cur = nullptr;

void trampoline() {
    (*g_body)();
    g_cur->state = DONE;
    swapcontext(&g_cur->uc, &g_sched);
}

void yield_to_scheduler() {
    Fiber* me = g_cur;
    swapcontext(&me->uc, &g_sched);
    g_cur = me;
    g_ctx = &me->ctx;
}
}  // namespace

void block_barrier() {
    g_cur->state = AT_BARRIER;
    yield_to_scheduler();
}

uint64_t wave_exchange(uint64_t v, int src_lane) {
    g_cur->wave_in = v;
    g_cur->wave_src = src_lane;
    g_cur->wave_kind = 0;
    g_cur->state = AT_WAVE;
    yield_to_scheduler();
    return g_cur->wave_out;
}

uint64_t wave_ballot(int pred) {
    g_cur->wave_in = pred ? 1 : 0;
    g_cur->wave_kind = 1;
    g_cur->state = AT_WAVE;
    yield_to_scheduler();
    return g_cur->wave_out;
}

void launch(const std::function<void()>& body, unsigned grid, unsigned block, size_t smem_bytes) { launch2(body, grid, 1, block, smem_bytes); }

// grid_x x grid_y workgroups; kernels that index their work with blockIdx.y see blockIdx.x = 0 / gridDim.x = 1 when grid_x = 1
void launch2(const std::function<void()>& body, unsigned grid, unsigned grid_y, unsigned block, size_t smem_bytes) {
    if (block == 0 || grid == 0 || grid_y == 0) return;
    if (block > 1024) {
        fprintf(stderr, "prl_emu: block size %u > 1024\n", block);
        abort();
    }
    if (g_stacks.size() < (size_t)block * STACK_BYTES) g_stacks.resize((size_t)block * STACK_BYTES);
    g_fibers.resize(block);
    std::vector<char> smem(smem_bytes + 64);
    g_body = &body;
    for (unsigned by = 0; by < grid_y; ++by)
    for (unsigned b = 0; b < grid; ++b) {
        memset(smem.data(), 0xCD, smem.size());  // LDS is uninitialised on real hardware: poison it
        char* smem_base = (char*)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
        for (unsigned t = 0; t < block; ++t) {
            Fiber& f = g_fibers[t];
            f.ctx = Ctx{t, b, block, grid, smem_base, by};
            f.state = RUNNABLE;
            getcontext(&f.uc);
            f.uc.uc_stack.ss_sp = g_stacks.data() + (size_t)t * STACK_BYTES;
            f.uc.uc_stack.ss_size = STACK_BYTES;
            f.uc.uc_link = &g_sched;
            makecontext(&f.uc, trampoline, 0);
        }
        unsigned n_done = 0;
        while (n_done < block) {
            bool progress = false;
            for (unsigned t = 0; t < block; ++t) {
                Fiber& f = g_fibers[t];
                if (f.state != RUNNABLE) continue;
                g_cur = &f;
                g_ctx = &f.ctx;
                swapcontext(&g_sched, &f.uc);
                progress = true;
                if (f.state == DONE) n_done++;
            }
            // wave rendezvous
            for (unsigned w0 = 0; w0 < block; w0 += 64) {
                unsigned w1 = w0 + 64 < block ? w0 + 64 : block;
                bool all = true, any = false;
                for (unsigned t = w0; t < w1; ++t) {
                    if (g_fibers[t].state == DONE) continue;
                    any = true;
                    if (g_fibers[t].state != AT_WAVE) all = false;
                }
                if (!any || !all) continue;
                uint64_t ballot = 0;
                for (unsigned t = w0; t < w1; ++t)
                    if (g_fibers[t].state == AT_WAVE && g_fibers[t].wave_kind == 1 && g_fibers[t].wave_in) ballot |= 1ull << (t - w0);
                for (unsigned t = w0; t < w1; ++t) {
                    Fiber& f = g_fibers[t];
                    if (f.state != AT_WAVE) continue;
                    if (f.wave_kind == 1) f.wave_out = ballot;
                    else {
                        unsigned s = w0 + (unsigned)f.wave_src;
                        f.wave_out = (s < w1 && g_fibers[s].state == AT_WAVE) ? g_fibers[s].wave_in : 0;
                    }
                }
                for (unsigned t = w0; t < w1; ++t)
                    if (g_fibers[t].state == AT_WAVE) g_fibers[t].state = RUNNABLE;
                progress = true;
            }
            // block barrier: released when every live fiber waits at it
            {
                bool all = true, any = false;
                for (unsigned t = 0; t < block; ++t) {
                    if (g_fibers[t].state == DONE) continue;
                    any = true;
                    if (g_fibers[t].state != AT_BARRIER) all = false;
                }
                if (any && all) {
                    for (unsigned t = 0; t < block; ++t)
                        if (g_fibers[t].state == AT_BARRIER) g_fibers[t].state = RUNNABLE;
                    progress = true;
                }
            }
            if (!progress) {
                fprintf(stderr, "prl_emu: deadlock in block %u (divergent barrier or wave op?)\n", b);
                abort();
            }
        }
    }
    g_body = nullptr;
    g_ctx = nullptr;
    g_cur = nullptr;
}

}  // namespace prl_emu

double prl_emu_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
