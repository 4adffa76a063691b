// Fused board-block kernels ("engine F") for the Flop5Holdem public tree: CFR / CFR+ / Linear CFR on 1326-hand ranges with every per-node
// vector of a board subtree kept ON CHIP.
//
// Why: the reference materialises reach / ev / ev_br for every node ([2][R] float32 each, nodes.py:33-36). On the FHP tree
// that is ~3.75 MB of transient traffic per board and iteration against ~0.4 MB of persistent state. Below
// each chance outcome the board subtree is independent (ValueFiller.py:76-78 is a plain sum over boards), its shape is the
// same for every board (betting never looks at the cards) and per hand everything except terminal equity is hand-local.
// So: one workgroup walks ONE board subtree depth-first with the subtree shape known at COMPILE time (prl_fhp_shape.h):
//   * 768 lanes; 541 of them x 2 adjacent RANK-SORTED POSITIONS cover the 1081 live hands of a board (sorted storage, prl_fhp.h);
//     reach / ev / regrets of the current DFS path live in VGPRs;
//   * HBM traffic per board and pass: 14 regret columns + ~15 KB of showdown plan in (prefetched into LDS by LDS-DMA while
//     the previous board is walked), the updated seat's 7 regret columns and float64 average columns read-modify-written,
//     one row of 1-4 root vectors out -- nothing else;
//   * the only cross-hand step, terminal equity, goes through LDS in the rank-sorted domain: six-op DPP wave scans over
//     the sorted range and over the 47 per-card blocker lists, the canonical order of DESIGN.md, so the result is
//     bit-identical to engine G (prl_tree_kernels.hip) and to the CPU oracle;
//   * the chance-node sum over boards is done afterwards in the canonical nested order (blocks of 32, groups of 32).
// No MFMA anywhere: nothing here is a contraction; measured, the kernel is bound by VALU issue.
#include <type_traits>

#include "prl_device.h"
#include "prl_fhp.h"
#include "prl_kernels.h"


// ---------------------------------------------------------------------------------------------------------------------
// launch configuration of the board-pass kernel: 768 lanes = 12 waves per CU, 3 per SIMD. A lane owns two ADJACENT hands (663
// lanes hold the 1326 hands), so every global access of the hand domain is one 8- or 16-byte vector instruction (the address
// unit takes ~16 clocks per wave instruction whatever its width); in the per-card scans a lane owns 3 list entries of one of
// the 48 card slots (4 per wave), so the 47 live cards of a board are scanned in one round.
// ---------------------------------------------------------------------------------------------------------------------
#if defined(PRL_EMU)
#define FHP_LB(t, w)
#else
#define FHP_LB(t, w) __launch_bounds__(t, w)
#endif

#define FHP_THREADS 768
#define FHP_SLOTS 2
#define FHP_LAUNCH_BOUNDS FHP_LB(768, 3)
#define FHP_SPEC PrlFhpSpec15
namespace fhp_shape15 {
#include "prl_fhp_pass.inc"
}
#undef FHP_SPEC
#define FHP_SPEC PrlFhpSpec9
namespace fhp_shape9 {
#include "prl_fhp_pass.inc"
}
#undef FHP_SPEC
#if !defined(PRL_FHP_NO_SHAPE21)
#define FHP_SPEC PrlFhpSpec21
namespace fhp_shape21 {
#include "prl_fhp_pass.inc"
}
#undef FHP_SPEC
#endif
#undef FHP_THREADS
#undef FHP_SLOTS
#undef FHP_LAUNCH_BOUNDS

const PrlFhpShapeDesc& prl_fhp_shape_desc(int shape_id) {
    static const PrlFhpShapeDesc d[PRL_FHP_N_SHAPES] = {prl_fhp_describe<PrlFhpDerive<PrlFhpSpec15>>(), prl_fhp_describe<PrlFhpDerive<PrlFhpSpec9>>(),
                                                        prl_fhp_describe<PrlFhpDerive<PrlFhpSpec21>>()};
    return d[shape_id];
}

bool prl_fhp_shape_compiled(int shape_id) {
#if !defined(PRL_FHP_NO_SHAPE21)
    return shape_id >= 0 && shape_id < PRL_FHP_N_SHAPES;
#else
    return shape_id == PRL_FHP_SHAPE_15 || shape_id == PRL_FHP_SHAPE_9;
#endif
}

// suit symmetrisation (prl_fhp.h): a thread per (vector, hand); the orbit sum runs over the class's hands in ascending hand index
PRL_GLOBAL void prl_k_fhp_symmetrize(const float* in, int n_vec, int R, const int32_t* class_of, const int32_t* class_start, const int32_t* class_hands, float* out) {
    const int i = (int)(prl_bid() * prl_nthreads() + prl_tid());
    if (i >= n_vec * R) return;
    const int v = i / R, h = i - v * R, k = class_of[h];
    const int a = class_start[k], b = class_start[k + 1];
    float s = in[(size_t)v * R + class_hands[a]];
    for (int j = a + 1; j < b; ++j) s = s + in[(size_t)v * R + class_hands[j]];
    out[i] = s / (float)(b - a);
}
void prl_launch_fhp_symmetrize(const float* in, int n_vec, int R, const int32_t* class_of, const int32_t* class_start, const int32_t* class_hands, float* out, void* stream) {
    const int n = n_vec * R;
    PRL_LAUNCH(prl_k_fhp_symmetrize, (n + 255) / 256, 256, 0, stream, in, n_vec, R, class_of, class_start, class_hands, out);
}

int prl_launch_fhp_pass(const PrlFhpParams& prm, int mode, int src0, int src1, void* stream) {
    switch (prm.shape) {
        case PRL_FHP_SHAPE_15: return fhp_shape15::launch_pass(prm, mode, src0, src1, stream);
        case PRL_FHP_SHAPE_9: return fhp_shape9::launch_pass(prm, mode, src0, src1, stream);
#if !defined(PRL_FHP_NO_SHAPE21)
        case PRL_FHP_SHAPE_21: return fhp_shape21::launch_pass(prm, mode, src0, src1, stream);
#endif
        default: return PRL_ERR_UNSUPPORTED;
    }
}

// materialise the strategy implied by the regrets (tests / prl_solver_get): board region -> board region, float64
PRL_GLOBAL void prl_k_fhp_strategy_from_regret(PrlFhpParams prm, double* out_region) {
    const size_t per_board = (size_t)prm.n_dec * prm.np;
    const size_t total = (size_t)prm.n_boards * per_board;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const size_t b = t / per_board;
        const int j = (int)((t % per_board) / prm.np);
        const size_t q = t % prm.np;
        const int A = prm.dec_nch[j], col0 = prm.dec_col0[j];
        const size_t base = (b * prm.n_cols_board + col0) * (size_t)prm.np + q;
        float tt[3];
        float sum = 0.f;
        for (int i = 0; i < A; ++i) {
            float r = prm.regret[base + (size_t)i * prm.np];
            tt[i] = prm.variant == PRL_CFR_PLUS ? r : (r > 0.f ? r : 0.f);
            sum = sum + tt[i];
        }
        const float unif = (float)(1.0 / (double)A);
        for (int i = 0; i < A; ++i) out_region[base + (size_t)i * prm.np] = (double)(sum > 0.f ? tt[i] / sum : unif);
    }
}

// Vanilla / Linear CFR: the average strategy of the board columns from the running sums, avg = avg_sum / sum_a avg_sum or
// uniform (VanillaCFR.py:54-77, LinearCFR.py:53-76: float32 division, stored in the float64 column array). The board pass
// only maintains avg_sum; this runs when the average is read (evaluation, prl_solver_get, checkpoint).
PRL_GLOBAL void prl_k_fhp_avg_from_sum(PrlFhpParams prm) {
    const size_t per_board = (size_t)prm.n_dec * prm.np;
    const size_t total = (size_t)prm.n_boards * per_board;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const size_t b = t / per_board;
        const int j = (int)((t % per_board) / prm.np);
        const size_t q = t % prm.np;
        const int A = prm.dec_nch[j], col0 = prm.dec_col0[j];
        const size_t base = (b * prm.n_cols_board + col0) * (size_t)prm.np + q;
        float as[3];
        for (int i = 0; i < A; ++i) as[i] = prm.avg_sum[base + (size_t)i * prm.np];
        float sum = as[0];
        for (int i = 1; i < A; ++i) sum = sum + as[i];
        for (int i = 0; i < A; ++i) prm.avg[base + (size_t)i * prm.np] = sum == 0.f ? 1.0 / (double)A : (double)(as[i] / sum);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// sorted storage <-> the caller's hand-order columns (prl_fhp.h). Not on the iteration path: prl_solver_get / set / checkpoints.
// A thread moves one (board, column, position); the permutation is the plan's `sh` (position -> hand).
// ---------------------------------------------------------------------------------------------------------------------
template <class E>
PRL_GLOBAL void prl_k_fhp_expand(PrlFhpParams prm, const E* region, int b0, int nb, const E* fill_by_col, const E* blocked_src, E* dst) {
    const int ncb = prm.n_cols_board, R = prm.R, NL = R - PRL_FHP_NBLOCKED;
    const size_t total = (size_t)nb * ncb * R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int q = (int)(t % R);
        const size_t bj = t / R;
        const int j = (int)(bj % ncb);
        const size_t b = (size_t)b0 + bj / ncb;
        const int h = prm.plan_pp[b * PRL_PP_STRIDE + q];
        E v;
        if (q < NL) v = region[(b * ncb + j) * (size_t)prm.np + q];
        else if (blocked_src) v = blocked_src[(b * ncb + j) * (size_t)PRL_FHP_NBLOCKED + (q - NL)];
        else v = fill_by_col ? fill_by_col[j] : (E)0;
        dst[bj * R + h] = v;
    }
}
template <class E>
PRL_GLOBAL void prl_k_fhp_compact(PrlFhpParams prm, const E* src, int b0, int nb, E* region, E* blocked_dst) {
    const int ncb = prm.n_cols_board, R = prm.R, NL = R - PRL_FHP_NBLOCKED, NPX = prm.np + PRL_FHP_NBLOCKED;
    const size_t total = (size_t)nb * ncb * NPX;  // np region elements (live positions + zero padding), then the blocked hands
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int q = (int)(t % NPX);
        const size_t bj = t / NPX;
        const int j = (int)(bj % ncb);
        const size_t b = (size_t)b0 + bj / ncb;
        const int16_t* sh = prm.plan_pp + b * PRL_PP_STRIDE;
        if (q < prm.np) region[(b * ncb + j) * (size_t)prm.np + q] = q < NL ? src[bj * R + sh[q]] : (E)0;
        else if (blocked_dst) blocked_dst[(b * ncb + j) * (size_t)PRL_FHP_NBLOCKED + (q - prm.np)] = src[bj * R + sh[NL + (q - prm.np)]];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// canonical chance sum over the per-board root values: blocks of 32 boards, groups of 32 blocks, then the groups
// level 0: in = per-board [n][2][R] -> out = per-block; level 1: per-block -> per-group; level 2: per-group -> dest [2][R]
// ---------------------------------------------------------------------------------------------------------------------
struct alignas(8) PrlF2 { float x, y; };
// two adjacent elements per lane (R2 is a multiple of 1326, hence even; rows are 8-byte aligned): half the load instructions
// of this HBM-bound kernel, the same running adds per element
PRL_GLOBAL void prl_k_fhp_sum_level(const float* __restrict__ in, int n_in, int fan, int R2, float* __restrict__ out) {
    const int n_out = (n_in + fan - 1) / fan;
    const int R2h = R2 / 2;
    const size_t total = (size_t)n_out * R2h;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int o = (int)(t / R2h);
        const int x = 2 * (int)(t % R2h);
        const int lo = o * fan, hi = lo + fan < n_in ? lo + fan : n_in;
        PrlF2 s = *(const PrlF2*)(in + (size_t)lo * R2 + x);
        for (int i = lo + 1; i < hi; ++i) {
            const PrlF2 v = *(const PrlF2*)(in + (size_t)i * R2 + x);
            s.x = s.x + v.x;
            s.y = s.y + v.y;
        }
        *(PrlF2*)(out + (size_t)o * R2 + x) = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
static inline int fhp_grid_for(size_t items, int block) {
    size_t g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 16384) g = 16384;
    return (int)g;
}

void prl_launch_fhp_strategy_from_regret(const PrlFhpParams& prm, double* out_region, void* stream) {
    if (prm.n_boards <= 0) return;
    size_t items = (size_t)prm.n_boards * prm.n_dec * prm.np;
    PRL_LAUNCH(prl_k_fhp_strategy_from_regret, fhp_grid_for(items, 256), 256, 0, stream, prm, out_region);
}

void prl_launch_fhp_avg_from_sum(const PrlFhpParams& prm, void* stream) {
    if (prm.n_boards <= 0) return;
    size_t items = (size_t)prm.n_boards * prm.n_dec * prm.np;
    PRL_LAUNCH(prl_k_fhp_avg_from_sum, fhp_grid_for(items, 256), 256, 0, stream, prm);
}

void prl_launch_fhp_expand(const PrlFhpParams& prm, const void* region, int elem, int b0, int nb, const void* fill_by_col, const void* blocked_src,
                           void* dst, void* stream) {
    if (nb <= 0) return;
    const size_t items = (size_t)nb * prm.n_cols_board * prm.R;
    if (elem == 4) PRL_LAUNCH(prl_k_fhp_expand<float>, fhp_grid_for(items, 256), 256, 0, stream, prm, (const float*)region, b0, nb, (const float*)fill_by_col, (const float*)blocked_src, (float*)dst);
    else PRL_LAUNCH(prl_k_fhp_expand<double>, fhp_grid_for(items, 256), 256, 0, stream, prm, (const double*)region, b0, nb, (const double*)fill_by_col, (const double*)blocked_src, (double*)dst);
}

void prl_launch_fhp_compact(const PrlFhpParams& prm, const void* src, int elem, int b0, int nb, void* region, void* blocked_dst, void* stream) {
    if (nb <= 0) return;
    const size_t items = (size_t)nb * prm.n_cols_board * (prm.np + PRL_FHP_NBLOCKED);
    if (elem == 4) PRL_LAUNCH(prl_k_fhp_compact<float>, fhp_grid_for(items, 256), 256, 0, stream, prm, (const float*)src, b0, nb, (float*)region, (float*)blocked_dst);
    else PRL_LAUNCH(prl_k_fhp_compact<double>, fhp_grid_for(items, 256), 256, 0, stream, prm, (const double*)src, b0, nb, (double*)region, (double*)blocked_dst);
}

// per-board [n_boards][2][R] -> dest [2][R] in the canonical nested order; scratch >= (ceil(n/32) + ceil(n/1024)) * 2R floats
void prl_launch_fhp_chance_sum(const float* d_board_vals, int n_boards, int W, float* d_scratch, float* d_dest, void* stream) {
    prl_launch_fhp_chance_finish(d_board_vals, n_boards, 0, W, d_scratch, d_dest, stream);
}

// Sharded solve: a rank reduces its own boards up to `level` (0: nothing, the per-board values themselves; 1: blocks of
// 32 boards; 2: groups of 32 blocks) -- the units are whole canonical units because the shard size is a multiple of the
// unit size -- and after the all-gather every rank finishes the remaining levels over ALL units in global order.
int prl_fhp_units_at_level(int n_boards, int level) {
    int n = n_boards;
    for (int l = 0; l < level; ++l) n = (n + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
    return n;
}

void prl_launch_fhp_chance_partial(const float* d_board_vals, int n_boards, int level, int W, float* d_scratch, float* d_units, void* stream) {
    const int R2 = W;
    if (level == 0) {
        PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)n_boards * R2 / 2, 256), 256, 0, stream, d_board_vals, n_boards, 1, R2, d_units);
        return;
    }
    const int n_blk = (n_boards + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
    float* blk = level == 1 ? d_units : d_scratch;
    PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)n_blk * R2 / 2, 256), 256, 0, stream, d_board_vals, n_boards, PRL_CHANCE_BLOCK, R2, blk);
    if (level == 1) return;
    const int n_grp = (n_blk + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
    PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)n_grp * R2 / 2, 256), 256, 0, stream, (const float*)blk, n_blk, PRL_CHANCE_BLOCK, R2, d_units);
}

// the same when the board pass already summed its boards per block (PrlFhpParams::block_sum): d_blocks = level-1 units
void prl_launch_fhp_chance_partial_from_blocks(const float* d_blocks, int n_blk, int level, int W, float* d_units, void* stream) {
    if (level <= 1) {  // the blocks are the units
        PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)n_blk * W / 2, 256), 256, 0, stream, d_blocks, n_blk, 1, W, d_units);
        return;
    }
    const int n_grp = (n_blk + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
    PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)n_grp * W / 2, 256), 256, 0, stream, d_blocks, n_blk, PRL_CHANCE_BLOCK, W, d_units);
}

// units of `level` (contiguous [n_units][2][R]) -> dest [2][R]; scratch >= (ceil(n/32) + ceil(n/1024) + 1) * 2R floats
void prl_launch_fhp_chance_finish(const float* d_units, int n_units, int level, int W, float* d_scratch, float* d_dest, void* stream) {
    const int R2 = W;
    const float* cur = d_units;
    int n = n_units;
    float* next = d_scratch;
    for (int l = level; l < 2; ++l) {
        if (n <= PRL_CHANCE_BLOCK) break;  // one block: the remaining levels would copy its sum (a multi-street tree with a handful of flops)
        const int n_out = (n + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
        PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)n_out * R2, 256), 256, 0, stream, cur, n, PRL_CHANCE_BLOCK, R2, next);
        cur = next;
        next += (size_t)n_out * R2;
        n = n_out;
    }
    PRL_LAUNCH(prl_k_fhp_sum_level, fhp_grid_for((size_t)R2, 256), 256, 0, stream, cur, n, n > 0 ? n : 1, R2, d_dest);
}

// all-gather layout [world][n_which][n_units][W] -> [n_which][world * n_units][W] (global unit order)
PRL_GLOBAL void prl_k_fhp_compact_gathered(const float* __restrict__ in, int world, int n_which, int n_units, int R2, float* __restrict__ out) {
    const size_t total = (size_t)world * n_which * n_units * R2;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int x = (int)(t % R2);
        size_t q = t / R2;
        const int u = (int)(q % n_units); q /= n_units;
        const int w = (int)(q % n_which);
        const int r = (int)(q / n_which);
        out[(((size_t)w * world + r) * n_units + u) * R2 + x] = in[t];
    }
}
void prl_launch_fhp_compact_gathered(const float* d_in, int world, int n_which, int n_units, int W, float* d_out, void* stream) {
    PRL_LAUNCH(prl_k_fhp_compact_gathered, fhp_grid_for((size_t)world * n_which * n_units * W, 256), 256, 0, stream, d_in, world, n_which, n_units, W, d_out);
}
