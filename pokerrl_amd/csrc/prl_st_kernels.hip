// Per-street fused engine (prl_st.h): dispatch to the street kernels of the registered specs (prl_st_spec*.hip, prl_st_pass.inc) + the
// small glue kernels between the streets and the trunk.
#include <string>
#include <type_traits>

#include "prl_device.h"
#include "prl_kernels.h"
#include "prl_st.h"

#include "prl_st_specs.h"

const PrlFhpShapeDesc& prl_st_spec_desc(int spec) {
    static const PrlFhpShapeDesc d[PRL_ST_N_SPECS] = {prl_fhp_describe<PrlFhpDerive<PrlFhpSpec9>>(), prl_fhp_describe<PrlFhpDerive<PrlFhpSpec15>>(),
                                                      prl_fhp_describe<PrlFhpDerive<PrlFhpSpec21>>(), prl_fhp_describe<PrlFhpDerive<PrlFhpSpec27>>(),
                                                      prl_fhp_describe<PrlFhpDerive<PrlFhpSpec33>>()};
    return d[spec];
}

int prl_launch_st_down(int spec, const PrlStParams& prm, int src0, int src1, void* stream) {
    switch (spec) {
        case PRL_ST_SPEC_9: return st_spec9::launch_down(prm, src0, src1, stream);
        case PRL_ST_SPEC_15: return st_spec15::launch_down(prm, src0, src1, stream);
        case PRL_ST_SPEC_21: return st_spec21::launch_down(prm, src0, src1, stream);
        case PRL_ST_SPEC_27: return st_spec27::launch_down(prm, src0, src1, stream);
        case PRL_ST_SPEC_33: return st_spec33::launch_down(prm, src0, src1, stream);
        default: return PRL_ERR_UNSUPPORTED;
    }
}

int prl_launch_st_pass(int spec, bool last, const PrlStParams& prm, int mode, int src0, int src1, void* stream) {
    switch (spec) {
        case PRL_ST_SPEC_9: return st_spec9::launch_pass(last, prm, mode, src0, src1, stream);
        case PRL_ST_SPEC_15: return st_spec15::launch_pass(last, prm, mode, src0, src1, stream);
        case PRL_ST_SPEC_21: return st_spec21::launch_pass(last, prm, mode, src0, src1, stream);
        case PRL_ST_SPEC_27: return st_spec27::launch_pass(last, prm, mode, src0, src1, stream);
        case PRL_ST_SPEC_33: return st_spec33::launch_pass(last, prm, mode, src0, src1, stream);
        default: return PRL_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// trunk glue
// ---------------------------------------------------------------------------------------------------------------------------------
// the canonical sum over the first deal's outcomes, one row [n_leaves][n_vec][R], to where the trunk's kernels read a leaf's values:
// copy d takes vector src_vec[d] of leaf j to ev / ev_br of that leaf's node (seat dst_seat[d]) or to the "half" buffer
// [n_leaves][2][R] (slot dst_seat[d])
PRL_GLOBAL void prl_k_st_scatter_trunk(const float* __restrict__ summed, const int32_t* __restrict__ leaf_nodes, int n_leaves, int R, PrlStScatter sc,
                                       float* __restrict__ ev, float* __restrict__ ev_br, float* __restrict__ half) {
    const int total = n_leaves * sc.n_dst * R;
    for (int t = (int)(prl_bid() * prl_nthreads() + prl_tid()); t < total; t += (int)(prl_nblocks() * prl_nthreads())) {
        const int h = t % R, d = (t / R) % sc.n_dst, j = t / (R * sc.n_dst);
        const float x = summed[((size_t)j * sc.n_vec + sc.src_vec[d]) * R + h];
        const int arr = sc.dst_arr[d], seat = sc.dst_seat[d];
        if (arr == 2) half[((size_t)j * 2 + seat) * R + h] = x;
        else (arr == 0 ? ev : ev_br)[((size_t)leaf_nodes[j] * 2 + seat) * R + h] = x;
    }
}
void prl_launch_st_scatter_trunk(const float* d_summed, const int32_t* d_leaf_nodes, int n_leaves, int R, const PrlStScatter& sc, float* d_ev, float* d_ev_br,
                                 float* d_half, void* stream) {
    const int total = n_leaves * sc.n_dst * R;
    PRL_LAUNCH(prl_k_st_scatter_trunk, (total + 255) / 256, 256, 0, stream, d_summed, d_leaf_nodes, n_leaves, R, sc, d_ev, d_ev_br, d_half);
}
// seat 1's half of the current iterate (value under its new strategy, best response; left by UPDATE1_EVAL1) -> the trunk's leaves
PRL_GLOBAL void prl_k_st_half_to_trunk(const float* __restrict__ half, const int32_t* __restrict__ leaf_nodes, int n_leaves, int R, float* __restrict__ ev,
                                       float* __restrict__ ev_br) {
    const int total = n_leaves * 2 * R;
    for (int t = (int)(prl_bid() * prl_nthreads() + prl_tid()); t < total; t += (int)(prl_nblocks() * prl_nthreads())) {
        const int h = t % R, v = (t / R) % 2, j = t / (2 * R);
        (v == 0 ? ev : ev_br)[((size_t)leaf_nodes[j] * 2 + 1) * R + h] = half[t];
    }
}
void prl_launch_st_half_to_trunk(const float* d_half, const int32_t* d_leaf_nodes, int n_leaves, int R, float* d_ev, float* d_ev_br, void* stream) {
    const int total = n_leaves * 2 * R;
    PRL_LAUNCH(prl_k_st_half_to_trunk, (total + 255) / 256, 256, 0, stream, d_half, d_leaf_nodes, n_leaves, R, d_ev, d_ev_br);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// strategies / averages of one street's columns on demand (prl_solver_get(strategy); Vanilla / Linear averages from their sums)
// ---------------------------------------------------------------------------------------------------------------------------------
struct PrlStDecTable { int32_t n_dec, n_cols, nch[PRL_FHP_MAX_DEC], col0[PRL_FHP_MAX_DEC]; };
static PrlStDecTable st_dec_table(int spec) {
    const PrlFhpShapeDesc& d = prl_st_spec_desc(spec);
    PrlStDecTable t = {};
    t.n_dec = d.n_dec; t.n_cols = d.n_cols;
    for (int j = 0; j < d.n_dec; ++j) { t.nch[j] = d.dec_nch[j]; t.col0[j] = d.dec_col0[j]; }
    return t;
}
PRL_GLOBAL void prl_k_st_strategy_from_regret(PrlStParams prm, PrlStDecTable dt, double* out_cols) {
    const size_t per_inst = (size_t)dt.n_dec * prm.R, total = (size_t)prm.n_inst * per_inst;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const size_t i = t / per_inst;
        const int j = (int)((t % per_inst) / prm.R);
        const size_t h = t % prm.R;
        const int A = dt.nch[j];
        const size_t base = ((size_t)prm.col_base + i * dt.n_cols + dt.col0[j]) * (size_t)prm.R + h;
        float tt[3];
        float sum = 0.f;
        for (int a = 0; a < A; ++a) {
            const float r = prm.regret[base + (size_t)a * prm.R];
            tt[a] = prm.variant == PRL_CFR_PLUS ? r : (r > 0.f ? r : 0.f);
            sum = sum + tt[a];
        }
        const float unif = (float)(1.0 / (double)A);
        for (int a = 0; a < A; ++a) out_cols[base + (size_t)a * prm.R] = (double)(sum > 0.f ? tt[a] / sum : unif);
    }
}
PRL_GLOBAL void prl_k_st_avg_from_sum(PrlStParams prm, PrlStDecTable dt) {  // VanillaCFR.py:54-77, LinearCFR.py:53-76
    const size_t per_inst = (size_t)dt.n_dec * prm.R, total = (size_t)prm.n_inst * per_inst;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const size_t i = t / per_inst;
        const int j = (int)((t % per_inst) / prm.R);
        const size_t h = t % prm.R;
        const int A = dt.nch[j];
        const size_t base = ((size_t)prm.col_base + i * dt.n_cols + dt.col0[j]) * (size_t)prm.R + h;
        float as[3];
        for (int a = 0; a < A; ++a) as[a] = prm.avg_sum[base + (size_t)a * prm.R];
        float sum = as[0];
        for (int a = 1; a < A; ++a) sum = sum + as[a];
        for (int a = 0; a < A; ++a) prm.avg[base + (size_t)a * prm.R] = sum == 0.f ? 1.0 / (double)A : (double)(as[a] / sum);
    }
}
static inline int st_grid_for(size_t items, int block) {
    size_t g = (items + block - 1) / block;
    return (int)(g < 1 ? 1 : g > 16384 ? 16384 : g);
}
void prl_launch_st_strategy_from_regret(const PrlStParams& prm, int spec, double* out_cols, void* stream) {
    if (prm.n_inst <= 0) return;
    const PrlStDecTable dt = st_dec_table(spec);
    PRL_LAUNCH(prl_k_st_strategy_from_regret, st_grid_for((size_t)prm.n_inst * dt.n_dec * prm.R, 256), 256, 0, stream, prm, dt, out_cols);
}
void prl_launch_st_avg_from_sum(const PrlStParams& prm, int spec, void* stream) {
    if (prm.n_inst <= 0) return;
    const PrlStDecTable dt = st_dec_table(spec);
    PRL_LAUNCH(prl_k_st_avg_from_sum, st_grid_for((size_t)prm.n_inst * dt.n_dec * prm.R, 256), 256, 0, stream, prm, dt);
}
