// prl_solver_*: device-resident tabular CFR / best-response solver behind the C ABI (include/pokerrl_hip.h section 5).
// Host orchestration only: uploads the flat tree, owns the HIP stream and the HBM arrays, issues the kernels in the
// reference's order (_CFRBase.py:110-134). Nothing here computes on the CPU.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "prl_cards.h"
#include "prl_device.h"
#include "prl_host.h"
#include "prl_kernels.h"
#include "prl_rt.h"
#include "prl_solver_types.h"

struct prl_solver {
    PrlFlatTree ft;  // host copy (levels, lists)
    PrlDevTree T{};
    PrlDevState S{};
    PrlDevState Seval{};      // scratch per-node vectors for the average-strategy evaluation (allocated lazily)
    bool eval_ready = false;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    int32_t* d_term_nodes = nullptr;
    int n_term = 0;
    int32_t* d_nodes_p[2] = {nullptr, nullptr};
    int n_nodes_p[2] = {0, 0};
    int32_t* d_col_node = nullptr;
    int variant = PRL_CFR_PLUS, delay = 0, iter = 0;
    bool ev_valid = false;    // S.ev / S.ev_br / S.expl correspond to the current strategy + reach
    bool keep_br_idx = false;
    float* d_expl_hist = nullptr;  // [cap][2] current-strategy exploitability after every iteration
    int hist_cap = 0;
    size_t bytes_allocated = 0;
};

namespace {

template <class T>
int dev_alloc(prl_solver* s, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        prl_set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        return PRL_ERR_OOM;
    }
    s->allocs.push_back(q);
    s->bytes_allocated += bytes;
    *p = (T*)q;
    return PRL_OK;
}

template <class T>
int dev_upload(prl_solver* s, const T** p, const std::vector<T>& v) {
    T* q = nullptr;
    int e = dev_alloc(s, &q, v.size());
    if (e) return e;
    if (!v.empty()) PRL_HIP_TRY(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return PRL_OK;
}

#define TRY(x) do { int e_ = (x); if (e_) return e_; } while (0)

int alloc_node_vectors(prl_solver* s, PrlDevState* st, bool with_br_idx) {
    const size_t nv = (size_t)s->T.n_nodes * 2 * s->T.R;
    TRY(dev_alloc(s, &st->reach, nv));
    TRY(dev_alloc(s, &st->ev, nv));
    TRY(dev_alloc(s, &st->ev_br, nv));
    TRY(dev_alloc(s, &st->expl, 2));
    st->br_idx = nullptr;
    if (with_br_idx) {
        TRY(dev_alloc(s, &st->br_idx, (size_t)s->T.n_nodes * s->T.R));
        PRL_HIP_TRY(hipMemsetAsync(st->br_idx, 0, (size_t)s->T.n_nodes * s->T.R * sizeof(int32_t), s->stream));  // terminal / chance rows stay 0
    }
    PRL_HIP_TRY(hipMemsetAsync(st->reach, 0, nv * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(st->ev, 0, nv * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(st->ev_br, 0, nv * sizeof(float), s->stream));
    return PRL_OK;
}

// generalised chance / equity constants, computed in float64 then rounded, as NumPy does
float chance_prob_f32(int n_children, int n_cards, int n_hole, int n_dealt) {
    double denom = (double)n_children * (double)prl_comb(n_cards - 2 * n_hole, n_dealt) / (double)prl_comb(n_cards, n_dealt);
    return (float)(1.0 / denom);
}
float eq_const_f32(int n_cards, int n_hole) {
    return (float)((double)prl_comb(n_cards, n_hole) / (double)prl_comb(n_cards - n_hole, n_hole));
}

int do_update_reach(prl_solver* s, const PrlDevState& st) {
    prl_launch_reach(s->T, st, s->ft.level_start.data(), s->stream);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int do_compute_ev(prl_solver* s, const PrlDevState& st) {
    prl_launch_ev(s->T, st, s->ft.level_start.data(), s->d_term_nodes, s->n_term, s->stream);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int ensure_ev(prl_solver* s) {
    if (s->ev_valid) return PRL_OK;
    TRY(do_compute_ev(s, s->S));
    s->ev_valid = true;
    return PRL_OK;
}

int ensure_hist(prl_solver* s, int need) {
    if (need <= s->hist_cap) return PRL_OK;
    int cap = s->hist_cap ? s->hist_cap : 1024;
    while (cap < need) cap *= 2;
    float* q = nullptr;
    TRY(dev_alloc(s, &q, (size_t)cap * 2));
    if (s->d_expl_hist && s->hist_cap) PRL_HIP_TRY(hipMemcpyAsync(q, s->d_expl_hist, (size_t)s->hist_cap * 2 * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    s->d_expl_hist = q;  // the old block stays in `allocs` and is released with the solver
    s->hist_cap = cap;
    return PRL_OK;
}

int record_expl(prl_solver* s) {
    TRY(ensure_hist(s, s->iter + 1));
    PRL_HIP_TRY(hipMemcpyAsync(s->d_expl_hist + (size_t)s->iter * 2, s->S.expl, 2 * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    return PRL_OK;
}

}  // namespace

extern "C" {

int32_t prl_solver_create(const prl_tree_t* tree, int32_t variant, int32_t delay, prl_solver_t** out) {
    if (!tree || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (variant < 0 || variant > 2 || delay < 0) { prl_set_error("bad variant / delay"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device: the solver has no CPU fallback"); return PRL_ERR_NO_DEVICE; }
    const PrlFlatTree& ft = *prl_tree_flat(tree);
    const PrlRules& r = ft.rules;
    if (r.n_hole_cards == 1 && (r.range_size > 128 || ft.board_len != 1)) { prl_set_error("1-card games: R <= 128, 1 board card"); return PRL_ERR_UNSUPPORTED; }
    if (r.n_hole_cards == 2 && (r.n_cards != 52 || r.n_suits != 4 || ft.board_len != 5 || r.rank_rule != 2)) {
        prl_set_error("2-card games: 52-card deck with 5-card boards (Flop5Holdem) only");
        return PRL_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < ft.n_nodes; ++i)
        if (ft.kind[i] == PRL_NODE_DECISION && ft.n_children[i] > 96) { prl_set_error("more than 96 actions at a node"); return PRL_ERR_UNSUPPORTED; }
    prl_solver* s = new prl_solver();
    s->ft = ft;
    s->variant = variant;
    s->delay = delay;
#define FAIL_IF(x) do { int e_ = (x); if (e_) { prl_solver_destroy(s); return e_; } } while (0)
    if (hipStreamCreate(&s->stream) != hipSuccess) { prl_set_error("hipStreamCreate failed"); delete s; return PRL_ERR_HIP; }
    PrlDevTree& T = s->T;
    T.n_nodes = ft.n_nodes; T.n_cols = ft.n_cols; T.R = r.range_size; T.n_hole = r.n_hole_cards; T.n_cards = r.n_cards;
    T.n_suits = r.n_suits; T.rank_rule = r.rank_rule; T.n_boards = ft.n_boards; T.board_len = ft.board_len; T.n_levels = ft.n_levels;
    FAIL_IF(dev_upload(s, &T.kind, ft.kind));
    FAIL_IF(dev_upload(s, &T.actor, ft.actor));
    FAIL_IF(dev_upload(s, &T.parent, ft.parent));
    FAIL_IF(dev_upload(s, &T.child_idx, ft.child_idx));
    FAIL_IF(dev_upload(s, &T.acted_last, ft.acted_last));
    FAIL_IF(dev_upload(s, &T.board_id, ft.board_id));
    FAIL_IF(dev_upload(s, &T.main_pot, ft.main_pot));
    FAIL_IF(dev_upload(s, &T.n_children, ft.n_children));
    FAIL_IF(dev_upload(s, &T.first_col, ft.first_col));
    FAIL_IF(dev_upload(s, &T.child_start, ft.child_start));
    FAIL_IF(dev_upload(s, &T.child_list, ft.child_list));
    FAIL_IF(dev_upload(s, &T.level_nodes, ft.level_nodes));
    FAIL_IF(dev_upload(s, &T.boards, ft.boards));
    std::vector<int16_t> hole((size_t)T.R * 2);
    for (int h = 0; h < T.R; ++h) {
        int c1, c2;
        prl_hand_cards(r, h, &c1, &c2);
        hole[2 * h] = (int16_t)c1;
        hole[2 * h + 1] = (int16_t)c2;
    }
    FAIL_IF(dev_upload(s, &T.hole, hole));
    int n_chance_children = ft.n_boards;
    for (int i = 0; i < ft.n_nodes; ++i)
        if (ft.kind[i] == PRL_NODE_CHANCE) { n_chance_children = ft.n_children[i]; break; }
    T.chance_prob = chance_prob_f32(n_chance_children, r.n_cards, r.n_hole_cards, ft.board_len);
    T.eq_const = eq_const_f32(r.n_cards, r.n_hole_cards);

    std::vector<int32_t> term, np[2];
    for (int i = 0; i < ft.n_nodes; ++i) {
        if (ft.kind[i] >= PRL_NODE_TERM_FOLD) term.push_back(i);
        if (ft.kind[i] == PRL_NODE_DECISION) np[ft.actor[i]].push_back(i);
    }
    s->n_term = (int)term.size();
    FAIL_IF(dev_upload(s, (const int32_t**)&s->d_term_nodes, term));
    for (int p = 0; p < 2; ++p) {
        s->n_nodes_p[p] = (int)np[p].size();
        FAIL_IF(dev_upload(s, (const int32_t**)&s->d_nodes_p[p], np[p]));
    }
    FAIL_IF(dev_upload(s, (const int32_t**)&s->d_col_node, ft.col_node));

    if (T.n_hole == 2) {  // showdown plans, built on the device
        const int n_plans = T.n_boards + 1;
        T.plan_stride = T.R;
        T.cl_stride = T.n_cards * (T.n_cards - 1);
        int16_t *sh, *pos, *gs, *ge, *cl;
        int32_t* nl;
        FAIL_IF(dev_alloc(s, &sh, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &pos, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &gs, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &ge, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &cl, (size_t)n_plans * T.cl_stride));
        FAIL_IF(dev_alloc(s, &nl, (size_t)n_plans));
        prl_launch_plan_build(T, n_plans, sh, pos, gs, ge, cl, nl, s->stream);
        T.plan_sh = sh; T.plan_pos = pos; T.plan_gs = gs; T.plan_ge = ge; T.plan_cl = cl; T.plan_nlive = nl;
    }

    const size_t nc = (size_t)T.n_cols * T.R;
    FAIL_IF(dev_alloc(s, &s->S.strategy, nc));
    FAIL_IF(dev_alloc(s, &s->S.strat_f64, (size_t)T.n_nodes));
    FAIL_IF(dev_alloc(s, &s->S.regret, nc));
    FAIL_IF(dev_alloc(s, &s->S.avg, nc));
    FAIL_IF(dev_alloc(s, &s->S.avg_f64, (size_t)T.n_nodes));
    if (variant != PRL_CFR_PLUS) FAIL_IF(dev_alloc(s, &s->S.avg_sum, nc));
    FAIL_IF(alloc_node_vectors(s, &s->S, true));
    if (hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        prl_set_error("device error while building the showdown plans");
        prl_solver_destroy(s);
        return PRL_ERR_HIP;
    }
#undef FAIL_IF
    *out = s;
    return prl_solver_reset(s);
}

void prl_solver_destroy(prl_solver_t* s) {
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// CFRBase.reset (_CFRBase.py:110-120): clear regrets / averages, uniform strategy, reach, EV (+ exploitability)
int32_t prl_solver_reset(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    const size_t nc = (size_t)s->T.n_cols * s->T.R;
    s->iter = 0;
    PRL_HIP_TRY(hipMemsetAsync(s->S.regret, 0, nc * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(s->S.avg, 0, nc * sizeof(double), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(s->S.avg_f64, 0, (size_t)s->T.n_nodes, s->stream));
    if (s->S.avg_sum) PRL_HIP_TRY(hipMemsetAsync(s->S.avg_sum, 0, nc * sizeof(float), s->stream));
    TRY(prl_solver_fill_uniform(s));
    TRY(ensure_ev(s));
    return record_expl(s);
}

int32_t prl_solver_fill_uniform(prl_solver_t* s) {  // PublicTree.fill_uniform_random (StrategyFiller.py:17-24)
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipMemsetAsync(s->S.strat_f64, 0, (size_t)s->T.n_nodes, s->stream));
    prl_launch_fill_uniform(s->T, s->S, s->d_col_node, s->stream);
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

// arbitrary strategy, column-major [n_cols][R] (= node.strategy.T per decision node, DFS order); float32 or float64 host data.
// Equivalent of fill_with_agent_policy / fill_random_random + update_reach_probs (StrategyFiller.py:26-43).
int32_t prl_solver_set_strategy(prl_solver_t* s, const void* strat, int32_t is_f64) {
    if (!s || !strat) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    const size_t nc = (size_t)s->T.n_cols * s->T.R;
    std::vector<double> tmp;
    const double* src = (const double*)strat;
    if (!is_f64) {
        tmp.resize(nc);
        const float* f = (const float*)strat;
        for (size_t i = 0; i < nc; ++i) tmp[i] = (double)f[i];  // exact widening: storage only, arithmetic stays float32
        src = tmp.data();
    }
    PRL_HIP_TRY(hipMemcpyAsync(s->S.strategy, src, nc * sizeof(double), hipMemcpyHostToDevice, s->stream));
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));  // tmp goes out of scope
    PRL_HIP_TRY(hipMemsetAsync(s->S.strat_f64, is_f64 ? 1 : 0, (size_t)s->T.n_nodes, s->stream));
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

int32_t prl_solver_update_reach(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

int32_t prl_solver_compute_ev(prl_solver_t* s) {  // PublicTree.compute_ev (PublicTree.py:128)
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    s->ev_valid = false;
    return ensure_ev(s);
}

// One CFRBase.iteration() (_CFRBase.py:122-134) without _evaluate_avg_strats (see prl_solver_eval_avg). Asynchronous.
// The reference recomputes the EVs at the top of the p = 0 half although nothing changed since the pass that closed the
// previous iteration; that pass is reused here (identical values), so an iteration costs two EV passes, not three.
int32_t prl_solver_iteration(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    for (int p = 0; p < 2; ++p) {
        TRY(ensure_ev(s));
        prl_launch_regret_strategy(s->T, s->S, s->d_nodes_p[p], s->n_nodes_p[p], p, s->variant, s->iter, s->stream);
        s->ev_valid = false;
        TRY(do_update_reach(s, s->S));
        int mode = 0;
        double m_old = 0., m_new = 0.;
        if (s->variant == PRL_CFR_PLUS) {  // CFRPlus.py:65-87: float64 weights from integer sums
            if (s->iter > s->delay) {
                long long cw = 0;
                for (int k = s->delay + 1; k <= s->iter; ++k) cw += k;
                long long nw = s->iter - s->delay + 1;
                m_old = (double)cw / (double)(cw + nw);
                m_new = (double)nw / (double)(cw + nw);
                mode = 2;
            } else if (s->iter == s->delay) mode = 1;
        }
        prl_launch_average(s->T, s->S, s->d_nodes_p[p], s->n_nodes_p[p], p, s->variant, s->iter, mode, m_old, m_new, s->stream);
    }
    s->iter += 1;
    TRY(ensure_ev(s));
    PRL_HIP_TRY(hipGetLastError());
    return record_expl(s);
}

int32_t prl_solver_iterations(prl_solver_t* s, int32_t n) {
    for (int i = 0; i < n; ++i) TRY(prl_solver_iteration(s));
    return PRL_OK;
}

// n iterations bracketed by HIP events on the solver's own stream (what bench.py's roofline figure is derived from)
int32_t prl_solver_time_iterations(prl_solver_t* s, int32_t n, float* out_ms) {
    if (!s || !out_ms || n < 0) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    TRY(ensure_hist(s, s->iter + n + 1));
    PRL_HIP_TRY(hipEventRecord(e0, s->stream));
    int rc = prl_solver_iterations(s, n);
    if (rc == PRL_OK) {
        PRL_HIP_TRY(hipEventRecord(e1, s->stream));
        PRL_HIP_TRY(hipEventSynchronize(e1));
        PRL_HIP_TRY(hipEventElapsedTime(out_ms, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int32_t prl_solver_sync(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

// root exploitability [seat 0, seat 1] of the CURRENT strategy, raw float32 as in node.exploitability (ValueFiller.py:101)
int32_t prl_solver_exploitability(prl_solver_t* s, float* out2) {
    if (!s || !out2) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    TRY(ensure_ev(s));
    PRL_HIP_TRY(hipMemcpyAsync(out2, s->S.expl, 2 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    return prl_solver_sync(s);
}

// _CFRBase._evaluate_avg_strats (_CFRBase.py:218-262): exploitability of the average strategy; training state untouched
int32_t prl_solver_eval_avg(prl_solver_t* s, float* out2) {
    if (!s || !out2) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (!s->eval_ready) {
        TRY(alloc_node_vectors(s, &s->Seval, false));
        s->eval_ready = true;
    }
    PrlDevState E = s->Seval;
    E.strategy = s->S.avg;
    E.strat_f64 = s->S.avg_f64;
    E.regret = nullptr; E.avg = nullptr; E.avg_sum = nullptr; E.avg_f64 = nullptr;
    TRY(do_update_reach(s, E));
    TRY(do_compute_ev(s, E));
    PRL_HIP_TRY(hipMemcpyAsync(out2, E.expl, 2 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    return prl_solver_sync(s);
}

int32_t prl_solver_get(prl_solver_t* s, int32_t field, void* out) {
    if (!s || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    const size_t nv = (size_t)s->T.n_nodes * 2 * s->T.R, nc = (size_t)s->T.n_cols * s->T.R;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (field) {
        case PRL_SF_REACH: src = s->S.reach; bytes = nv * 4; break;
        case PRL_SF_EV: TRY(ensure_ev(s)); src = s->S.ev; bytes = nv * 4; break;
        case PRL_SF_EV_BR: TRY(ensure_ev(s)); src = s->S.ev_br; bytes = nv * 4; break;
        case PRL_SF_STRATEGY: src = s->S.strategy; bytes = nc * 8; break;
        case PRL_SF_STRAT_F64: src = s->S.strat_f64; bytes = (size_t)s->T.n_nodes; break;
        case PRL_SF_REGRET: src = s->S.regret; bytes = nc * 4; break;
        case PRL_SF_AVG: src = s->S.avg; bytes = nc * 8; break;
        case PRL_SF_AVG_F64: src = s->S.avg_f64; bytes = (size_t)s->T.n_nodes; break;
        case PRL_SF_AVG_SUM: src = s->S.avg_sum; bytes = nc * 4; break;
        case PRL_SF_BR_IDX: TRY(ensure_ev(s)); src = s->S.br_idx; bytes = (size_t)s->T.n_nodes * s->T.R * 4; break;
        case PRL_SF_EXPL_HISTORY: src = s->d_expl_hist; bytes = (size_t)(s->iter + 1) * 2 * 4; break;
        case PRL_SF_ITER: *(int32_t*)out = s->iter; return PRL_OK;
        case PRL_SF_CONSTANTS: ((float*)out)[0] = s->T.chance_prob; ((float*)out)[1] = s->T.eq_const; return PRL_OK;
        case PRL_SF_BYTES_ALLOCATED: *(int64_t*)out = (int64_t)s->bytes_allocated; return PRL_OK;
        default: prl_set_error("unknown solver field"); return PRL_ERR_ARG;
    }
    if (!src) { prl_set_error("field not available for this variant"); return PRL_ERR_STATE; }
    PRL_HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, s->stream));
    return prl_solver_sync(s);
}

}  // extern "C"
