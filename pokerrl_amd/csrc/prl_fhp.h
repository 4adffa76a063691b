// Compile-time descriptions of board subtrees ("shapes") + runtime parameters of the fused board kernels.
//
// A shape is the DFS pre-order listing of a board subtree: per node its kind, the seat to act and the number of children --
// everything else (children, parents, action columns, who folded, terminal slots ...) is DERIVED from that listing by constexpr
// code (PrlFhpDerive), so registering another betting structure is three arrays and one line in prl_fhp_kernels.hip. The board
// pass is instantiated once per registered shape (the walk is a compile-time DFS: per-hand values live in registers, which is
// what makes the pass fast), and prl_fhp_match_shape() picks the instantiation whose listing equals the flat tree of the actual
// game; trees with any other subtree fall back to the general level-synchronous engine. Pots are runtime data.
//
// Registered (reference game: PokerRL/game/games.py:222-254, pot-size raises, BB acts first post-flop; captured from the
// reference env in tests/golden/tree_Flop5Holdem_1board.npz and by walking the env for the variants):
//   FHP15  Flop5Holdem, two raises per round, stacks >= 1100 (the benchmark tree):
//     0 seat1 {check -> 1, bet -> 9};  1 seat0 {check -> 2 SD, bet -> 3};  3 seat1 {fold 4, call 5 SD, raise -> 6};
//     6 seat0 {fold 7, call 8 SD};  9 seat0 {fold 10, call 11 SD, raise -> 12};  12 seat1 {fold 13, call 14 SD}
//   FHP9   the same game when the first post-flop bet is the last one: stacks 400..900 (the bet is all-in), or one raise per
//          post-flop round:  0 seat1 {check -> 1, bet -> 6};  1 seat0 {check -> 2 SD, bet -> 3};  3 seat1 {fold 4, call 5 SD};
//          6 seat0 {fold 7, call 8 SD}
//   FHP21  three raises post-flop (MAX_N_RAISES_PER_ROUND[FLOP] = 3): 21 nodes, 20 action columns
#pragma once
#include "prl_defs.h"
#include "prl_solver_types.h"
#include "prl_tree.h"

enum { PRL_SRC_REGRET = 0, PRL_SRC_UNIFORM64 = 1, PRL_SRC_ARR64 = 2, PRL_SRC_ARR32 = 3,
       PRL_SRC_STRAT32 = 4 /* PrlFhpParams::regret points at float32 STRATEGY columns (regret layout): played as is */,
       PRL_SRC_AVGF32 = 5 /* PrlFhpParams::avg32: the opt-in float32 running average, widened and played with float64 arithmetic like ARR64 */ };
// UPDATE0 / UPDATE1: that seat's values + regret / average update. EVAL: both seats + best response.
// UPDATE0_EVAL: EVAL and UPDATE0 of the same strategy in one pass (the evaluation that closes iteration t and the first
// half of iteration t + 1 read the same regrets).
// Steady state of prl_solver_iterations (two single-seat passes per iteration, PRL_FHP_* below):
//   UPDATE1_EVAL1: seat 1's batch; after its regrets are updated the hand-local bottom-up phase runs a second time with
//                  the NEW strategy: seat 1's value and best response against seat 0's (final) strategy of this iteration
//                  -- seat 1's half of the exploitability, at the price of one phase E instead of a whole batch
//   UPDATE0_BR:    seat 0's batch of the next iteration with best response: seat 0's half of the exploitability of the
//                  previous iterate (its own strategy is still the old one) + the update
//   EVAL0:         seat 0's batch with best response, no update (closes a run of iterations)
enum { PRL_FHP_UPDATE0 = 0, PRL_FHP_UPDATE1 = 1, PRL_FHP_EVAL = 2, PRL_FHP_UPDATE0_EVAL = 3, PRL_FHP_UPDATE0_BR = 4, PRL_FHP_UPDATE1_EVAL1 = 5,
       PRL_FHP_EVAL0 = 6 };
constexpr bool prl_fhp_runs_seat(int mode, int p) {
    return mode == PRL_FHP_EVAL || mode == PRL_FHP_UPDATE0_EVAL ? true
         : (mode == PRL_FHP_UPDATE1 || mode == PRL_FHP_UPDATE1_EVAL1) ? p == 1 : p == 0;
}
constexpr bool prl_fhp_updates(int mode, int p) {
    return p == 0 ? (mode == PRL_FHP_UPDATE0 || mode == PRL_FHP_UPDATE0_EVAL || mode == PRL_FHP_UPDATE0_BR)
                  : (mode == PRL_FHP_UPDATE1 || mode == PRL_FHP_UPDATE1_EVAL1);
}
constexpr bool prl_fhp_with_br(int mode) { return mode >= PRL_FHP_EVAL; }
// root vectors a pass stores per board (row length of PrlFhpParams::board_out in units of R), see prl_k_fhp_pass
constexpr int prl_fhp_out_width(int mode) {
    return mode == PRL_FHP_UPDATE1_EVAL1 ? 3
         : (prl_fhp_runs_seat(mode, 0) && prl_fhp_runs_seat(mode, 1) ? 2 : 1) * (prl_fhp_with_br(mode) ? 2 : 1);
}

#define PRL_FHP_MAX_NODES 40
#define PRL_FHP_MAX_DEC 12

// the three arrays of a shape
struct PrlFhpSpec15 {
    static constexpr int N_NODES = 15;
    static constexpr int K(int n) { constexpr int t[N_NODES] = {0, 0, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3}; return t[n]; }
    static constexpr int A(int n) { constexpr int t[N_NODES] = {1, 0, -1, 1, -1, -1, 0, -1, -1, 0, -1, -1, 1, -1, -1}; return t[n]; }
    static constexpr int C(int n) { constexpr int t[N_NODES] = {2, 2, 0, 3, 0, 0, 2, 0, 0, 3, 0, 0, 2, 0, 0}; return t[n]; }
};
struct PrlFhpSpec9 {
    static constexpr int N_NODES = 9;
    static constexpr int K(int n) { constexpr int t[N_NODES] = {0, 0, 3, 0, 2, 3, 0, 2, 3}; return t[n]; }
    static constexpr int A(int n) { constexpr int t[N_NODES] = {1, 0, -1, 1, -1, -1, 0, -1, -1}; return t[n]; }
    static constexpr int C(int n) { constexpr int t[N_NODES] = {2, 2, 0, 2, 0, 0, 2, 0, 0}; return t[n]; }
};
struct PrlFhpSpec21 {
    static constexpr int N_NODES = 21;
    static constexpr int K(int n) { constexpr int t[N_NODES] = {0, 0, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3}; return t[n]; }
    static constexpr int A(int n) { constexpr int t[N_NODES] = {1, 0, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1}; return t[n]; }
    static constexpr int C(int n) { constexpr int t[N_NODES] = {2, 2, 0, 3, 0, 0, 3, 0, 0, 2, 0, 0, 3, 0, 0, 3, 0, 0, 2, 0, 0}; return t[n]; }
};

// four raises per round (LimitHoldem's MAX_N_RAISES_PER_ROUND, games.py:134-167): the betting subtree of every post-flop street of
// a full-limit game. Used by the per-street engine (prl_st.h), whose non-final streets read the kind-3 leaves as "the street goes on"
// (a chance node in the flat tree) instead of showdowns.
struct PrlFhpSpec27 {
    static constexpr int N_NODES = 27;
    static constexpr int K(int n) { constexpr int t[N_NODES] = {0, 0, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3}; return t[n]; }
    static constexpr int A(int n) { constexpr int t[N_NODES] = {1, 0, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1}; return t[n]; }
    static constexpr int C(int n) { constexpr int t[N_NODES] = {2, 2, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 2, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 2, 0, 0}; return t[n]; }
};
// five raises per round: DiscretizedNLHoldem with pot-sized raises at its 200-big-blind default stacks (games.py:114-131) -- 100, 300, 900, 2700, 8100, all-in
struct PrlFhpSpec33 {
    static constexpr int N_NODES = 33;
    static constexpr int K(int n) { constexpr int t[N_NODES] = {0, 0, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3}; return t[n]; }
    static constexpr int A(int n) { constexpr int t[N_NODES] = {1, 0, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1, 1, -1, -1, 0, -1, -1}; return t[n]; }
    static constexpr int C(int n) { constexpr int t[N_NODES] = {2, 2, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 2, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 2, 0, 0}; return t[n]; }
};
// everything the walk needs, derived from a spec (all constexpr: evaluated by the compiler for the template recursion)
template <class S>
struct PrlFhpDerive {
    static constexpr int N_NODES = S::N_NODES;
    static constexpr int kind(int n) { return S::K(n); }
    static constexpr int actor(int n) { return S::A(n); }
    static constexpr int nch(int n) { return S::C(n); }
    static constexpr int subtree_size(int n) {
        int size = 1;
        for (int i = 0, c = n + 1; i < nch(n); ++i) { const int t = subtree_size(c); size += t; c += t; }
        return size;
    }
    static constexpr int child(int n, int i) {
        if (i >= nch(n)) return -1;
        int c = n + 1;
        for (int k = 0; k < i; ++k) c += subtree_size(c);
        return c;
    }
    static constexpr int parent(int n) {
        for (int p = n - 1; p >= 0; --p)
            for (int i = 0; i < nch(p); ++i)
                if (child(p, i) == n) return p;
        return -1;
    }
    static constexpr int count_kind(int k, int below) { int c = 0; for (int m = 0; m < below; ++m) c += kind(m) == k; return c; }
    static constexpr int N_DEC = count_kind(PRL_NODE_DECISION, N_NODES);
    static constexpr int N_SHOW = count_kind(PRL_NODE_TERM_SHOWDOWN, N_NODES);
    static constexpr int N_FOLD = count_kind(PRL_NODE_TERM_FOLD, N_NODES);
    static constexpr int col0(int n) {  // first action column of a decision node (columns in DFS order)
        if (kind(n) != PRL_NODE_DECISION) return -1;
        int c = 0;
        for (int m = 0; m < n; ++m) c += kind(m) == PRL_NODE_DECISION ? nch(m) : 0;
        return c;
    }
    static constexpr int n_cols_() { int c = 0; for (int m = 0; m < N_NODES; ++m) c += kind(m) == PRL_NODE_DECISION ? nch(m) : 0; return c; }
    static constexpr int N_COLS = n_cols_();
    static constexpr int folder(int n) { return kind(n) == PRL_NODE_TERM_FOLD ? actor(parent(n)) : -1; }  // the seat that folded
    // slot of a terminal among a seat's terminal vectors: showdown nodes 0 .. N_SHOW-1, then the fold nodes
    static constexpr int term_slot(int n) {
        return kind(n) == PRL_NODE_TERM_SHOWDOWN ? count_kind(PRL_NODE_TERM_SHOWDOWN, n)
             : kind(n) == PRL_NODE_TERM_FOLD ? N_SHOW + count_kind(PRL_NODE_TERM_FOLD, n) : -1;
    }
    static constexpr int col_node(int col) {
        for (int m = 0; m < N_NODES; ++m)
            if (kind(m) == PRL_NODE_DECISION && col0(m) <= col && col < col0(m) + nch(m)) return m;
        return -1;
    }
    static constexpr int col_actor(int col) { return actor(col_node(col)); }
    static constexpr int dec_node(int j) {
        for (int m = 0, k = 0; m < N_NODES; ++m)
            if (kind(m) == PRL_NODE_DECISION && k++ == j) return m;
        return -1;
    }
};

// runtime view of a registered shape (host: matching; device: the small bookkeeping kernels)
struct PrlFhpShapeDesc {
    int32_t n_nodes, n_cols, n_dec;
    int32_t kind[PRL_FHP_MAX_NODES], actor[PRL_FHP_MAX_NODES], nch[PRL_FHP_MAX_NODES], parent[PRL_FHP_MAX_NODES], col0[PRL_FHP_MAX_NODES],
        folder[PRL_FHP_MAX_NODES];
    int32_t dec_nch[PRL_FHP_MAX_DEC], dec_col0[PRL_FHP_MAX_DEC];
};
template <class D>
inline PrlFhpShapeDesc prl_fhp_describe() {
    PrlFhpShapeDesc d = {};
    d.n_nodes = D::N_NODES; d.n_cols = D::N_COLS; d.n_dec = D::N_DEC;
    static_assert(D::N_NODES <= PRL_FHP_MAX_NODES && D::N_DEC <= PRL_FHP_MAX_DEC, "enlarge PRL_FHP_MAX_*");
    for (int n = 0; n < D::N_NODES; ++n) {
        d.kind[n] = D::kind(n); d.actor[n] = D::actor(n); d.nch[n] = D::nch(n); d.parent[n] = D::parent(n); d.col0[n] = D::col0(n);
        d.folder[n] = D::folder(n);
    }
    for (int j = 0; j < D::N_DEC; ++j) { d.dec_nch[j] = D::nch(D::dec_node(j)); d.dec_col0[j] = D::col0(D::dec_node(j)); }
    return d;
}
enum { PRL_FHP_SHAPE_15 = 0, PRL_FHP_SHAPE_9 = 1, PRL_FHP_SHAPE_21 = 2, PRL_FHP_N_SHAPES = 3 };
const PrlFhpShapeDesc& prl_fhp_shape_desc(int shape_id);  // prl_fhp_kernels.hip

// SORTED STORAGE (round 4). Inside a board subtree nothing but the root vectors ever meets another board, so the board's action columns
// need not be indexed by hand: they are stored in the board's own RANK-SORTED order, live hands only -- element q of a board column belongs
// to the hand at sorted position q of that board's showdown plan (PRL_PP_*: `sh`), FHP_NP = 1088 elements per column (1081 live positions
// of a 5-card board + padding). The hands a board blocks (245 of 1326) have no storage: their regrets are 0 for ever, their strategy is
// the uniform one, their running average follows a scalar recurrence (prl_solver.hip: blocked_avg) -- 18.5 % fewer HBM bytes per pass.
// A lane of the board pass owns two ADJACENT POSITIONS, so its scatter into the sorted domain (phase B), its reads at its own position
// (phase E) and its column loads / stores are all linear. The arrays below are the BOARD REGIONS of the solver's column arrays:
// [n_boards][n_cols_board][np]; the trunk's columns stay [R] in hand order in front of them. prl_solver_get / set / checkpoints translate
// (prl_launch_fhp_expand / _compact).
#define PRL_FHP_NP 1088
struct PrlFhpParams {
    int32_t n_boards, R;
    int32_t np;                 // elements per board column = PRL_FHP_NP
    int32_t variant, iter;
    int32_t max_grid;
    int32_t shape;              // PRL_FHP_SHAPE_*: which instantiation of the board pass walks this tree
    int32_t n_cols_board, n_dec;  // action columns / decision nodes per board subtree, and per decision node (DFS order):
    int32_t dec_nch[PRL_FHP_MAX_DEC], dec_col0[PRL_FHP_MAX_DEC];
    float chance_prob, eq_const;
    float pot[PRL_FHP_MAX_NODES];      // main pot of the terminal nodes (by local node id)
    const float* chance_reach;  // [2][R] reach at the chance node (trunk state, hand order)
    const float* board_w;       // [n_boards] the chance weight of every board: chance_prob, or chance_prob * multiplicity of a weighted board (prl_solver_create_weighted)
    float* regret;              // board region [n_boards][n_cols_board][np]; PRL_SRC_STRAT32: an explicit float32 strategy in the same layout
    double* avg;                // board region: average strategy, updated by the update passes when avg_mode != 0
    float* avg32;               // opt-in (prl_solver_create_opts: PRL_SOLVER_AVG_F32): the same average STORED as float32 -- read, widened, blended in
                                // float64 with the reference's weights, rounded on the store; `avg` is not touched by the board pass then
    int32_t avg_mode;           // 0: no update (before the delay), 1: avg = strategy, 2: avg = m_old * avg + m_new * strategy
    double m_old, m_new;        // CFRPlus.py:65-87 weights (float64)
    // Vanilla / Linear CFR: the reach-weighted average of seat q needs q's NEW reach, known only after the trunk update that
    // follows q's pass -- so it rides on the next pass that walks q's reach (phase B for seat q): bit q of avgsum_mask
    float* avg_sum;             // board region: node.data["avg_strat_sum"] (VanillaCFR.py:40-55, LinearCFR.py:41-57)
    int32_t avgsum_mask, avgsum_iter[2];
    int32_t block_sum;          // 1: board_out holds one row per PRL_CHANCE_BLOCK boards (level 0 of the chance sum done by the pass)
    int32_t exp;                // FHP_EXPERIMENT builds: run-time switch between two code paths (prl_debug_set_experiment)
    int32_t no_steady;          // tests: 1 = never take the CFR+ steady-state specialisation of the pass (prl_fhp_pass.inc, FhpCtxT)
    const double* strat_arr;    // board region of an explicit float64 strategy (the average being evaluated / caller-provided)
    float* board_out;           // [n_boards or n_blocks][prl_fhp_out_width(mode)][R] root vectors (hand order)
    const int16_t* plan_pp;     // [n_boards][PRL_PP_STRIDE] position-domain plans (prl_solver_types.h)
    const uint32_t* plan_clx;   // [n_boards][PRL_CLX_WORDS] per-lane records of the per-card scans (prl_solver_types.h)
    unsigned long long* timing; // PRL_FHP_TIMING builds: [8] shader-clock accumulators per phase (prologue, B, C, D, E, epilogue)
};

// host: does the flat tree consist of a trunk + ONE chance node whose board subtrees all have one of the registered shapes?
// On success returns the shape id (else -1) and fills the chance node id, the first board node, the global column base and the
// terminal pots.
int prl_fhp_match_shape(const PrlFlatTree& t, int* chance_node, int* first_board_node, int* col_base, float* pots /*[PRL_FHP_MAX_NODES]*/);

int prl_launch_fhp_pass(const PrlFhpParams& prm, int mode, int src0, int src1, void* stream);
// the strategy the regrets imply, board region -> board region (float64)
void prl_launch_fhp_strategy_from_regret(const PrlFhpParams& prm, double* out_region, void* stream);
void prl_launch_fhp_avg_from_sum(const PrlFhpParams& prm, void* stream);
// suit symmetrisation of chance-summed root vectors (prl_solver_create_weighted): out[v][h] = (sum over the hands of h's class, ascending) / class size
void prl_launch_fhp_symmetrize(const float* in, int n_vec, int R, const int32_t* class_of, const int32_t* class_start, const int32_t* class_hands, float* out, void* stream);  // Vanilla / Linear: avg columns of the boards from avg_sum
// sorted storage <-> the caller's [n_cols][R] hand-order columns, boards [b0, b0 + nb): elem = 4 (float32) or 8 (float64) bytes.
// expand: dst[(b - b0) * n_cols_board + j][h] = region[b][j][pos_b(h)] for live hands; blocked hands get fill[j] (elem bytes each, by
// LOCAL column j; nullptr = zeros) or, when `blocked_src` is given, blocked_src[b][j][k] (k-th blocked hand of the board, hand order).
// compact: the inverse; `blocked_dst` (optional) keeps what the caller had for the blocked hands.
void prl_launch_fhp_expand(const PrlFhpParams& prm, const void* region, int elem, int b0, int nb, const void* fill_by_col, const void* blocked_src,
                           void* dst, void* stream);
void prl_launch_fhp_compact(const PrlFhpParams& prm, const void* src, int elem, int b0, int nb, void* region, void* blocked_dst, void* stream);
#define PRL_FHP_NBLOCKED (1326 - 1081)
// W = floats per board / unit: 2R for both seats' vectors, R for one seat's
void prl_launch_fhp_chance_sum(const float* d_board_vals, int n_boards, int W, float* d_scratch, float* d_dest, void* stream);
// sharded solve (prl_solver_create_sharded): local reduction up to `level`, all-gather, then the remaining levels
int prl_fhp_units_at_level(int n_boards, int level);
void prl_launch_fhp_chance_partial(const float* d_board_vals, int n_boards, int level, int W, float* d_scratch, float* d_units, void* stream);
void prl_launch_fhp_chance_partial_from_blocks(const float* d_blocks, int n_blk, int level, int W, float* d_units, void* stream);
void prl_launch_fhp_chance_finish(const float* d_units, int n_units, int level, int W, float* d_scratch, float* d_dest, void* stream);
void prl_launch_fhp_compact_gathered(const float* d_in, int world, int n_which, int n_units, int W, float* d_out, void* stream);
