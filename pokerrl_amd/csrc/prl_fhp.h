// Compile-time description of the Flop5Holdem board subtree + runtime parameters of the fused board kernels.
//
// Shape (local node ids in DFS pre-order; reference game: PokerRL/game/games.py:222-254, pot-size raises, at most two
// raises per round, BB acts first post-flop -- captured from the reference env in tests/golden/tree_Flop5Holdem_1board.npz):
//
//   0 seat1 {check -> 1, bet -> 9}
//   1   seat0 {check -> 2 SHOWDOWN, bet -> 3}
//   3     seat1 {fold -> 4, call -> 5 SHOWDOWN, raise -> 6}
//   6       seat0 {fold -> 7, call -> 8 SHOWDOWN}
//   9   seat0 {fold -> 10, call -> 11 SHOWDOWN, raise -> 12}
//   12    seat1 {fold -> 13, call -> 14 SHOWDOWN}
//
// Only the SHAPE is compiled in; pots are runtime data, and prl_fhp_shape_matches() checks the flat tree of the actual
// game against it before the fused engine is selected (otherwise the general level-synchronous engine is used).
#pragma once
#include "prl_defs.h"
#include "prl_solver_types.h"
#include "prl_tree.h"

enum { PRL_SRC_REGRET = 0, PRL_SRC_UNIFORM64 = 1, PRL_SRC_ARR64 = 2, PRL_SRC_ARR32 = 3,
       PRL_SRC_STRAT32 = 4 /* PrlFhpParams::regret points at float32 STRATEGY columns (regret layout): played as is */ };
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

struct PrlFhpShape {
    static constexpr int N_NODES = 15;
    static constexpr int N_COLS = 14;
    static constexpr int N_DEC = 6;
    static constexpr int N_DEC_PER_SEAT = 3;

    static constexpr int kind(int n) {
        constexpr int K[N_NODES] = {0, 0, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3, 0, 2, 3};
        return K[n];
    }
    static constexpr int actor(int n) {
        constexpr int A[N_NODES] = {1, 0, -1, 1, -1, -1, 0, -1, -1, 0, -1, -1, 1, -1, -1};
        return A[n];
    }
    static constexpr int nch(int n) {
        constexpr int C[N_NODES] = {2, 2, 0, 3, 0, 0, 2, 0, 0, 3, 0, 0, 2, 0, 0};
        return C[n];
    }
    static constexpr int child(int n, int i) {
        constexpr int C[N_NODES][3] = {{1, 9, -1}, {2, 3, -1}, {-1, -1, -1}, {4, 5, 6},  {-1, -1, -1}, {-1, -1, -1}, {7, 8, -1}, {-1, -1, -1},
                                       {-1, -1, -1}, {10, 11, 12}, {-1, -1, -1}, {-1, -1, -1}, {13, 14, -1}, {-1, -1, -1}, {-1, -1, -1}};
        return C[n][i];
    }
    static constexpr int col0(int n) {
        constexpr int C[N_NODES] = {0, 2, -1, 4, -1, -1, 7, -1, -1, 9, -1, -1, 12, -1, -1};
        return C[n];
    }
    // seat that folded at a fold node (= the parent's actor)
    static constexpr int folder(int n) {
        constexpr int F[N_NODES] = {-1, -1, -1, -1, 1, -1, -1, 0, -1, -1, 0, -1, -1, 1, -1};
        return F[n];
    }
    // slot of a terminal node among the 9 terminal vectors of a seat: showdown nodes 0..4, fold nodes 5..8
    static constexpr int term_slot(int n) {
        constexpr int T[N_NODES] = {-1, -1, 0, -1, 5, 1, -1, 6, 2, -1, 7, 3, -1, 8, 4};
        return T[n];
    }
    static constexpr int parent(int n) {
        constexpr int P[N_NODES] = {-1, 0, 1, 1, 3, 3, 3, 6, 6, 0, 9, 9, 9, 12, 12};
        return P[n];
    }
    static constexpr int col_actor(int col) {
        constexpr int A[N_COLS] = {1, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1};
        return A[col];
    }
    static constexpr int dec_node(int j) {
        constexpr int D[N_DEC] = {0, 1, 3, 6, 9, 12};
        return D[j];
    }
    static constexpr int seat_node(int seat, int j) {
        constexpr int D[2][N_DEC_PER_SEAT] = {{1, 6, 9}, {0, 3, 12}};
        return D[seat][j];
    }
};

struct PrlFhpParams {
    int32_t n_boards, R;
    int32_t col_base;           // global action column of board 0, local column 0
    int32_t variant, iter;
    int32_t max_grid;
    float chance_prob, eq_const;
    float pot[PrlFhpShape::N_NODES];   // main pot of the terminal nodes (by local node id)
    const float* chance_reach;  // [2][R] reach at the chance node (trunk state)
    float* regret;              // [n_cols][R] (global column ids)
    double* avg;                // [n_cols][R] average strategy, updated by the update passes when avg_mode != 0
    int32_t avg_mode;           // 0: no update (before the delay), 1: avg = strategy, 2: avg = m_old * avg + m_new * strategy
    double m_old, m_new;        // CFRPlus.py:65-87 weights (float64)
    // Vanilla / Linear CFR: the reach-weighted average of seat q needs q's NEW reach, known only after the trunk update that
    // follows q's pass -- so it rides on the next pass that walks q's reach (phase B for seat q): bit q of avgsum_mask
    float* avg_sum;             // [n_cols][R] node.data["avg_strat_sum"] (VanillaCFR.py:40-55, LinearCFR.py:41-57)
    int32_t avgsum_mask, avgsum_iter[2];
    const double* strat_arr;    // explicit strategy (average strategy / caller-provided), [n_cols][R]
    float* board_out;           // [n_boards][prl_fhp_out_width(mode)][R] root vectors of every board subtree
    const uint16_t* hole_packed;// [R] c1 | c2 << 8
    int32_t plan_stride;
    const int16_t *plan_pos, *plan_hgs, *plan_hge;
    const uint32_t* plan_clx;   // [n_boards][PRL_CLX_WORDS] per-lane records of the per-card scans (prl_solver_types.h)
    const int32_t* plan_nlive;
    unsigned long long* timing; // PRL_FHP_TIMING builds: [8] shader-clock accumulators per phase (prologue, B, C, D, E, epilogue)
};

// host: does the flat tree consist of a trunk + ONE chance node whose board subtrees all have the compiled shape?
// On success fills the chance node id, the first board node, the global column base and the terminal pots.
bool prl_fhp_shape_matches(const PrlFlatTree& t, int* chance_node, int* first_board_node, int* col_base, float* pots /*[15]*/);

int prl_launch_fhp_pass(const PrlFhpParams& prm, int mode, int src0, int src1, void* stream);
void prl_launch_fhp_strategy_from_regret(const PrlFhpParams& prm, double* out_cols, void* stream);
void prl_launch_fhp_avg_from_sum(const PrlFhpParams& prm, void* stream);  // Vanilla / Linear: avg columns of the boards from avg_sum
// W = floats per board / unit: 2R for both seats' vectors, R for one seat's
void prl_launch_fhp_chance_sum(const float* d_board_vals, int n_boards, int W, float* d_scratch, float* d_dest, void* stream);
// sharded solve (prl_solver_create_sharded): local reduction up to `level`, all-gather, then the remaining levels
int prl_fhp_units_at_level(int n_boards, int level);
void prl_launch_fhp_chance_partial(const float* d_board_vals, int n_boards, int level, int W, float* d_scratch, float* d_units, void* stream);
void prl_launch_fhp_chance_finish(const float* d_units, int n_units, int level, int W, float* d_scratch, float* d_dest, void* stream);
void prl_launch_fhp_compact_gathered(const float* d_in, int world, int n_which, int n_units, int W, float* d_out, void* stream);
