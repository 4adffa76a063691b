// Host-callable launchers of the device kernels (defined in the .hip files).
#pragma once
#include "prl_defs.h"

// prl_handeval_kernels.hip
void prl_launch_hand_rank_boards(const int8_t* d_boards, int n_boards, const uint16_t* d_hole_lut, int32_t* d_out, void* stream);

// prl_tree_kernels.hip / prl_plan_kernels.hip (engine G: level-synchronous kernels over node vectors in HBM)
struct PrlDevTree;
struct PrlDevState;
void prl_launch_fill_uniform(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_col_node, void* stream);
void prl_launch_reach(const PrlDevTree& T, const PrlDevState& S, const int32_t* h_level_start, void* stream, bool root_is_set = false);
void prl_launch_terminals(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_term_nodes, int n_term, void* stream);
void prl_launch_ev_levels(const PrlDevTree& T, const PrlDevState& S, const int32_t* h_level_start, void* stream, float* d_expl_copy = nullptr);
void prl_launch_ev(const PrlDevTree& T, const PrlDevState& S, const int32_t* h_level_start, const int32_t* d_term_nodes, int n_term,
                   void* stream, float* d_expl_copy = nullptr);
void prl_launch_regret_strategy(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, int iter,
                                void* stream);
void prl_launch_average(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, int iter, int mode,
                        double m_old, double m_new, void* stream);
struct PrlIterDev;
struct PrlSmallJob;  // prl_solver_types.h
size_t prl_small_state_bytes(const PrlDevTree& T, const PrlDevState& S);
void prl_small_lds_plan(const PrlDevTree& T, const PrlDevState& S, int n_term, int n_nodes_p0, int n_nodes_p1, bool* state_in_lds, bool* tree_in_lds, size_t* bytes);
void prl_launch_small_iterations_many(const PrlSmallJob* d_jobs, int n_jobs, size_t lds_bytes, void* stream);
void prl_launch_small_iterations(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_level_start, const int32_t* d_term_nodes, int n_term,
                                 const int32_t* d_nodes_p0, int n0, const int32_t* d_nodes_p1, int n1, int variant, int delay, int n_iters,
                                 PrlIterDev* d_ip, void* stream);
void prl_launch_iter_begin(PrlIterDev* d_ip, int variant, int delay, void* stream);
void prl_launch_iter_end(PrlIterDev* d_ip, const float* d_expl, void* stream);
void prl_launch_regret_strategy_dev(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, const PrlIterDev* d_ip,
                                    void* stream);
void prl_launch_average_dev(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, const PrlIterDev* d_ip, void* stream);
void prl_launch_plan_build(const PrlDevTree& T, int n_plans, int16_t* plan_sh, int16_t* plan_pos, int16_t* plan_gs, int16_t* plan_ge,
                           int16_t* plan_cl, int32_t* plan_nlive, int16_t* plan_hgs, int16_t* plan_hge, uint32_t* plan_clx, int32_t* plan_ndealt, uint8_t* plan_klh,
                           int16_t* plan_pp, void* stream);
void prl_launch_hand_rank_checksums(const int8_t* d_boards, int n_boards, int chunk, const uint16_t* d_hole_lut, unsigned long long* d_out,
                                    void* stream);
