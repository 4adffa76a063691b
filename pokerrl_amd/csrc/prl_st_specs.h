// internal: launch configuration shared by the per-spec translation units of the street kernels + their launchers
#pragma once
#include "prl_st.h"

#if defined(PRL_EMU)
#define ST_LB(t, w)
#else
#define ST_LB(t, w) __launch_bounds__(t, w)
#endif
// 768 lanes = 12 waves per CU, 3 per SIMD: a lane owns two adjacent hands (663 lanes hold the 1326 hands); in the last street's per-card
// scans a lane owns 3 list entries of one of 48 card slots (prl_fhp_kernels.hip has the same geometry)
#define ST_THREADS 768
#define FHP_SLOTS 2
#define ST_LAUNCH_BOUNDS ST_LB(768, 3)

#define PRL_ST_DECLARE_SPEC(ns)                                                                         \
    namespace ns {                                                                                      \
    int launch_down(const PrlStParams& prm, int src0, int src1, void* stream);                          \
    int launch_pass(bool last, const PrlStParams& prm, int mode, int src0, int src1, void* stream);     \
    }
PRL_ST_DECLARE_SPEC(st_spec9)
PRL_ST_DECLARE_SPEC(st_spec15)
PRL_ST_DECLARE_SPEC(st_spec21)
PRL_ST_DECLARE_SPEC(st_spec27)
PRL_ST_DECLARE_SPEC(st_spec33)
