// Host-callable launchers of the device kernels (defined in the .hip files).
#pragma once
#include "prl_defs.h"

// prl_handeval_kernels.hip
void prl_launch_hand_rank_boards(const int8_t* d_boards, int n_boards, const uint16_t* d_hole_lut, int32_t* d_out, void* stream);
