// Error-checking glue around the HIP runtime for the C-ABI layer.
#pragma once
#include <string>

#include "prl_device.h"
#include "prl_host.h"

#define PRL_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t prl_e_ = (expr);                                                                \
        if (prl_e_ != hipSuccess) {                                                                \
            prl_set_error(std::string(#expr) + ": " + hipGetErrorString(prl_e_));                  \
            return PRL_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)
