// Internal (non-exported) host helpers shared by the C-ABI translation units.
#pragma once
#include <string>

#include "prl_defs.h"
#include "prl_tree.h"

void prl_set_error(const std::string& msg);
extern "C" const PrlFlatTree* prl_tree_flat(const prl_tree_t* tree);
