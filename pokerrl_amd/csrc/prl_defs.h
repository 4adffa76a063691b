// pokerrl_amd native core -- shared definitions.
//
// One source tree, two builds:
//   * product build  (hipcc --offload-arch=gfx950): real HIP kernels for MI355X; the only build the package loads.
//   * PRL_EMU build  (g++ -DPRL_EMU, tests/emu only): the SAME kernel sources run through a tiny single-process SIMT
//     emulator so that kernel logic can be checked against the oracle in the GPU-less CI container. It is test
//     infrastructure; nothing in pokerrl_amd/ ever loads it.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(PRL_EMU)
#define PRL_HD
#define PRL_DEV
#define PRL_GLOBAL
#else
#include <hip/hip_runtime.h>
#define PRL_HD __host__ __device__
#define PRL_DEV __device__
#define PRL_GLOBAL __global__
#endif

#define PRL_INLINE inline __attribute__((always_inline))

// ---- game-wide constants (reference: PokerRL/game/Poker.py:7-46) ----------------------------------------------------
enum { PRL_FOLD = 0, PRL_CHECK_CALL = 1, PRL_BET_RAISE = 2 };
enum { PRL_PREFLOP = 0, PRL_FLOP = 1, PRL_TURN = 2, PRL_RIVER = 3 };
#define PRL_CARD_NOT_DEALT (-127)

// node kinds of the flat public tree
enum { PRL_NODE_DECISION = 0, PRL_NODE_CHANCE = 1, PRL_NODE_TERM_FOLD = 2, PRL_NODE_TERM_SHOWDOWN = 3 };

// CFR variants (reference: PokerRL/cfr/{VanillaCFR,CFRPlus,LinearCFR}.py)
enum { PRL_CFR_VANILLA = 0, PRL_CFR_PLUS = 1, PRL_CFR_LINEAR = 2 };

// status codes, PrlRules, PrlGame and every exported prototype live in the public C header
#include "pokerrl_hip.h"

#define PRL_MAX_BOARD_CARDS 5
#define PRL_WAVE 64
