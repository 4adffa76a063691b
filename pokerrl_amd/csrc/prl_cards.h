// Card / hand index arithmetic shared by host and device code.
//
// Replaces the reference's binary-only lib_luts.so (call sites PokerRL/game/_/cpp_wrappers/CppLUT.py:38-47,73-94;
// layouts pinned by test/game/test_look_up_table.py:110-167):
//   card_1d = rank * n_suits + suit                      (test_look_up_table.py:110-116)
//   range_idx(c1 < c2) = lexicographic counter over all pairs (test_look_up_table.py:136-143)
#pragma once
#include "prl_defs.h"

PRL_HD PRL_INLINE int prl_card_1d(int rank, int suit, int n_suits) { return rank * n_suits + suit; }
PRL_HD PRL_INLINE int prl_card_rank(int c, int n_suits) { return c / n_suits; }
PRL_HD PRL_INLINE int prl_card_suit(int c, int n_suits) { return c % n_suits; }

// number of 2-card hands that precede (c1, *) in lexicographic order for a deck of n cards: sum_{i<c1} (n-1-i)
PRL_HD PRL_INLINE int prl_pair_row_offset(int c1, int n) { return c1 * (2 * n - c1 - 1) / 2; }

// (c1 < c2) -> range idx ; inverse of prl_hole_cards_2()
PRL_HD PRL_INLINE int prl_range_idx_2(int c1, int c2, int n) { return prl_pair_row_offset(c1, n) + (c2 - c1 - 1); }

// range idx -> (c1 < c2), closed form free: walks rows (n <= 52, <= 51 iterations)
PRL_HD PRL_INLINE void prl_hole_cards_2(int idx, int n, int* c1, int* c2) {
    int a = 0;
    int row = n - 1;
    while (idx >= row) {
        idx -= row;
        row--;
        a++;
    }
    *c1 = a;
    *c2 = a + 1 + idx;
}

PRL_HD PRL_INLINE long long prl_comb(int n, int k) {
    if (k < 0 || k > n) return 0;
    long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;  // exact at every step
    return r;
}

// hands of a range index as 1d cards (c2 = -1 for 1-card games)
PRL_HD PRL_INLINE void prl_hand_cards(const PrlRules& r, int idx, int* c1, int* c2) {
    if (r.n_hole_cards == 1) {
        *c1 = idx;
        *c2 = -1;
    } else {
        prl_hole_cards_2(idx, r.n_cards, c1, c2);
    }
}

// Counter-based decks for the batched engines (no reference counterpart: the reference shuffles with np.random, one hand at a time): the
// first n_deal cards of hand `hand_id` depend on (seed, hand_id) only. A partial Fisher-Yates shuffle of 0 .. n_cards-1 driven by a
// SplitMix-style hash: card d = what lies at position j = d + hash_d % (n_cards - d), position j then takes what lay at position d.
// n_deal <= 16, n_cards <= 65536.
//
// x % m of a 64-bit x without the 64-bit division (on the GPU a ~70-instruction scalar reciprocal chain and ~45 vector instructions per card):
// with inv = floor((2^32 - 1) / m), q' = mulhi(v, inv) is floor(v / m) or one less for every 32-bit v (v / m - v inv / 2^32 =
// v (2^32 - inv m) / (m 2^32) <= v / 2^32 < 1 because 2^32 - inv m <= m), so one conditional subtraction gives v % m; the two halves of x are
// joined through 2^32 % m = (2^32 - inv m) % m. m <= 65536 keeps (hi % m)(2^32 % m) + lo % m inside 32 bits.
struct PrlSmallMod { uint32_t m, inv, pow32; };
PRL_HD PRL_INLINE PrlSmallMod prl_small_mod(uint32_t m) {
    PrlSmallMod k;
    k.m = m;
    k.inv = 0xFFFFFFFFu / m;
    const uint32_t t = 0u - k.inv * m;  // 2^32 - inv m, in 1 .. m
    k.pow32 = t == m ? 0u : t;
    return k;
}
PRL_HD PRL_INLINE uint32_t prl_mod32(uint32_t v, const PrlSmallMod& k) {
    const uint32_t r = v - (uint32_t)(((unsigned long long)v * k.inv) >> 32) * k.m;
    return r >= k.m ? r - k.m : r;
}
PRL_HD PRL_INLINE uint32_t prl_mod64(unsigned long long x, const PrlSmallMod& k) {
    return prl_mod32(prl_mod32((uint32_t)(x >> 32), k) * k.pow32 + prl_mod32((uint32_t)x, k), k);
}
// The deck as a sparse permutation: step k leaves ONE remembered move (position j_k now holds what lay at position k; positions below the
// current card are never looked at again), and "what lies at position p now" is the latest move onto p, else p itself. The moves live in two
// fully unrolled register arrays scanned by selects -- no array indexed at run time, no loop whose trip count differs between lanes (round 6;
// before: an in-place update list walked by a data-dependent loop, `s_set_gpr_idx` indexing on the GPU).
PRL_HD PRL_INLINE void prl_deal_hand(int n_cards, int n_deal, unsigned long long seed, unsigned long long hand_id, int8_t* out) {
    const unsigned long long idx = (hand_id + 1ull) * 0x9E3779B97F4A7C15ull + seed;
    int pos[16], val[16];
#if defined(__clang__)
#pragma unroll
#endif
    for (int d = 0; d < 16; ++d) {
        if (d >= n_deal) continue;  // (not `break`: a loop with one exit unrolls)
        unsigned long long x = idx + (unsigned long long)(d + 1) * 0xBF58476D1CE4E5B9ull;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27; x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
        const int j = d + (int)prl_mod64(x, prl_small_mod((uint32_t)(n_cards - d)));
        int a = d, b = j;
#if defined(__clang__)
#pragma unroll
#endif
        for (int k = 0; k < d; ++k) { a = pos[k] == d ? val[k] : a; b = pos[k] == j ? val[k] : b; }
        pos[d] = j != d ? j : -1;
        val[d] = a;
        out[d] = (int8_t)b;
    }
}
