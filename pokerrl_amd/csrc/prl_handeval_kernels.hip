// Batched 7-card evaluation: boards[N][5] -> ranks[N][1326] (int32, -1 where a hole card is on the board).
// Replaces get_hand_rank_all_hands_on_given_boards_52_holdem (lib_hand_eval.so; CppHandeval.py:45-65), the inner kernel
// of LBR (LocalLBRWorker.py:406-425) and the source of the per-board showdown order of the FHP public tree.
//
// Mapping: one workgroup per board, 256 lanes stride over the 1326 hands (hand index fastest -> coalesced 4-byte
// stores, the only HBM traffic worth mentioning: 4 B per evaluation; inputs are 5 B per 1326 evaluations plus a 2.6 KB
// hole-card table that lives in L1/L2). The evaluation itself is ~100 integer ops on four 13-bit masks (prl_handeval.h),
// no LDS, no tables -- nothing here is a contraction, so no MFMA.
#include "prl_device.h"
#include "prl_handeval.h"
#include "prl_kernels.h"

PRL_GLOBAL void prl_k_hand_rank_boards(const int8_t* __restrict__ boards, int n_boards, const uint16_t* __restrict__ hole_lut,
                                       int32_t* __restrict__ out) {
    for (int b = (int)prl_bid(); b < n_boards; b += (int)prl_nblocks()) {
        uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        unsigned long long on_board = 0ull;
        for (int i = 0; i < 5; ++i) {
            int c = boards[(size_t)b * 5 + i];
            uint32_t bit = 1u << (c >> 2);
            int su = c & 3;
            s0 |= su == 0 ? bit : 0u;
            s1 |= su == 1 ? bit : 0u;
            s2 |= su == 2 ? bit : 0u;
            s3 |= su == 3 ? bit : 0u;
            on_board |= 1ull << c;
        }
        int32_t* row = out + (size_t)b * 1326;
        for (int h = (int)prl_tid(); h < 1326; h += (int)prl_nthreads()) {
            uint32_t cc = hole_lut[h];
            int c1 = (int)(cc & 0xFFu), c2 = (int)(cc >> 8);
            int32_t r = -1;
            if (!((on_board >> c1) & 1ull) && !((on_board >> c2) & 1ull)) {
                uint32_t b1 = 1u << (c1 >> 2), b2 = 1u << (c2 >> 2);
                int u1 = c1 & 3, u2 = c2 & 3;
                r = prl_rank7_masks(s0 | (u1 == 0 ? b1 : 0u) | (u2 == 0 ? b2 : 0u), s1 | (u1 == 1 ? b1 : 0u) | (u2 == 1 ? b2 : 0u),
                                    s2 | (u1 == 2 ? b1 : 0u) | (u2 == 2 ? b2 : 0u), s3 | (u1 == 3 ? b1 : 0u) | (u2 == 3 ? b2 : 0u));
            }
            row[h] = r;
        }
    }
}

void prl_launch_hand_rank_boards(const int8_t* d_boards, int n_boards, const uint16_t* d_hole_lut, int32_t* d_out, void* stream) {
    if (n_boards <= 0) return;
    // >> 256 workgroups fill the 256 CUs; cap the grid and stride so a 2.6 M-board sweep does not queue 2.6 M blocks
    int grid = n_boards < 65536 ? n_boards : 65536;
    PRL_LAUNCH(prl_k_hand_rank_boards, grid, 256, 0, stream, d_boards, n_boards, d_hole_lut, d_out);
}

// ---------------------------------------------------------------------------------------------------------------------
// Verification aid: order-sensitive 64-bit checksum of the rank table, one value per chunk of `chunk` boards, computed
// without materialising the table (an exhaustive C(52,5) x 1326 sweep would be 13.8 GB). Same definition as
// tests/golden/make_golden.py:_board_checksum, which was run over the reference binary for all 2,598,960 boards:
//   per_board(b) = sum_h (rank[b,h] + 2) * (h * 2654435761 + 12345)         (mod 2^64)
//   chunk        = sum_{b in chunk} per_board(b) * (b_local * 0x9E3779B97F4A7C15 + 1)
// ---------------------------------------------------------------------------------------------------------------------
PRL_GLOBAL void prl_k_hand_rank_checksums(const int8_t* __restrict__ boards, int n_boards, int chunk, const uint16_t* __restrict__ hole_lut,
                                          unsigned long long* __restrict__ out) {
    unsigned long long* red = (unsigned long long*)prl_smem();  // [nthreads]
    const int n_chunks = (n_boards + chunk - 1) / chunk;
    for (int ck = (int)prl_bid(); ck < n_chunks; ck += (int)prl_nblocks()) {
        unsigned long long acc = 0ull;
        const int b_lo = ck * chunk, b_hi = b_lo + chunk < n_boards ? b_lo + chunk : n_boards;
        for (int b = b_lo; b < b_hi; ++b) {
            uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            unsigned long long on_board = 0ull;
            for (int i = 0; i < 5; ++i) {
                int c = boards[(size_t)b * 5 + i];
                uint32_t bit = 1u << (c >> 2);
                int su = c & 3;
                s0 |= su == 0 ? bit : 0u;
                s1 |= su == 1 ? bit : 0u;
                s2 |= su == 2 ? bit : 0u;
                s3 |= su == 3 ? bit : 0u;
                on_board |= 1ull << c;
            }
            unsigned long long per = 0ull;
            for (int h = (int)prl_tid(); h < 1326; h += (int)prl_nthreads()) {
                uint32_t cc = hole_lut[h];
                int c1 = (int)(cc & 0xFFu), c2 = (int)(cc >> 8);
                long long r = -1;
                if (!((on_board >> c1) & 1ull) && !((on_board >> c2) & 1ull)) {
                    uint32_t b1 = 1u << (c1 >> 2), b2 = 1u << (c2 >> 2);
                    int u1 = c1 & 3, u2 = c2 & 3;
                    r = prl_rank7_masks(s0 | (u1 == 0 ? b1 : 0u) | (u2 == 0 ? b2 : 0u), s1 | (u1 == 1 ? b1 : 0u) | (u2 == 1 ? b2 : 0u),
                                        s2 | (u1 == 2 ? b1 : 0u) | (u2 == 2 ? b2 : 0u), s3 | (u1 == 3 ? b1 : 0u) | (u2 == 3 ? b2 : 0u));
                }
                per += (unsigned long long)(r + 2) * ((unsigned long long)h * 2654435761ull + 12345ull);
            }
            acc += per * ((unsigned long long)(b - b_lo) * 0x9E3779B97F4A7C15ull + 1ull);
        }
        red[prl_tid()] = acc;
        prl_sync();
        for (unsigned s = prl_nthreads() >> 1; s > 0; s >>= 1) {  // integer adds mod 2^64: order-free
            if (prl_tid() < s) red[prl_tid()] += red[prl_tid() + s];
            prl_sync();
        }
        if (prl_tid() == 0) out[ck] = red[0];
        prl_sync();
    }
}

void prl_launch_hand_rank_checksums(const int8_t* d_boards, int n_boards, int chunk, const uint16_t* d_hole_lut, unsigned long long* d_out,
                                    void* stream) {
    if (n_boards <= 0) return;
    int n_chunks = (n_boards + chunk - 1) / chunk;
    PRL_LAUNCH(prl_k_hand_rank_checksums, n_chunks < 65536 ? n_chunks : 65536, 256, 256 * sizeof(unsigned long long), stream, d_boards,
               n_boards, chunk, d_hole_lut, d_out);
}
