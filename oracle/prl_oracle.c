/*
 * prl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, not product code).
 *
 * A plain-C restatement of the reference's tabular hot path, written to mirror the reference's own structure (recursive
 * walks over tree nodes, one function per reference function) so that it can be read side by side with it:
 *
 *   hand evaluator      lib_hand_eval.so (binary only; spec SURVEY.md 2.2)        -> orc_rank7()
 *   reach push-down     PokerRL/game/_/tree/_/StrategyFiller.py:118-146           -> update_reach()
 *   chance strategy     StrategyFiller.py:148-169                                 -> chance weight in update_reach()
 *   uniform fill        StrategyFiller.py:48-65                                   -> orc_fill_uniform()
 *   EV / BR pull-up     PokerRL/game/_/tree/_/ValueFiller.py:21-101               -> compute_ev()
 *   terminal equity     ValueFiller.py:103-175                                    -> fold_equity_1card(), showdown_*()
 *   regrets             PokerRL/cfr/_CFRBase.py:146-185 + variant formulas        -> compute_regrets()
 *   regret matching     VanillaCFR.py:32-52, CFRPlus.py:43-63, LinearCFR.py:33-51 -> compute_new_strategy()
 *   average strategy    VanillaCFR.py:54-77, CFRPlus.py:65-87, LinearCFR.py:53-76 -> add_strategy_to_average()
 *   iteration order     _CFRBase.py:122-134                                       -> orc_cfr_iteration()
 *   avg-strategy eval   _CFRBase.py:218-262                                       -> orc_eval_avg()
 *
 * Numerics follow the reference under NumPy 2.2.6 bit for bit on 1-hole-card games (SURVEY.md 8a dtype ledger and
 * Appendix A): float32 storage, float64 islands where a float64 strategy meets float32 data, NumPy's pairwise order for
 * inner-axis sums, sequential order for outer-axis sums, separate multiply and add (build with -ffp-contract=off).
 * It is pinned against fixtures captured from the reference itself (tests/golden/cfr_*.npz, handrank*.npz).
 *
 * For 2-hole-card games (Flop5Holdem) the reference cannot run (ValueFiller.py:19,57-59,145-175 are 1-card only), so the
 * terminal equity here is the generalisation of SURVEY.md Appendix C with an EXPLICIT float32 summation order (wave-64
 * Hillis-Steele scans over the rank-sorted range + per-card blocker lists), which the HIP kernels reproduce exactly.
 * That part is "pinned to the restatement" and cross-checked against the O(R^2) definition (orc_showdown_bruteforce).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRL_ORC_MAX_ACTIONS 128

enum { K_DECISION = 0, K_CHANCE = 1, K_FOLD = 2, K_SHOWDOWN = 3 };
enum { V_VANILLA = 0, V_PLUS = 1, V_LINEAR = 2 };

/* ------------------------------------------------------------------------------------------------------------------ */
/* hand evaluator: max over the 21 five-card subsets + the binary's quads-kicker quirk                  */
/* ------------------------------------------------------------------------------------------------------------------ */
static int rank5(const int r[5], const int s[5]) {
    int v[5], i, j;
    for (i = 0; i < 5; ++i) v[i] = r[i];
    for (i = 0; i < 5; ++i)
        for (j = i + 1; j < 5; ++j)
            if (v[j] > v[i]) { int t = v[i]; v[i] = v[j]; v[j] = t; }
    int flush = s[0] == s[1] && s[1] == s[2] && s[2] == s[3] && s[3] == s[4];
    int distinct = v[0] != v[1] && v[1] != v[2] && v[2] != v[3] && v[3] != v[4];
    int straight_top = -1;
    if (distinct) {
        if (v[0] - v[4] == 4) straight_top = v[0];
        else if (v[0] == 12 && v[1] == 3 && v[2] == 2 && v[3] == 1 && v[4] == 0) straight_top = 3; /* wheel */
    }
    if (flush && straight_top >= 0) return 1240827 + straight_top;
    int cnt[13];
    memset(cnt, 0, sizeof cnt);
    for (i = 0; i < 5; ++i) cnt[v[i]]++;
    int four = -1, three = -1, p1 = -1, p2 = -1;
    for (i = 12; i >= 0; --i) {
        if (cnt[i] == 4) four = i;
        else if (cnt[i] == 3) three = i;
        else if (cnt[i] == 2) { if (p1 < 0) p1 = i; else p2 = i; }
    }
    if (four >= 0) { int k = -1; for (i = 12; i >= 0; --i) if (cnt[i] == 1) k = i; return 1240618 + 13 * four + k; }
    if (three >= 0 && p1 >= 0) return 1240409 + 13 * three + p1;
    if (flush) return 664398 + (((v[0] * 13 + v[1]) * 13 + v[2]) * 13 + v[3]) * 13 + v[4];
    if (straight_top >= 0) return 664384 + straight_top;
    if (three >= 0) { int k[2], n = 0; for (i = 12; i >= 0; --i) if (cnt[i] == 1) k[n++] = i; return 661446 + 169 * three + 13 * k[0] + k[1]; }
    if (p2 >= 0) { int k = -1; for (i = 12; i >= 0; --i) if (cnt[i] == 1) k = i; return 658508 + 169 * p1 + 13 * p2 + k; }
    if (p1 >= 0) { int k[3], n = 0; for (i = 12; i >= 0; --i) if (cnt[i] == 1) k[n++] = i; return 576011 + 2197 * p1 + 169 * k[0] + 13 * k[1] + k[2]; }
    return (((v[0] * 13 + v[1]) * 13 + v[2]) * 13 + v[3]) * 13 + v[4];
}

int32_t orc_rank7(const int8_t* board5, int c1, int c2) {
    int r[7], s[7], i, j;
    for (i = 0; i < 5; ++i) { r[i] = board5[i] >> 2; s[i] = board5[i] & 3; }
    r[5] = c1 >> 2; s[5] = c1 & 3;
    r[6] = c2 >> 2; s[6] = c2 & 3;
    /* quads: kicker = neighbour of the four-of-a-kind in the rank-sorted seven cards */
    int sr[7];
    for (i = 0; i < 7; ++i) sr[i] = r[i];
    for (i = 0; i < 7; ++i)
        for (j = i + 1; j < 7; ++j)
            if (sr[j] > sr[i]) { int t = sr[i]; sr[i] = sr[j]; sr[j] = t; }
    for (i = 0; i + 3 < 7; ++i)
        if (sr[i] == sr[i + 3]) {
            int kicker = i > 0 ? sr[i - 1] : sr[4];
            return 1240618 + 13 * sr[i] + kicker;
        }
    int best = -1, a, b;
    for (a = 0; a < 7; ++a)
        for (b = a + 1; b < 7; ++b) { /* leave out cards a and b */
            int rr[5], ss[5], n = 0;
            for (i = 0; i < 7; ++i)
                if (i != a && i != b) { rr[n] = r[i]; ss[n] = s[i]; n++; }
            int v = rank5(rr, ss);
            if (v > best) best = v;
        }
    return best;
}

void orc_rank_boards(const int8_t* boards, int n_boards, int32_t* out /* [n][1326] */) {
    for (int b = 0; b < n_boards; ++b) {
        const int8_t* bd = boards + 5 * b;
        int idx = 0;
        for (int c1 = 0; c1 < 52; ++c1)
            for (int c2 = c1 + 1; c2 < 52; ++c2, ++idx) {
                int blocked = 0;
                for (int i = 0; i < 5; ++i) blocked |= (bd[i] == c1) | (bd[i] == c2);
                out[(size_t)b * 1326 + idx] = blocked ? -1 : orc_rank7(bd, c1, c2);
            }
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* summation orders                                                                                                    */
/* ------------------------------------------------------------------------------------------------------------------ */
/* NumPy's float32 pairwise sum along a contiguous axis (numpy/_core/src/umath/loops_utils.h.src), SURVEY Appendix A */
static float np_sum_f32(const float* a, int n, int stride) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = res + a[i * stride];
        return res;
    }
    if (n <= 128) {
        float r[8];
        int i, j;
        for (j = 0; j < 8; ++j) r[j] = a[j * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (j = 0; j < 8; ++j) r[j] = r[j] + a[(i + j) * stride];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i * stride];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum_f32(a, n2, stride) + np_sum_f32(a + n2 * stride, n - n2, stride);
}

static double np_sum_f64(const double* a, int n, int stride) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; ++i) res = res + a[i * stride];
        return res;
    }
    if (n <= 128) {
        double r[8];
        int i, j;
        for (j = 0; j < 8; ++j) r[j] = a[j * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (j = 0; j < 8; ++j) r[j] = r[j] + a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i * stride];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum_f64(a, n2, stride) + np_sum_f64(a + n2 * stride, n - n2, stride);
}

/* canonical wave-64 inclusive scan (DESIGN.md "summation order"), all lanes in lock-step: Hillis-Steele inside each row of
 * 16 lanes (d = 1, 2, 4, 8), then rows 1 and 3 add lane 15 of the row below, then lanes 32..63 add lane 31 -- the
 * association of the six-DPP-op scan of GCN/CDNA hardware (csrc/prl_device.h: prl_wave_scan_canonical) */
static void scan64(float v[64]) {
    float t[64];
    for (int d = 1; d < 16; d <<= 1) {
        for (int l = 0; l < 64; ++l) t[l] = (l & 15) >= d ? v[l - d] : 0.f;
        for (int l = 0; l < 64; ++l) v[l] = v[l] + t[l];
    }
    for (int l = 0; l < 64; ++l) t[l] = ((l >> 4) & 1) ? v[(l >> 4) * 16 - 1] : 0.f;
    for (int l = 0; l < 64; ++l) v[l] = v[l] + t[l];
    for (int l = 0; l < 64; ++l) t[l] = l >= 32 ? v[31] : 0.f;
    for (int l = 0; l < 64; ++l) v[l] = v[l] + t[l];
}

/* exclusive prefix P[0..n] of y[0..n) in the canonical chunked order: 64-wide scans + sequential chunk carries */
static void prefix_chunked(const float* y, int n, float* P /* n+1 */) {
    float carry = 0.f;
    int n_chunks = (n + 63) / 64;
    for (int k = 0; k < n_chunks; ++k) {
        float v[64];
        for (int l = 0; l < 64; ++l) v[l] = (64 * k + l < n) ? y[64 * k + l] : 0.f;
        scan64(v);
        for (int l = 0; l < 64; ++l) {
            int j = 64 * k + l;
            if (j <= n) P[j] = l == 0 ? carry : carry + v[l - 1];
        }
        carry = carry + v[63];
    }
    if (n % 64 == 0) P[n] = carry;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* context                                                                                                             */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int n_live;        /* hands not blocked by the board */
    int16_t* sh;       /* [n_live] hand index at sorted position */
    int16_t* pos;      /* [R] sorted position of a hand, -1 if blocked */
    int16_t* gs;       /* [n_live] first position of the tie group */
    int16_t* ge;       /* [n_live] one past the last position of the tie group */
    int n_t;           /* entries per card list */
    int16_t* cl;       /* [n_cards][n_t] sorted positions of the hands containing card c (ascending); -1 rows for board cards */
} Plan;

typedef struct {
    int n_nodes, n_cols, R, n_hole, n_cards, n_suits, rank_rule, n_boards, board_len;
    const int32_t *kind, *actor, *parent, *child_idx, *action, *acted_last, *round, *board_id, *main_pot, *n_children,
        *first_col, *child_start, *child_list;
    const int8_t* boards;
    float chance_prob, eq_const;
    float* chance_w;      /* [n_nodes] weight of every outcome of a chance node for a hand it does not block (0 elsewhere) */
    /* weighted boards / suit isomorphism (include/pokerrl_hip.h: prl_solver_create_weighted; no reference counterpart -- the reference lists every board,
     * PublicTree.py:188-210): per-child weight of the one chance node = chance_prob x (boards the child stands for), and the chance node's values
     * averaged over every hand's suit orbit */
    float* board_w;       /* [children of the chance node] or NULL */
    int32_t* sym_class;   /* [R] class of a hand under suit permutations, or NULL */
    int16_t* hole;        /* [R][2] */
    /* state */
    double* strategy;     /* [n_cols][R] */
    uint8_t* strat_f64;   /* [n_nodes] */
    float *reach, *ev, *ev_br;   /* [n_nodes][2][R] */
    int32_t* br_idx;      /* [n_nodes][R] */
    float* regret;        /* [n_cols][R] */
    float* avg_sum;       /* [n_cols][R] */
    double* avg;          /* [n_cols][R] */
    uint8_t* avg_f64;     /* [n_nodes] */
    float expl[2];
    int variant, delay, iter;
    int unsupported;      /* tree holds a terminal this restatement does not cover */
    /* chunked runs of trees too big for one oracle instance (tests/golden/make_fhp_golden_chunked.py): the values of ONE node given from
     * outside, its subtree skipped -- the chance node of a Flop5Holdem tree whose boards are evaluated chunk by chunk */
    int ov_node;
    const float *ov_ev, *ov_ev_br;
    Plan* plans;          /* [n_boards + 1], last = "no board" identity plan; built lazily */
    uint8_t* plan_ready;
    int32_t* tmp_ranks;
} Orc;

#define V2(o, arr, node, p) ((o)->arr + ((size_t)(node) * 2 + (p)) * (o)->R)
#define COL(o, arr, col) ((o)->arr + (size_t)(col) * (o)->R)

static int child_of(const Orc* o, int node, int i) { return o->child_list[o->child_start[node] + i]; }

static int hand_blocked(const Orc* o, int h, const int8_t* board, int n) {
    for (int i = 0; i < n; ++i)
        if (board[i] >= 0 && (board[i] == o->hole[2 * h] || board[i] == o->hole[2 * h + 1])) return 1;
    return 0;
}

Orc* orc_create(int n_nodes, int n_cols, int R, int n_hole, int n_cards, int n_suits, int rank_rule, int n_boards,
                int board_len, const int32_t* kind, const int32_t* actor, const int32_t* parent, const int32_t* child_idx,
                const int32_t* action, const int32_t* acted_last, const int32_t* round, const int32_t* board_id,
                const int32_t* main_pot, const int32_t* n_children, const int32_t* first_col, const int32_t* child_start,
                const int32_t* child_list, const int8_t* boards, float chance_prob, float eq_const) {
    Orc* o = (Orc*)calloc(1, sizeof(Orc));
    o->n_nodes = n_nodes; o->n_cols = n_cols; o->R = R; o->n_hole = n_hole; o->n_cards = n_cards; o->n_suits = n_suits;
    o->rank_rule = rank_rule; o->n_boards = n_boards; o->board_len = board_len;
#define DUP(name, count, type) { type* p_ = (type*)malloc(sizeof(type) * (size_t)((count) > 0 ? (count) : 1)); memcpy(p_, name, sizeof(type) * (size_t)(count)); o->name = p_; }
    DUP(kind, n_nodes, int32_t) DUP(actor, n_nodes, int32_t) DUP(parent, n_nodes, int32_t) DUP(child_idx, n_nodes, int32_t)
    DUP(action, n_nodes, int32_t) DUP(acted_last, n_nodes, int32_t) DUP(round, n_nodes, int32_t) DUP(board_id, n_nodes, int32_t)
    DUP(main_pot, n_nodes, int32_t) DUP(n_children, n_nodes, int32_t) DUP(first_col, n_nodes, int32_t)
    DUP(child_start, n_nodes + 1, int32_t) DUP(child_list, n_nodes - 1, int32_t) DUP(boards, n_boards * board_len, int8_t)
#undef DUP
    o->chance_prob = chance_prob;
    o->eq_const = eq_const;
    o->chance_w = (float*)calloc((size_t)n_nodes, sizeof(float));
    for (int n = 0; n < n_nodes; ++n)
        if (kind[n] == K_CHANCE) o->chance_w[n] = chance_prob; /* one dealing round; orc_set_chance_weights overrides per node */
    /* all-in before the deal on a 2-card tree (ValueFiller.py:160-175 generalised) is not restated: refuse the tree */
    if (n_hole == 2)
        for (int n = 0; n < n_nodes; ++n)
            if (kind[n] == K_SHOWDOWN && board_id[n] < 0) { o->unsupported = 1; }
    o->hole = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)R);
    if (n_hole == 1) {
        for (int h = 0; h < R; ++h) { o->hole[2 * h] = (int16_t)h; o->hole[2 * h + 1] = -1; }
    } else {
        int idx = 0;
        for (int a = 0; a < n_cards; ++a)
            for (int b = a + 1; b < n_cards; ++b, ++idx) { o->hole[2 * idx] = (int16_t)a; o->hole[2 * idx + 1] = (int16_t)b; }
    }
    size_t nv = (size_t)n_nodes * 2 * R, nc = (size_t)n_cols * R;
    o->strategy = (double*)calloc(nc, sizeof(double));
    o->strat_f64 = (uint8_t*)calloc(n_nodes, 1);
    o->reach = (float*)calloc(nv, sizeof(float));
    o->ev = (float*)calloc(nv, sizeof(float));
    o->ev_br = (float*)calloc(nv, sizeof(float));
    o->br_idx = (int32_t*)calloc((size_t)n_nodes * R, sizeof(int32_t));
    o->regret = (float*)calloc(nc, sizeof(float));
    o->avg_sum = (float*)calloc(nc, sizeof(float));
    o->avg = (double*)calloc(nc, sizeof(double));
    o->avg_f64 = (uint8_t*)calloc(n_nodes, 1);
    o->plans = (Plan*)calloc((size_t)n_boards + 1, sizeof(Plan));
    o->plan_ready = (uint8_t*)calloc((size_t)n_boards + 1, 1);
    o->tmp_ranks = (int32_t*)malloc(sizeof(int32_t) * (size_t)R);
    return o;
}

static void plan_free(Plan* p) { free(p->sh); free(p->pos); free(p->gs); free(p->ge); free(p->cl); }

void orc_destroy(Orc* o) {
    if (!o) return;
    for (int b = 0; b <= o->n_boards; ++b)
        if (o->plan_ready[b]) plan_free(&o->plans[b]);
    free((void*)o->kind); free((void*)o->actor); free((void*)o->parent); free((void*)o->child_idx); free((void*)o->action);
    free((void*)o->acted_last); free((void*)o->round); free((void*)o->board_id); free((void*)o->main_pot);
    free((void*)o->n_children); free((void*)o->first_col); free((void*)o->child_start); free((void*)o->child_list);
    free((void*)o->boards);
    free(o->chance_w); free(o->board_w); free(o->sym_class);
    free(o->hole); free(o->strategy); free(o->strat_f64); free(o->reach); free(o->ev); free(o->ev_br); free(o->br_idx);
    free(o->regret); free(o->avg_sum); free(o->avg); free(o->avg_f64); free(o->plans); free(o->plan_ready); free(o->tmp_ranks);
    free(o);
}

/* hand strength of every hand on a board (game_rules.py:68-75,133-140,213-223) */
static void ranks_on_board(const Orc* o, const int8_t* board, int32_t* out) {
    for (int h = 0; h < o->R; ++h) {
        if (o->n_hole == 1) {
            int hr = h / o->n_suits, br = board[0] / o->n_suits;
            int bonus = o->rank_rule == 1 ? 10000 : 100;
            out[h] = hr == br ? bonus + hr : hr;
        } else {
            out[h] = hand_blocked(o, h, board, o->board_len) ? -1 : orc_rank7(board, o->hole[2 * h], o->hole[2 * h + 1]);
        }
    }
}

/* showdown plan of a board for 2-card games: live hands sorted by (rank, hand index), tie groups, per-card lists */
static const Plan* get_plan(Orc* o, int board_id) {
    int slot = board_id < 0 ? o->n_boards : board_id;
    if (o->plan_ready[slot]) return &o->plans[slot];
    Plan* p = &o->plans[slot];
    const int R = o->R;
    const int8_t* board = board_id < 0 ? NULL : o->boards + (size_t)board_id * o->board_len;
    int32_t* rk = (int32_t*)malloc(sizeof(int32_t) * (size_t)R); /* plans of different boards are built by different threads */
    /* a row of the board table may be a PREFIX (cards not dealt yet are -1): like "no board" it has no hand ranks -- hand-index
     * order, one tie group -- but the hands it blocks are out (fold terminals before the last street) */
    int n_dealt = 0;
    if (board) for (int i = 0; i < o->board_len; ++i) n_dealt += board[i] >= 0;
    if (board && n_dealt == o->board_len) ranks_on_board(o, board, rk);
    else for (int h = 0; h < R; ++h) rk[h] = (board && hand_blocked(o, h, board, o->board_len)) ? -1 : 0;
    p->sh = (int16_t*)malloc(sizeof(int16_t) * R);
    p->pos = (int16_t*)malloc(sizeof(int16_t) * R);
    p->gs = (int16_t*)malloc(sizeof(int16_t) * R);
    p->ge = (int16_t*)malloc(sizeof(int16_t) * R);
    int n = 0;
    for (int h = 0; h < R; ++h) p->pos[h] = -1;
    for (int h = 0; h < R; ++h)
        if (rk[h] >= 0) p->sh[n++] = (int16_t)h;
    /* stable insertion sort by rank keeps hand-index order inside a tie group */
    for (int i = 1; i < n; ++i) {
        int16_t x = p->sh[i];
        int j = i - 1;
        while (j >= 0 && rk[p->sh[j]] > rk[x]) { p->sh[j + 1] = p->sh[j]; j--; }
        p->sh[j + 1] = x;
    }
    p->n_live = n;
    for (int i = 0; i < n; ++i) p->pos[p->sh[i]] = (int16_t)i;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && rk[p->sh[j]] == rk[p->sh[i]]) j++;
        for (int k = i; k < j; ++k) { p->gs[k] = (int16_t)i; p->ge[k] = (int16_t)j; }
        i = j;
    }
    p->n_t = o->n_cards - 1 - n_dealt;
    p->cl = (int16_t*)malloc(sizeof(int16_t) * (size_t)o->n_cards * (o->n_cards - 1));
    for (int c = 0; c < o->n_cards; ++c) {
        int16_t* row = p->cl + (size_t)c * (o->n_cards - 1);
        int m = 0;
        for (int i = 0; i < n; ++i) { /* ascending sorted position */
            int h = p->sh[i];
            if (o->hole[2 * h] == c || o->hole[2 * h + 1] == c) row[m++] = (int16_t)i;
        }
        for (int k = m; k < o->n_cards - 1; ++k) row[k] = -1;
    }
    free(rk);
    o->plan_ready[slot] = 1;
    return p;
}

/* all plans up front (in parallel): afterwards get_plan() only reads, so board subtrees can be evaluated concurrently */
static void ensure_plans(Orc* o) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int b = 0; b <= o->n_boards; ++b)
        if (!o->plan_ready[b]) get_plan(o, b == o->n_boards ? -1 : b);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* terminal equity                                                                                                     */
/* ------------------------------------------------------------------------------------------------------------------ */
/* ValueFiller._get_fold_eq_* (ValueFiller.py:103-125), 1-card: sum(reach[opp]) - reach[opp] */
static void fold_equity_1card(const Orc* o, const float* reach_opp, float* eq) {
    float total = np_sum_f32(reach_opp, o->R, 1);
    for (int h = 0; h < o->R; ++h) eq[h] = total - reach_opp[h];
}

/* ValueFiller._get_call_eq_final_street (ValueFiller.py:127-158), 1-card: O(R^2), ascending h_opp, float32 running sum */
static void showdown_equity_1card(const Orc* o, const float* reach_opp, int board_card, float* eq) {
    int8_t bc = (int8_t)board_card;
    int32_t rk[256]; /* 1-card ranges: R = number of cards (<= 52) */
    ranks_on_board(o, &bc, rk);
    for (int h = 0; h < o->R; ++h) {
        float e = 0.f;
        if (h != board_card)
            for (int ho = 0; ho < o->R; ++ho)
                if (ho != h && ho != board_card) {
                    if (rk[h] > rk[ho]) e = e + reach_opp[ho];
                    else if (rk[h] < rk[ho]) e = e - reach_opp[ho];
                }
        eq[h] = e;
    }
}

/* per-card scan of the sorted-domain vector y restricted to the hands containing card c, in the canonical "row16" order
 * (csrc/prl_device.h: prl_row16_scan): the list, zero-padded to 16 * E entries with E = ceil(n_t / 16), is owned by 16 lanes,
 * E consecutive entries each; sequential prefix inside a lane, Hillis-Steele scan (d = 1, 2, 4, 8) of the lane totals, and
 * Q_incl[i*E + k] = carry_i + local_k with carry_i = the scanned total of lane i-1 (0 for lane 0). Q[0] = 0, Q[t] = Q_incl[t-1]. */
static void card_scan(const Plan* p, const Orc* o, const float* y, int c, float Q[65]) {
    const int16_t* row = p->cl + (size_t)c * (o->n_cards - 1);
    const int E = (p->n_t + 15) / 16;
    float x[64], l[64], tot[16], tmp[16];
    for (int t = 0; t < 64; ++t) x[t] = (t < p->n_t && row[t] >= 0) ? y[row[t]] : 0.f;
    for (int i = 0; i < 16; ++i) {
        float run = 0.f;
        for (int k = 0; k < E; ++k) {
            run = k == 0 ? x[i * E + k] : run + x[i * E + k];
            l[i * E + k] = run;
        }
        tot[i] = run;
    }
    for (int d = 1; d < 16; d <<= 1) {
        for (int i = 0; i < 16; ++i) tmp[i] = i >= d ? tot[i - d] : 0.f;
        for (int i = 0; i < 16; ++i) tot[i] = tot[i] + tmp[i];
    }
    for (int t = 0; t <= 64; ++t) Q[t] = 0.f;
    for (int i = 0; i < 16; ++i) {
        const float carry = i > 0 ? tot[i - 1] : 0.f;
        for (int k = 0; k < E; ++k) Q[i * E + k + 1] = carry + l[i * E + k];
    }
}

/* 2-card terminal equity in the canonical scan order. mode 0 = fold, 1 = showdown. x = opponent reach (hand domain). */
static void terminal_equity_2card(Orc* o, const float* x, int board_id, int mode, float* eq) {
    const Plan* p = get_plan(o, board_id);
    const int R = o->R, n = p->n_live;
    float* y = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* P = (float*)malloc(sizeof(float) * (size_t)(n + 2));
    for (int i = 0; i < n; ++i) y[i] = x[p->sh[i]];
    prefix_chunked(y, n, P);
    const float T = P[n];
    float(*Q)[65] = (float(*)[65])malloc(sizeof(float) * 65 * (size_t)o->n_cards);
    for (int c = 0; c < o->n_cards; ++c) card_scan(p, o, y, c, Q[c]);
    for (int h = 0; h < R; ++h) {
        int i = p->pos[h];
        if (i < 0) { eq[h] = 0.f; continue; }
        int cs[2] = {o->hole[2 * h], o->hole[2 * h + 1]};
        if (mode == 0) {
            float m = Q[cs[0]][p->n_t] + Q[cs[1]][p->n_t];
            eq[h] = T - (m - x[h]);
        } else {
            int gs = p->gs[i], ge = p->ge[i];
            float G = P[gs] - (T - P[ge]);
            float K[2];
            for (int k = 0; k < 2; ++k) {
                const int16_t* row = p->cl + (size_t)cs[k] * (o->n_cards - 1);
                int lo = 0, hi = 0;
                while (lo < p->n_t && row[lo] < gs) lo++;
                hi = lo;
                while (hi < p->n_t && row[hi] < ge) hi++;
                K[k] = Q[cs[k]][lo] - (Q[cs[k]][p->n_t] - Q[cs[k]][hi]);
            }
            eq[h] = G - (K[0] + K[1]); /* the two corrections are added first: a + b is commutative bit for bit, so the GPU may accumulate them in any order */
        }
    }
    free(Q); free(P); free(y);
}

/* O(R^2) definition of the 2-card showdown equity in float64 (cross-check only; SURVEY Appendix C last-but-one row) */
void orc_showdown_bruteforce(Orc* o, const float* x, int board_id, double* eq) {
    const int8_t* board = o->boards + (size_t)board_id * o->board_len;
    int32_t* rk = (int32_t*)malloc(sizeof(int32_t) * o->R);
    ranks_on_board(o, board, rk);
    for (int h = 0; h < o->R; ++h) {
        double e = 0.;
        if (rk[h] >= 0)
            for (int ho = 0; ho < o->R; ++ho) {
                if (rk[ho] < 0) continue;
                if (o->hole[2 * ho] == o->hole[2 * h] || o->hole[2 * ho] == o->hole[2 * h + 1] ||
                    o->hole[2 * ho + 1] == o->hole[2 * h] || o->hole[2 * ho + 1] == o->hole[2 * h + 1]) continue;
                if (rk[h] > rk[ho]) e += x[ho];
                else if (rk[h] < rk[ho]) e -= x[ho];
            }
        eq[h] = e;
    }
    free(rk);
}

/* test hook: canonical-order terminal equity of one vector */
void orc_terminal_equity(Orc* o, const float* x, int board_id, int mode, float* eq) {
    if (o->n_hole == 2) terminal_equity_2card(o, x, board_id, mode, eq);
    else if (mode == 0) fold_equity_1card(o, x, eq);
    else showdown_equity_1card(o, x, o->boards[(size_t)board_id * o->board_len], eq);
}

/* ValueFiller.compute_cf_values_heads_up, terminal branch (ValueFiller.py:34-62) */
static void terminal_values(Orc* o, int node) {
    const int R = o->R;
    float* eq = (float*)malloc(sizeof(float) * 2 * (size_t)R);
    const int bid = o->board_id[node];
    const int fold = o->kind[node] == K_FOLD;
    for (int p = 0; p < 2; ++p) {
        const float* x = V2(o, reach, node, 1 - p);
        float* e = eq + (size_t)p * R;
        if (o->n_hole == 2) {
            if (!fold && bid < 0) abort(); /* refused at orc_create (orc_unsupported) */
            terminal_equity_2card(o, x, bid, fold ? 0 : 1, e);
            for (int h = 0; h < R; ++h) e[h] = e[h] * o->eq_const;
        } else if (fold) {
            fold_equity_1card(o, x, e);
            for (int h = 0; h < R; ++h) e[h] = e[h] * o->eq_const;
        } else if (bid >= 0) {
            showdown_equity_1card(o, x, o->boards[(size_t)bid * o->board_len], e);
            for (int h = 0; h < R; ++h) e[h] = e[h] * o->eq_const;
        } else {
            /* _get_call_eq_preflop (ValueFiller.py:160-175): mean over the N-2 possible board cards */
            float* acc = (float*)calloc(R, sizeof(float));
            float* xr = (float*)malloc(sizeof(float) * R);
            float* one = (float*)malloc(sizeof(float) * R);
            for (int c = 0; c < o->n_cards; ++c) {
                memcpy(xr, x, sizeof(float) * R);
                xr[c] = 0.f;
                showdown_equity_1card(o, xr, c, one);
                for (int h = 0; h < R; ++h) acc[h] = acc[h] + one[h] * o->eq_const;
            }
            float div = (float)(o->n_cards - 2);
            for (int h = 0; h < R; ++h) e[h] = acc[h] / div;
            free(acc); free(xr); free(one);
        }
    }
    if (fold) {
        float* e = eq + (size_t)o->acted_last[node] * R;
        for (int h = 0; h < R; ++h) e[h] = -e[h];
    }
    /* hands blocked by the board are worth 0 (ValueFiller.py:57-59; generalised: any hole card on the board) */
    if (bid >= 0) {
        const int8_t* board = o->boards + (size_t)bid * o->board_len;
        for (int h = 0; h < R; ++h)
            if (hand_blocked(o, h, board, o->board_len)) { eq[h] = 0.f; eq[R + h] = 0.f; }
    }
    const float pot = (float)o->main_pot[node];
    for (int p = 0; p < 2; ++p)
        for (int h = 0; h < R; ++h) {
            float v = (eq[(size_t)p * R + h] * pot) / 2.f;
            V2(o, ev, node, p)[h] = v;
            V2(o, ev_br, node, p)[h] = v;
        }
    free(eq);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* tree passes                                                                                                         */
/* ------------------------------------------------------------------------------------------------------------------ */
/* StrategyFiller._update_reach_probs (StrategyFiller.py:118-146) */
static void update_reach(Orc* o, int node) {
    const int R = o->R;
    if (o->kind[node] == K_DECISION) {
        const int a = o->actor[node];
        for (int i = 0; i < o->n_children[node]; ++i) {
            int c = child_of(o, node, i);
            memcpy(V2(o, reach, c, 0), V2(o, reach, node, 0), sizeof(float) * 2 * R);
            const double* s = COL(o, strategy, o->first_col[node] + i);
            const float* rp = V2(o, reach, node, a);
            float* rc = V2(o, reach, c, a);
            if (o->strat_f64[node]) for (int h = 0; h < R; ++h) rc[h] = (float)(s[h] * (double)rp[h]);
            else for (int h = 0; h < R; ++h) rc[h] = (float)s[h] * rp[h];
            update_reach(o, c);
        }
    } else if (o->kind[node] == K_CHANCE) {
        /* board subtrees are independent: threads only change who computes a subtree, never the arithmetic inside it */
#pragma omp parallel for schedule(dynamic, 4) if (o->n_children[node] >= 64)
        for (int i = 0; i < o->n_children[node]; ++i) {
            int c = child_of(o, node, i);
            const int8_t* board = o->boards + (size_t)o->board_id[c] * o->board_len;
            for (int p = 0; p < 2; ++p)
                for (int h = 0; h < R; ++h) {
                    float w = hand_blocked(o, h, board, o->board_len) ? 0.f : (o->board_w ? o->board_w[i] : o->chance_w[node]); /* StrategyFiller.py:159-166 */
                    V2(o, reach, c, p)[h] = V2(o, reach, node, p)[h] * w;
                }
            update_reach(o, c);
        }
    }
}

/* ValueFiller.compute_cf_values_heads_up (ValueFiller.py:21-101) */
/* suit isomorphism: a hand's value at the chance node = the mean over its suit orbit (the hands of its class, ascending hand index, running adds; one
 * correctly rounded division) of the multiplicity-weighted sum of the board values */
static void symmetrize_node(Orc* o, int node) {
    const int R = o->R;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)R);
    for (int p = 0; p < 2; ++p)
        for (int which = 0; which < 2; ++which) {
            float* v = (which ? o->ev_br : o->ev) + ((size_t)node * 2 + p) * R;
            for (int h = 0; h < R; ++h) {
                float sum = 0.f;
                int n = 0;
                for (int g = 0; g < R; ++g)
                    if (o->sym_class[g] == o->sym_class[h]) { sum = n == 0 ? v[g] : sum + v[g]; ++n; }
                tmp[h] = sum / (float)n;
            }
            memcpy(v, tmp, sizeof(float) * (size_t)R);
        }
    free(tmp);
}

static void compute_ev(Orc* o, int node) {
    const int R = o->R;
    const int A = o->n_children[node];
    if (o->ov_ev && node == o->ov_node) {
        memcpy(V2(o, ev, node, 0), o->ov_ev, sizeof(float) * 2 * (size_t)R);
        memcpy(V2(o, ev_br, node, 0), o->ov_ev_br, sizeof(float) * 2 * (size_t)R);
        if (o->kind[node] == K_CHANCE && o->sym_class) symmetrize_node(o, node);  /* the override is the weighted SUM of a chunked run (make_fhp_golden_chunked.py) */
        return;
    }
    if (o->kind[node] >= K_FOLD) { terminal_values(o, node); return; }
    if (o->kind[node] == K_CHANCE && A >= 64) {
        if (o->n_hole == 2) ensure_plans(o);
#pragma omp parallel for schedule(dynamic, 4)
        for (int i = 0; i < A; ++i) compute_ev(o, child_of(o, node, i));
    } else {
        for (int i = 0; i < A; ++i) compute_ev(o, child_of(o, node, i));
    }
    if (o->kind[node] == K_CHANCE) {
        /* ValueFiller.py:76-78 sums the chance children in child order (NumPy outer-axis reduce = running add). The
         * canonical order here is the same running add, nested: blocks of 32 children, groups of 32 blocks, then the
         * groups -- identical to the reference for <= 32 boards (every Leduc game) and independent of how many GPUs
         * share the boards of a big tree (each GPU owns whole groups; DESIGN.md "multi-GPU"). */
        for (int p = 0; p < 2; ++p)
#pragma omp parallel for schedule(static) if (A >= 64)
            for (int h = 0; h < R; ++h)
                for (int which = 0; which < 2; ++which) {
                    float* arr = which ? o->ev_br : o->ev;
                    float total = 0.f;
                    for (int g0 = 0, gi = 0; g0 < A; g0 += 32 * 32, ++gi) {
                        float gsum = 0.f;
                        for (int b0 = g0, bi = 0; b0 < A && b0 < g0 + 32 * 32; b0 += 32, ++bi) {
                            float bsum = arr[((size_t)child_of(o, node, b0) * 2 + p) * R + h];
                            for (int i = b0 + 1; i < A && i < b0 + 32; ++i)
                                bsum = bsum + arr[((size_t)child_of(o, node, i) * 2 + p) * R + h];
                            gsum = bi == 0 ? bsum : gsum + bsum;
                        }
                        total = gi == 0 ? gsum : total + gsum;
                    }
                    arr[((size_t)node * 2 + p) * R + h] = total;
                }
        if (o->sym_class) symmetrize_node(o, node);
        return;
    }
    const int pl = o->actor[node], op = 1 - pl;
    for (int h = 0; h < R; ++h) {
        /* actor: strategy-weighted sum (ValueFiller.py:87) */
        if (o->strat_f64[node]) {
            double acc = 0.;
            for (int i = 0; i < A; ++i) {
                double prod = COL(o, strategy, o->first_col[node] + i)[h] * (double)V2(o, ev, child_of(o, node, i), pl)[h];
                acc = i == 0 ? prod : acc + prod;
            }
            V2(o, ev, node, pl)[h] = (float)acc;
        } else {
            float acc = 0.f;
            for (int i = 0; i < A; ++i) {
                float prod = (float)COL(o, strategy, o->first_col[node] + i)[h] * V2(o, ev, child_of(o, node, i), pl)[h];
                acc = i == 0 ? prod : acc + prod;
            }
            V2(o, ev, node, pl)[h] = acc;
        }
        /* opponent: plain sums (ValueFiller.py:88,90); actor BR: max / first argmax (ValueFiller.py:91-93) */
        float s = V2(o, ev, child_of(o, node, 0), op)[h], sb = V2(o, ev_br, child_of(o, node, 0), op)[h];
        float mx = V2(o, ev_br, child_of(o, node, 0), pl)[h];
        int arg = 0;
        for (int i = 1; i < A; ++i) {
            s = s + V2(o, ev, child_of(o, node, i), op)[h];
            sb = sb + V2(o, ev_br, child_of(o, node, i), op)[h];
            float v = V2(o, ev_br, child_of(o, node, i), pl)[h];
            if (v > mx) { mx = v; arg = i; }
        }
        V2(o, ev, node, op)[h] = s;
        V2(o, ev_br, node, op)[h] = sb;
        V2(o, ev_br, node, pl)[h] = mx;
        o->br_idx[(size_t)node * R + h] = arg;
    }
}

/* root exploitability (ValueFiller.py:96-101) */
static void root_exploitability(Orc* o) {
    const int R = o->R;
    float* eps = (float*)malloc(sizeof(float) * (size_t)R);
    for (int p = 0; p < 2; ++p) {
        for (int h = 0; h < R; ++h) {
            float w = V2(o, ev, 0, p)[h] * V2(o, reach, 0, p)[h];
            float wb = V2(o, ev_br, 0, p)[h] * V2(o, reach, 0, p)[h];
            eps[h] = wb - w;
        }
        if (o->n_hole == 1) o->expl[p] = np_sum_f32(eps, R, 1);
        else {
            float* P = (float*)malloc(sizeof(float) * (size_t)(R + 2));
            prefix_chunked(eps, R, P);
            o->expl[p] = P[R];
            free(P);
        }
    }
    free(eps);
}

void orc_update_reach(Orc* o) {
    const float r0 = (float)(1.0 / (double)o->R); /* PublicTree.py:122-124 */
    for (int i = 0; i < 2 * o->R; ++i) o->reach[i] = r0;
    update_reach(o, 0);
}

void orc_compute_ev(Orc* o) {
    compute_ev(o, 0);
    root_exploitability(o);
}

/* StrategyFiller.fill_uniform_random (StrategyFiller.py:17-24,48-65): float64 uniform, then reach */
void orc_fill_uniform(Orc* o) {
    for (int n = 0; n < o->n_nodes; ++n)
        if (o->kind[n] == K_DECISION) {
            const int A = o->n_children[n];
            o->strat_f64[n] = 1;
            for (int i = 0; i < A; ++i) {
                double* s = COL(o, strategy, o->first_col[n] + i);
                for (int h = 0; h < o->R; ++h) s[h] = 1.0 / (double)A;
            }
        }
    orc_update_reach(o);
}

/* arbitrary strategy (fill_with_agent_policy / random fill semantics): column-major [n_cols][R] */
void orc_set_strategy(Orc* o, const double* strat, int is_f64) {
    memcpy(o->strategy, strat, sizeof(double) * (size_t)o->n_cols * o->R);
    for (int n = 0; n < o->n_nodes; ++n) o->strat_f64[n] = (uint8_t)(is_f64 != 0);
    orc_update_reach(o);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* CFR                                                                                                                 */
/* ------------------------------------------------------------------------------------------------------------------ */
void orc_cfr_reset(Orc* o, int variant, int delay) { /* _CFRBase.reset (_CFRBase.py:110-120) */
    o->variant = variant;
    o->delay = delay;
    o->iter = 0;
    memset(o->regret, 0, sizeof(float) * (size_t)o->n_cols * o->R);
    memset(o->avg_sum, 0, sizeof(float) * (size_t)o->n_cols * o->R);
    memset(o->avg, 0, sizeof(double) * (size_t)o->n_cols * o->R);
    memset(o->avg_f64, 0, o->n_nodes);
    orc_fill_uniform(o);
    orc_compute_ev(o);
}

/* orc_cfr_reset without the evaluation (chunked runs set a chunk up and evaluate it themselves) */
void orc_cfr_configure(Orc* o, int variant, int delay) {
    o->variant = variant;
    o->delay = delay;
    o->iter = 0;
    memset(o->regret, 0, sizeof(float) * (size_t)o->n_cols * o->R);
    memset(o->avg_sum, 0, sizeof(float) * (size_t)o->n_cols * o->R);
    memset(o->avg, 0, sizeof(double) * (size_t)o->n_cols * o->R);
    memset(o->avg_f64, 0, o->n_nodes);
    orc_fill_uniform(o);
}

static void compute_regrets(Orc* o, int p) { /* _CFRBase._compute_regrets (_CFRBase.py:146-185) */
    const int R = o->R;
#pragma omp parallel for schedule(static, 64) if (o->n_nodes >= 1024)
    for (int n = 0; n < o->n_nodes; ++n) {
        if (o->kind[n] != K_DECISION || o->actor[n] != p) continue;
        const float* strat_ev = V2(o, ev, n, p);
        for (int i = 0; i < o->n_children[n]; ++i) {
            const float* ev_a = V2(o, ev, child_of(o, n, i), p);
            float* reg = COL(o, regret, o->first_col[n] + i);
            for (int h = 0; h < R; ++h) {
                float d = ev_a[h] - strat_ev[h];
                float r;
                if (o->iter == 0) r = d;                                          /* *_first_it */
                else if (o->variant == V_LINEAR) r = ((float)(o->iter + 1) * d) + reg[h]; /* LinearCFR.py:27-28 */
                else r = d + reg[h];                                              /* VanillaCFR.py:26-27, CFRPlus.py:37-38 */
                if (o->variant == V_PLUS) r = r > 0.f ? r : 0.f;                  /* np.maximum(..., 0) */
                reg[h] = r;
            }
        }
    }
}

static void compute_new_strategy(Orc* o, int p) { /* VanillaCFR.py:32-52, CFRPlus.py:43-63, LinearCFR.py:33-51 */
    const int R = o->R;
#pragma omp parallel for schedule(static, 64) if (o->n_nodes >= 1024)
    for (int n = 0; n < o->n_nodes; ++n) {
        float tmp[PRL_ORC_MAX_ACTIONS];
        if (o->kind[n] != K_DECISION || o->actor[n] != p) continue;
        const int A = o->n_children[n];
        const float unif = (float)(1.0 / (double)A);
        for (int h = 0; h < R; ++h) {
            for (int i = 0; i < A; ++i) {
                float r = COL(o, regret, o->first_col[n] + i)[h];
                tmp[i] = (o->variant == V_PLUS) ? r : (r > 0.f ? r : 0.f);
            }
            float s = np_sum_f32(tmp, A, 1);
            for (int i = 0; i < A; ++i) COL(o, strategy, o->first_col[n] + i)[h] = s > 0.f ? (double)(tmp[i] / s) : (double)unif;
        }
        o->strat_f64[n] = 0;
    }
}

static void add_strategy_to_average(Orc* o, int p) { /* VanillaCFR.py:54-77, CFRPlus.py:65-87, LinearCFR.py:53-76 */
    const int R = o->R;
#pragma omp parallel for schedule(static, 64) if (o->n_nodes >= 1024)
    for (int n = 0; n < o->n_nodes; ++n) {
        float tmp[PRL_ORC_MAX_ACTIONS];
        if (o->kind[n] != K_DECISION || o->actor[n] != p) continue;
        const int A = o->n_children[n];
        if (o->variant == V_PLUS) {
            if (o->iter > o->delay) {
                long long cw = 0;
                for (int k = o->delay + 1; k <= o->iter; ++k) cw += k;
                long long nw = o->iter - o->delay + 1;
                double m_old = (double)cw / (double)(cw + nw), m_new = (double)nw / (double)(cw + nw);
                for (int i = 0; i < A; ++i) {
                    double* av = COL(o, avg, o->first_col[n] + i);
                    const double* st = COL(o, strategy, o->first_col[n] + i);
                    for (int h = 0; h < R; ++h) av[h] = m_old * av[h] + m_new * st[h];
                }
                o->avg_f64[n] = 1;
            } else if (o->iter == o->delay) {
                for (int i = 0; i < A; ++i)
                    memcpy(COL(o, avg, o->first_col[n] + i), COL(o, strategy, o->first_col[n] + i), sizeof(double) * R);
                o->avg_f64[n] = o->strat_f64[n];
            }
            continue;
        }
        const float* rp = V2(o, reach, n, p);
        for (int h = 0; h < R; ++h) {
            for (int i = 0; i < A; ++i) {
                float contrib = (float)COL(o, strategy, o->first_col[n] + i)[h] * rp[h];
                if (o->variant == V_LINEAR) contrib = contrib * (float)(o->iter + 1);
                float* as = COL(o, avg_sum, o->first_col[n] + i);
                as[h] = o->iter > 0 ? as[h] + contrib : contrib;
                tmp[i] = as[h];
            }
            float s = np_sum_f32(tmp, A, 1);
            for (int i = 0; i < A; ++i)
                COL(o, avg, o->first_col[n] + i)[h] = (s == 0.f) ? 1.0 / (double)A : (double)(tmp[i] / s);
        }
        o->avg_f64[n] = 1;
    }
}

void orc_cfr_iteration(Orc* o) { /* _CFRBase.iteration (_CFRBase.py:122-134), without _evaluate_avg_strats */
    for (int p = 0; p < 2; ++p) {
        orc_compute_ev(o);
        compute_regrets(o, p);
        compute_new_strategy(o, p);
        orc_update_reach(o);
        add_strategy_to_average(o, p);
    }
    o->iter += 1;
    orc_compute_ev(o);
}

/* _CFRBase._evaluate_avg_strats (_CFRBase.py:218-262) on scratch buffers: the training tree is left untouched */
void orc_eval_avg(Orc* o, float out_expl[2]) {
    size_t nv = (size_t)o->n_nodes * 2 * o->R, nc = (size_t)o->n_cols * o->R;
    double* s_strategy = o->strategy;
    uint8_t* s_flags = o->strat_f64;
    float *s_reach = o->reach, *s_ev = o->ev, *s_ev_br = o->ev_br;
    float s_expl[2] = {o->expl[0], o->expl[1]};
    o->strategy = (double*)malloc(sizeof(double) * nc);
    memcpy(o->strategy, o->avg, sizeof(double) * nc);
    o->strat_f64 = (uint8_t*)malloc(o->n_nodes);
    memcpy(o->strat_f64, o->avg_f64, o->n_nodes);
    o->reach = (float*)calloc(nv, sizeof(float));
    o->ev = (float*)calloc(nv, sizeof(float));
    o->ev_br = (float*)calloc(nv, sizeof(float));
    orc_update_reach(o);
    orc_compute_ev(o);
    out_expl[0] = o->expl[0];
    out_expl[1] = o->expl[1];
    free(o->strategy); free(o->strat_f64); free(o->reach); free(o->ev); free(o->ev_br);
    o->strategy = s_strategy; o->strat_f64 = s_flags; o->reach = s_reach; o->ev = s_ev; o->ev_br = s_ev_br;
    o->expl[0] = s_expl[0]; o->expl[1] = s_expl[1];
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* accessors                                                                                                           */
/* ------------------------------------------------------------------------------------------------------------------ */
/* pieces of orc_cfr_iteration, for runs that evaluate a tree chunk by chunk (tests/golden/make_fhp_golden_chunked.py) */
void orc_compute_regrets(Orc* o, int p) { compute_regrets(o, p); }
void orc_compute_new_strategy(Orc* o, int p) { compute_new_strategy(o, p); }
void orc_add_strategy_to_average(Orc* o, int p) { add_strategy_to_average(o, p); }
void orc_set_iter(Orc* o, int it) { o->iter = it; }
void orc_set_override(Orc* o, int node, const float* ev2R, const float* ev_br2R) { o->ov_node = node; o->ov_ev = ev2R; o->ov_ev_br = ev_br2R; }
float* orc_reach(Orc* o) { return o->reach; }
float* orc_ev(Orc* o) { return o->ev; }
float* orc_ev_br(Orc* o) { return o->ev_br; }
float* orc_regret(Orc* o) { return o->regret; }
float* orc_avg_sum(Orc* o) { return o->avg_sum; }
double* orc_strategy(Orc* o) { return o->strategy; }
double* orc_avg(Orc* o) { return o->avg; }
uint8_t* orc_strat_f64(Orc* o) { return o->strat_f64; }
uint8_t* orc_avg_f64(Orc* o) { return o->avg_f64; }
int32_t* orc_br_idx(Orc* o) { return o->br_idx; }
float* orc_expl(Orc* o) { return o->expl; }
int orc_iter(Orc* o) { return o->iter; }
void orc_set_chance_weights(Orc* o, const float* w) { memcpy(o->chance_w, w, sizeof(float) * (size_t)o->n_nodes); }
/* weighted boards: w[i] for the i-th child of the (one) chance node; NULL clears */
void orc_set_board_weights(Orc* o, const float* w, int n) {
    free(o->board_w);
    o->board_w = NULL;
    if (w) { o->board_w = (float*)malloc(sizeof(float) * (size_t)n); memcpy(o->board_w, w, sizeof(float) * (size_t)n); }
}
/* suit symmetrisation of the chance node's values: class_of[h] for every hand; NULL clears */
void orc_set_symmetrize(Orc* o, const int32_t* class_of) {
    free(o->sym_class);
    o->sym_class = NULL;
    if (class_of) { o->sym_class = (int32_t*)malloc(sizeof(int32_t) * (size_t)o->R); memcpy(o->sym_class, class_of, sizeof(int32_t) * (size_t)o->R); }
}
int orc_unsupported(Orc* o) { return o->unsupported; }
/* worker threads for the per-board / per-node loops (results do not depend on it); bench.py's cpu_baseline uses 1 */
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_max_threads(void) { return omp_get_max_threads(); }
float orc_np_sum_f32(const float* a, int n) { return np_sum_f32(a, n, 1); }
double orc_np_sum_f64(const double* a, int n) { return np_sum_f64(a, n, 1); }
void orc_prefix_chunked(const float* y, int n, float* P) { prefix_chunked(y, n, P); }
