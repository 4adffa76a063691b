"""
BatchedLBR: LBR with every hand played start to finish on the GPU (BASELINE.json config 5: "2^20 batched PokerEnv rollouts +
7-card eval"). Same computation as LocalLBRWorker.run -- the per-hand winnings are bit-identical for the same decks and the
same agent draws -- but the agent has to be one the kernel can query: one of the library's synthetic agents ("uniform", "hash") or a TABULAR policy
resident in HBM ("table": pokerrl_amd.rl.tabular_agent.PolicyTable, e.g. a CFR solver's average strategy), because a host EvalAgent cannot be
queried from inside a kernel (batched querying of neural agents is the "next" row of SURVEY.md section 8f).

    lbr = BatchedLBR(t_prof, agent_kind="hash", agent_seed=7)
    winnings = lbr.run(agent_seat_id=0, n_hands=1 << 20, deck_seed=0)          # float32 [n_hands], mbb per hand
    lbr = BatchedLBR(t_prof, agent_kind="table", table=PolicyTable.from_cfr(cfr))   # LBR against the solver's own output
"""
import ctypes

import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.eval.lbr import _util

AGENT_KINDS = {"uniform": 0, "hash": 1, "table": 2}


def deal_decks(n_hands, n_cards_in_deck, n_deal, seed, first_hand=0):
    """Counter-based decks, dealt on the GPU (prl_deal_decks); deal_decks_host is the same algorithm in NumPy (tests compare them)."""
    out = np.zeros((n_hands, n_deal), np.int8)
    _native.check(_native.lib().prl_deal_decks(int(n_hands), int(n_cards_in_deck), int(n_deal), int(seed) & (2 ** 64 - 1), int(first_hand),
                                               out.ctypes.data_as(ctypes.c_void_p)))
    return out


def deal_decks_host(n_hands, n_cards_in_deck, n_deal, seed, first_hand=0):
    """Counter-based decks: hand i's cards depend on (seed, i) only, so any split of the hands over GPUs deals the same cards.
    A partial Fisher-Yates shuffle of 0..n_cards-1 driven by a SplitMix-style hash; returns int8 [n_hands, n_deal]."""
    with np.errstate(over="ignore"):
        idx = (np.arange(first_hand, first_hand + n_hands, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
    deck = np.tile(np.arange(n_cards_in_deck, dtype=np.int8), (n_hands, 1))
    rows = np.arange(n_hands)
    with np.errstate(over="ignore"):  # 64-bit wrap-around is the point of the hash
        for i in range(n_deal):
            x = idx + np.uint64(i + 1) * np.uint64(0xBF58476D1CE4E5B9)
            x ^= x >> np.uint64(30)
            x *= np.uint64(0xBF58476D1CE4E5B9)
            x ^= x >> np.uint64(27)
            x *= np.uint64(0x94D049BB133111EB)
            x ^= x >> np.uint64(31)
            j = i + (x % np.uint64(n_cards_in_deck - i)).astype(np.int64)
            tmp = deck[rows, i].copy()
            deck[rows, i] = deck[rows, j]
            deck[rows, j] = tmp
    return np.ascontiguousarray(deck[:, :n_deal])


class BatchedLBR:
    def __init__(self, t_prof, agent_kind="hash", agent_seed=7, table=None):
        assert t_prof.n_seats == 2
        assert (agent_kind == "table") == (table is not None), "agent_kind 'table' plays a PolicyTable"
        self.table = table
        self.t_prof = t_prof
        self.lbr_args = t_prof.module_args["lbr"]
        self._lbr_bldr = _util.get_env_builder_lbr(t_prof=t_prof)
        env_cls = self._lbr_bldr.env_cls
        self._env_cls = env_cls
        self._g_lbr = env_cls.native_game(self._lbr_bldr.env_args)
        self._g_agent = env_cls.native_game(t_prof.module_args["env"])
        self._rules = env_cls.native_rules()
        self.agent_kind = AGENT_KINDS[agent_kind]
        self.agent_seed = int(agent_seed)
        self.n_deal = 2 * self._rules.n_hole_cards + self._rules.n_board_cards
        self.last_stats = None

    def set_stack_size(self, stack_size):
        for g in (self._g_lbr, self._g_agent):
            g.start_stack[0], g.start_stack[1] = int(stack_size[0]), int(stack_size[1])

    def run(self, agent_seat_id, n_hands, decks=None, deck_seed=0, episode_base=0, first_hand=0):
        """decks: int8 [n_hands, 2 * N_HOLE_CARDS + N_BOARD_CARDS] 1d cards (seat 0, seat 1, board in deal order), or None for
        counter-based decks from (deck_seed, first_hand + i). Returns float32 [n_hands]; self.last_stats has the counters."""
        L = _native.lib()
        _native.require_device()
        if decks is None:
            decks = deal_decks(n_hands, self._rules.n_cards, self.n_deal, deck_seed, first_hand)
        decks = np.ascontiguousarray(decks, dtype=np.int8)
        assert decks.shape == (n_hands, self.n_deal)
        args = self._lbr_bldr.env_args
        stacks = [self._g_lbr.start_stack[0], self._g_lbr.start_stack[1]]
        reward_scalar = (float(sum(stacks)) / 2.0 / 5.0) if args.scale_rewards else 1.0
        ctr = self.lbr_args.lbr_check_to_round
        out = np.zeros(n_hands, np.float32)
        stats = np.zeros(4, np.uint64)
        ms = ctypes.c_float()
        fn, agent = (L.prl_lbr_batch_run, self.agent_kind) if self.table is None else (L.prl_lbr_batch_run_table, self.table.device())
        _native.check(fn(ctypes.byref(self._g_lbr), ctypes.byref(self._g_agent), ctypes.byref(self._rules), int(n_hands),
                         int(agent_seat_id), -1 if ctr is None else int(ctr), agent, self.agent_seed,
                         int(episode_base), reward_scalar, float(self._env_cls.EV_NORMALIZER),
                         decks.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                         stats.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ms)), L)
        self.last_stats = {"env_steps": int(stats[0]), "lbr_lookaheads": int(stats[1]), "range_board_equities": int(stats[2]),
                           "agent_actions": int(stats[3]), "device_ms": float(ms.value)}
        return out

    def run_sharded(self, agent_seat_id, n_hands_total, deck_seed=0, episode_base=0, group=None, device="cuda"):
        """LocalLBRMaster's hand split (LocalLBRMaster.py:53-69) over the ranks of a torch.distributed group, one process per GPU:
        rank r plays hands [r * n, (r + 1) * n) of the SAME counter-based deck / agent-draw streams (n = n_hands_total / world),
        so every hand is played exactly as the one-GPU run plays it; (sum, sum of squares, count) are all-reduced.
        Returns (mean, +-95 % half width, n_hands_total, this rank's float32 winnings) -- EvaluatorMasterBase.py:127-132's numbers.
        Without an initialised process group it is the one-rank run."""
        import torch
        import torch.distributed as dist
        world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_available() and dist.is_initialized() else (1, 0)
        assert n_hands_total % world == 0, "hands are split evenly over the ranks"
        n = n_hands_total // world
        x = self.run(agent_seat_id, n, deck_seed=deck_seed, episode_base=episode_base + rank * n, first_hand=rank * n)
        x64 = x.astype(np.float64)
        agg = torch.tensor([x64.sum(), (x64 * x64).sum(), float(n)], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(agg, group=group)
        tot, sq, cnt = (float(v) for v in agg.cpu())
        mean = tot / cnt
        sd = np.sqrt(max(sq / cnt - mean * mean, 0.0))
        return mean, 1.96 * sd / np.sqrt(cnt), int(cnt), x


class BatchedLBRWorker:
    """LocalLBRWorker's interface (`run(agent_seat_id, n_iterations, mode, stack_size)`, `update_weights`; PokerRL/eval/lbr/LocalLBRWorker.py:27-59) on
    the batched engine, so that LocalLBRMaster (LocalLBRMaster.py:36-69: per mode / stack size / seat it concatenates its workers' scores and logs mean and
    95 % confidence under the reference's experiment names) drives the GPU evaluation unchanged:

        master = LocalLBRMaster(t_prof, chief); master.set_worker_handles(BatchedLBRWorker(t_prof, table=PolicyTable.from_cfr(cfr)))
        master.update_weights(); master.evaluate(iter_nr)

    The agent is a tabular one (what `update_weights` receives from the chief's `pull_current_eval_strategy` -- a PolicyTable, or None to keep the
    present one) or one of the synthetic kinds. Decks are counter-based (deck_seed, hand number): every `run` continues the hand numbering, so
    successive evaluations -- and several workers given disjoint `first_hand` -- play fresh, reproducible hands."""

    def __init__(self, t_prof, chief_handle=None, agent_kind="table", agent_seed=7, table=None, deck_seed=0, first_hand=0):
        self.t_prof = t_prof
        self.chief_handle = chief_handle
        self._kind, self._seed = agent_kind, int(agent_seed)
        self._lbr = None if (agent_kind == "table" and table is None) else BatchedLBR(t_prof, agent_kind=agent_kind, agent_seed=agent_seed, table=table)
        self.deck_seed, self._next_hand = int(deck_seed), int(first_hand)

    def update_weights(self, weights_for_eval_agent):
        if weights_for_eval_agent is None:
            return
        assert self._kind == "table", "only the tabular agent has weights"
        self._lbr = BatchedLBR(self.t_prof, agent_kind="table", agent_seed=self._seed, table=weights_for_eval_agent)

    def run(self, agent_seat_id, n_iterations, mode, stack_size):
        """float32 [n_iterations] per-hand winnings of LBR; None when there is no table yet (the reference's worker returns None for a mode its agent cannot play)"""
        if self._lbr is None:
            return None
        self._lbr.set_stack_size(stack_size)
        w = self._lbr.run(agent_seat_id, int(n_iterations), deck_seed=self.deck_seed, first_hand=self._next_hand, episode_base=self._next_hand)
        self._next_hand += int(n_iterations)
        return w
