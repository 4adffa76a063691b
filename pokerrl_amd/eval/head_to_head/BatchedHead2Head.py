"""
BatchedHead2Head: head-to-head evaluation with every hand played on the GPU, one lane per hand (SURVEY.md section 8f-3 on the
batched env). Same episodes as LocalHead2HeadMaster.play -- identical float32 winnings for the same decks and agent draws --
but both players have to be agents the kernel can query: the library's synthetic agents or tabular policies in HBM (kind "table":
pokerrl_amd.rl.tabular_agent.PolicyTable) -- a host EvalAgent cannot be queried from inside a kernel.

    h2h = BatchedHead2Head(t_prof, kinds=("hash", "hash"), seeds=(11, 12))
    h2h = BatchedHead2Head(t_prof, kinds=("table", "hash"), seeds=(11, 12), tables=(PolicyTable.from_cfr(cfr), None))
    w = h2h.play(n_hands=1 << 20, deck_seed=0)        # float32 [2 * n_hands]: the reference agent in seat 0, then in seat 1
"""
import ctypes

import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.eval._.EvaluatorMasterBase import EvaluatorMasterBase
from pokerrl_amd.eval.head_to_head.LocalHead2HeadMaster import LocalHead2HeadMaster
from pokerrl_amd.eval.lbr.BatchedLBR import AGENT_KINDS, deal_decks
from pokerrl_amd.rl import rl_util


class BatchedHead2Head:
    def __init__(self, t_prof, kinds=("hash", "hash"), seeds=(11, 12), tables=(None, None)):
        assert t_prof.n_seats == 2
        assert all((k == "table") == (t is not None) for k, t in zip(kinds, tables)), "kind 'table' plays a PolicyTable"
        self.tables = tuple(tables)
        self.t_prof = t_prof
        self._bldr = rl_util.get_env_builder(t_prof=t_prof)
        env_cls = self._bldr.env_cls
        self._env_cls = env_cls
        self._game = env_cls.native_game(self._bldr.env_args)
        self._rules = env_cls.native_rules()
        self.kinds = [AGENT_KINDS[k] for k in kinds]
        self.seeds = [int(s) for s in seeds]
        self.n_deal = 2 * self._rules.n_hole_cards + self._rules.n_board_cards
        self.last_stats = None

    def set_stack_size(self, stack_size):
        self._game.start_stack[0], self._game.start_stack[1] = int(stack_size[0]), int(stack_size[1])

    def run(self, ref_seat, n_hands, decks=None, deck_seed=0, episode_base=0, first_hand=0):
        """n_hands with the reference agent (kinds[0], seeds[0]) in `ref_seat`; decks int8 [n_hands, 2*N_HOLE + N_BOARD] (seat 0,
        seat 1, board in deal order) or None for counter-based decks. Returns float32 [n_hands]."""
        L = _native.lib()
        _native.require_device()
        if decks is None:
            decks = deal_decks(n_hands, self._rules.n_cards, self.n_deal, deck_seed, first_hand)
        decks = np.ascontiguousarray(decks, dtype=np.int8)
        assert decks.shape == (n_hands, self.n_deal)
        args = self._bldr.env_args
        stacks = [self._game.start_stack[0], self._game.start_stack[1]]
        reward_scalar = (float(sum(stacks)) / 2.0 / 5.0) if args.scale_rewards else 1.0
        out = np.zeros(n_hands, np.float32)
        stats = np.zeros(2, np.uint64)
        ms = ctypes.c_float()
        tail = (int(episode_base), reward_scalar, float(self._env_cls.EV_NORMALIZER), decks.ctypes.data_as(ctypes.c_void_p),
                out.ctypes.data_as(ctypes.c_void_p), stats.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ms))
        if any(t is not None for t in self.tables):
            dev = [None if t is None else t.device() for t in self.tables]
            _native.check(L.prl_h2h_batch_run_tables(ctypes.byref(self._game), ctypes.byref(self._rules), int(n_hands), int(ref_seat), self.kinds[0],
                                                     self.seeds[0], dev[0], self.kinds[1], self.seeds[1], dev[1], *tail), L)
        else:
            _native.check(L.prl_h2h_batch_run(ctypes.byref(self._game), ctypes.byref(self._rules), int(n_hands), int(ref_seat), self.kinds[0],
                                              self.seeds[0], self.kinds[1], self.seeds[1], *tail), L)
        self.last_stats = {"env_steps": int(stats[0]), "showdowns": int(stats[1]), "device_ms": float(ms.value)}
        return out

    def play(self, n_hands, decks=None, deck_seed=0, first_hand=0):
        """LocalHead2HeadMaster.play's layout: [2 * n_hands], seat 0 block then seat 1 block; episodes are numbered through both
        blocks like the host evaluator's agents count them. decks: [2 * n_hands, n_deal] or None (counter-based decks from hand `first_hand` on)."""
        halves = []
        for ref_seat in range(2):
            d = None if decks is None else decks[ref_seat * n_hands:(ref_seat + 1) * n_hands]
            at = first_hand + ref_seat * n_hands
            halves.append(self.run(ref_seat, n_hands, decks=d, deck_seed=deck_seed, episode_base=at, first_hand=at))
        return np.concatenate(halves)


class _BatchedSide:
    """one player of BatchedHead2HeadMaster: what LocalHead2HeadMaster asks of an EvalAgent, for an agent that lives in the kernel"""

    def __init__(self, kind, seed, table):
        self.kind, self.seed, self.table, self._mode = kind, int(seed), table, None

    def set_mode(self, mode):
        self._mode = mode

    def get_mode(self):
        return self._mode

    def set_stack_size(self, stack_size):
        pass  # the master hands the stack size to play()

    def can_compute_mode(self):
        return self.kind != "table" or self.table is not None

    def update_weights(self, w):
        """w: None (keep), a PolicyTable (this side plays it), or a dict mode -> PolicyTable (each side picks its mode's, like the reference's agents)"""
        if isinstance(w, dict):
            w = w.get(self._mode)
        if w is not None:
            self.kind, self.table = "table", w


class BatchedHead2HeadMaster(LocalHead2HeadMaster):
    """LocalHead2HeadMaster (PokerRL/eval/head_to_head/LocalHead2HeadMaster.py:10-135: set_modes / update_weights / evaluate(iter_nr), mean +- 95 %
    confidence under the reference's experiment names) with the hands played by BatchedHead2Head: tabular policies in HBM (what update_weights pulls
    from the chief: a PolicyTable per mode) or the synthetic agents. Counter-based decks; every evaluation continues the hand numbering."""

    def __init__(self, t_prof, chief_handle, kinds=("table", "table"), seeds=(11, 12), tables=(None, None), deck_seed=0):
        bldr = rl_util.get_env_builder(t_prof=t_prof)
        assert bldr.N_SEATS == 2, "Only HU supported!"
        EvaluatorMasterBase.__init__(self, t_prof=t_prof, eval_env_bldr=bldr, chief_handle=chief_handle, eval_type="Head2Head_Winnings", log_conf_interval=True)
        self._args = t_prof.module_args["h2h"]
        self._env_bldr = bldr
        self._eval_agents = [_BatchedSide(k, s, t) for k, s, t in zip(kinds, seeds, tables)]
        self._REFERENCE_AGENT = 0
        self.deck_seed, self._next_hand = int(deck_seed), 0

    def play(self, stack_size):
        a = self._eval_agents
        b = BatchedHead2Head(self._t_prof, kinds=(a[0].kind, a[1].kind), seeds=(a[0].seed, a[1].seed), tables=(a[0].table, a[1].table))
        b.set_stack_size(stack_size)
        w = b.play(self._args.n_hands, deck_seed=self.deck_seed, first_hand=self._next_hand)
        self._next_hand += 2 * self._args.n_hands
        return w
