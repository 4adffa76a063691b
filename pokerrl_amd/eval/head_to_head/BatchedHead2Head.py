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

    def play(self, n_hands, decks=None, deck_seed=0):
        """LocalHead2HeadMaster.play's layout: [2 * n_hands], seat 0 block then seat 1 block; episodes are numbered through both
        blocks like the host evaluator's agents count them. decks: [2 * n_hands, n_deal] or None."""
        halves = []
        for ref_seat in range(2):
            d = None if decks is None else decks[ref_seat * n_hands:(ref_seat + 1) * n_hands]
            halves.append(self.run(ref_seat, n_hands, decks=d, deck_seed=deck_seed, episode_base=ref_seat * n_hands, first_hand=ref_seat * n_hands))
        return np.concatenate(halves)
