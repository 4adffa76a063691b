"""
LocalLBRWorker: Local Best Response against an EvalAgent (API and semantics of PokerRL/eval/lbr/LocalLBRWorker.py:12-376).

The episode logic (env stepping, agent queries, range updates) is host code over the native betting engine; what the
reference spends its time on -- enumerating the boards still to come, ranking all RANGE_SIZE hands on each and summing the
agent's range over the hands LBR beats, once per candidate action (LocalLBRWorker.py:379-512) -- is ONE call per LBR decision
into the HIP library (prl_lbr_checkdown_equity), batched over the candidate ranges. No CPU fallback: without the library /
a device the worker raises.
"""
import ctypes

import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.eval.lbr import _util
from pokerrl_amd.game.Poker import Poker
from pokerrl_amd.game.PokerRange import PokerRange


class LocalLBRWorker:
    def __init__(self, t_prof, chief_handle, eval_agent_cls):
        assert t_prof.n_seats == 2
        self.t_prof = t_prof
        self.lbr_args = t_prof.module_args["lbr"]
        self._eval_env_bldr = _util.get_env_builder_lbr(t_prof=t_prof)
        self.check_to_round = self.lbr_args.lbr_check_to_round
        self.chief_handle = chief_handle
        self.agent = _AgentWrapper(t_prof=t_prof, lbr_args=self.lbr_args, eval_agent_cls=eval_agent_cls)
        self._env = None  # LBR's env: the agent's game with LBR's bet sizes
        self.agent_range = PokerRange(env_bldr=self._eval_env_bldr)
        assert self.check_to_round is None or (self.check_to_round in self._eval_env_bldr.rules.ALL_ROUNDS_LIST)
        self._rules_struct = self._eval_env_bldr.env_cls.native_rules()
        self.n_equity_calls = 0

    def run(self, agent_seat_id, n_iterations, mode, stack_size):
        """Per-hand winnings of LBR (a lower bound of the agent's exploitability), float32 [n_iterations]; None if the
        agent cannot play `mode`."""
        self.agent.set_mode(mode=mode)
        self.agent.to_stack_size(stack_size)
        self.agent_range.reset()
        self._env = self._eval_env_bldr.get_new_env(is_evaluating=True, stack_size=stack_size)
        if not self.agent.can_compute_mode():
            return None
        limit = self._eval_env_bldr.env_cls.IS_FIXED_LIMIT_GAME
        winnings = np.empty(shape=n_iterations, dtype=np.float32)
        for i in range(n_iterations):
            winnings[i] = self._play_hand(agent_seat_id=agent_seat_id, limit=limit)
        return winnings

    def update_weights(self, weights_for_eval_agent):
        self.agent.update_weights(weights_for_eval_agent)

    # ------------------------------------------------------------------------------------------------------------------
    def _reset_episode(self):
        ret = self._env.reset()
        self.agent.reset(deck_state_dict=self._env.cards_state_dict())
        self.agent_range.reset()
        return ret

    def _play_hand(self, agent_seat_id, limit):
        env, lbr_seat = self._env, 1 - agent_seat_id
        _obs, reward, done, _info = self._reset_episode()
        lbr_hand = env.get_hole_cards_of_player(p_id=lbr_seat)
        self.agent_range.set_cards_to_zero_prob(cards_2d=lbr_hand)
        while not done:
            raise_frac = None
            if env.current_player.seat_id == lbr_seat:
                if (self.check_to_round is not None) and (env.current_round < self.check_to_round):
                    action = Poker.CHECK_CALL
                else:
                    action = self._lbr_action(agent_seat_id=agent_seat_id, lbr_hand=lbr_hand, limit=limit)
                if (not limit) and action >= 2:
                    raise_frac = env.bet_sizes_list_as_frac_of_pot[action - 2]
                    self.agent.notify_of_raise_frac_action(p_id_acted=lbr_seat, frac=raise_frac)
                else:
                    self.agent.notify_of_action(p_id_acted=lbr_seat, action_he_did=action)
            else:
                action, a_probs = self.agent.get_action(step_env=True, need_probs=True)
                self.agent_range.update_after_action(action=action, all_a_probs_for_all_hands=a_probs)
                if (not limit) and action >= 2:  # the agent's bet sizes may differ from LBR's: step by pot fraction
                    raise_frac = self.agent.cpu_agent.env_bldr.env_args.bet_sizes_list_as_frac_of_pot[action - 2]
            old_round = env.current_round
            if raise_frac is not None:
                _obs, reward, done, _info = env.step_raise_pot_frac(pot_frac=raise_frac)
            else:
                _obs, reward, done, _info = env.step(action=action)
            if env.current_round != old_round:
                self.agent_range.update_after_new_round(new_round=env.current_round, board_now_2d=env.board)
        return reward[lbr_seat] * env.REWARD_SCALAR * env.EV_NORMALIZER

    def _lbr_action(self, agent_seat_id, lbr_hand, limit):
        """argmax over LBR's actions of the one-step look-ahead value with check-down equities (LocalLBRWorker.py:91-154, :205-270)."""
        env, lbr_seat = self._env, 1 - agent_seat_id
        n_u = 3 if limit else 2 + len(env.bet_sizes_list_as_frac_of_pot)
        utility = np.full(shape=n_u, fill_value=-1.0, dtype=np.float32)  # illegal: -1, fold: 0
        utility[Poker.FOLD] = 0.0
        asked = env.seats[agent_seat_id].current_bet - env.seats[lbr_seat].current_bet
        pot_before = env.get_all_winnable_money()

        legal = env.get_legal_actions()
        raises = [a for a in legal if a not in (Poker.FOLD, Poker.CHECK_CALL)] if not limit else ([Poker.BET_RAISE] if Poker.BET_RAISE in legal else [])
        cand_ranges = [np.copy(self.agent_range.range)]
        sims = []
        if raises:
            saved_env, saved_agent_env, saved_range = env.state_dict(), self.agent.env_state_dict(), self.agent_range.state_dict()
            for r in raises:
                env.step(action=r)
                pot_after = env.get_all_winnable_money()
                if limit:
                    self.agent.notify_of_action(p_id_acted=lbr_seat, action_he_did=r)
                    _, a_probs = self.agent.get_action(step_env=False, need_probs=True)
                else:
                    self.agent.notify_of_raise_frac_action(p_id_acted=lbr_seat, frac=env.bet_sizes_list_as_frac_of_pot[r - 2])
                    a_probs = self.agent.get_a_probs_for_each_hand()
                fold_prob = np.sum(self.agent_range.range * a_probs[:, Poker.FOLD])
                self.agent_range.mul_and_norm(1 - a_probs[:, Poker.FOLD])
                cand_ranges.append(np.copy(self.agent_range.range))
                sims.append((r, pot_after, fold_prob))
                self.agent_range.load_state_dict(saved_range)
                env.load_state_dict(saved_env)
                self.agent.load_env_state_dict(saved_agent_env)

        wps = self._checkdown_equity(lbr_hand_2d=lbr_hand, ranges=np.stack(cand_ranges))
        wp = wps[0]
        utility[Poker.CHECK_CALL] = wp * pot_before - (1 - wp) * asked
        for (r, pot_after, fold_prob), wp_now in zip(sims, wps[1:]):
            chips_in = pot_after - pot_before
            ev_if_not_fold = (wp_now * pot_after) - ((1 - wp_now) * chips_in)
            utility[r] = fold_prob * pot_before + (1 - fold_prob) * ev_if_not_fold
        return int(np.argmax(utility))

    def _checkdown_equity(self, lbr_hand_2d, ranges):
        """float32 [n]: P(LBR wins at showdown if the hand is checked down) for each candidate agent range."""
        lut = self._eval_env_bldr.lut_holder
        board_1d = np.asarray(lut.get_1d_cards(self._env.board))
        dealt = np.ascontiguousarray([c for c in board_1d if c != Poker.CARD_NOT_DEALT_TOKEN_1D], dtype=np.int8)
        hand_1d = np.ascontiguousarray(lut.get_1d_cards(cards_2d=lbr_hand_2d), dtype=np.int8)
        rg = np.ascontiguousarray(ranges, dtype=np.float32)
        out = np.zeros(rg.shape[0], dtype=np.float32)
        L = _native.lib()
        _native.require_device()
        _native.check(L.prl_lbr_checkdown_equity(ctypes.byref(self._rules_struct), dealt.ctypes.data_as(ctypes.c_void_p), int(dealt.shape[0]),
                                                 hand_1d.ctypes.data_as(ctypes.c_void_p), rg.ctypes.data_as(ctypes.c_void_p),
                                                 int(rg.shape[0]), out.ctypes.data_as(ctypes.c_void_p)), L)
        self.n_equity_calls += 1
        return out


class _AgentWrapper:
    """The agent LBR plays against (LocalLBRWorker.py:311-376): a host EvalAgent with its own internal env."""

    def __init__(self, t_prof, lbr_args, eval_agent_cls):
        self.cpu_agent = eval_agent_cls(t_prof=t_prof, device=None)

    def get_action(self, step_env, need_probs):
        return self.cpu_agent.get_action(step_env=step_env, need_probs=need_probs)

    def get_a_probs_for_each_hand(self):
        return self.cpu_agent.get_a_probs_for_each_hand()

    def get_mode(self):
        return self.cpu_agent.get_mode()

    def set_mode(self, mode):
        self.cpu_agent.set_mode(mode)

    def to_stack_size(self, stack_size):
        self.cpu_agent.set_stack_size(stack_size=stack_size)

    def can_compute_mode(self):
        return self.cpu_agent.can_compute_mode()

    def update_weights(self, w):
        self.cpu_agent.update_weights(w)

    def reset(self, deck_state_dict):
        self.cpu_agent.reset(deck_state_dict=deck_state_dict)

    def notify_of_action(self, p_id_acted, action_he_did):
        self.cpu_agent.notify_of_action(p_id_acted=p_id_acted, action_he_did=action_he_did)

    def notify_of_raise_frac_action(self, p_id_acted, frac):
        self.cpu_agent.notify_of_raise_frac_action(p_id_acted=p_id_acted, frac=frac)

    def env_state_dict(self):
        return self.cpu_agent.env_state_dict()

    def load_env_state_dict(self, state_dict):
        self.cpu_agent.load_env_state_dict(state_dict)
