"""Host mirror of the library's synthetic hash agent (csrc/prl_lbr_batch.hip: lbrb_agent_*), as an EvalAgent: the fixture agent of the LBR tests ("seeded hash-based [R, N_ACTIONS] policy"). The same source
drives the reference (golden generation, tests/golden/make_lbr_golden.py) and pokerrl_amd: make_agent_cls binds it to
either package's EvalAgentBase. Deterministic: the policy is an integer hash of the public betting state, hand and action;
the action draw is a counter-based hash of (seed, episode, step) -- no global RNG besides the env's deck shuffle."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    """vectorised 32-bit integer finaliser on uint64 arrays (values kept below 2**32)"""
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def state_key(env, seed):
    k = np.uint64(seed) & M32
    vals = [env.current_round, env.main_pot, env.seats[0].current_bet, env.seats[1].current_bet, env.seats[0].stack,
            env.seats[1].stack, env.current_player.seat_id]
    vals += [int(c) & 0xFF for c in np.asarray(env.board).reshape(-1)]
    for v in vals:
        k = mix32(k * np.uint64(31) + np.uint64(int(v) & 0xFFFFFFFF))
    return k


def policy(env, seed, range_size, n_actions):
    """float32 [R, N_ACTIONS]: weights ((hash >> 8) & 0xFFFF) + 1 on the legal actions, normalised per hand"""
    legal = env.get_legal_actions()
    key = state_key(env, seed)
    h = np.arange(range_size, dtype=np.uint64)[:, None]
    a = np.asarray(legal, dtype=np.uint64)[None, :]
    w = (((mix32(key + h * np.uint64(0x9E3779B1) + a * np.uint64(0x85EBCA6B)) >> np.uint64(8)) & np.uint64(0xFFFF)) + np.uint64(1))
    w = w.astype(np.float32)
    s = np.zeros(range_size, dtype=np.float32)
    for j in range(len(legal)):  # explicit left-to-right float32 sum
        s = s + w[:, j]
    p = np.zeros((range_size, n_actions), dtype=np.float32)
    p[:, legal] = w / s[:, None]
    return p


def make_agent_cls(EvalAgentBase, seed=7, record=None):
    class HashAgent(EvalAgentBase):
        ALL_MODES = ["HASH", "HASH2"]  # head-to-head tests pit the two modes against each other: mode k plays seed + k
        BASE_SEED = seed
        SEED = seed
        RECORD = record  # optional list: deck state of every episode

        def __init__(self, t_prof, mode=None, device=None):
            super().__init__(t_prof=t_prof, mode=mode, device=device)
            self._episode = 0
            self._step = 0

        def can_compute_mode(self):
            return True

        def set_mode(self, mode):  # called once at the start of every LocalLBRWorker.run: the draws restart per run
            super().set_mode(mode)
            self.SEED = self.BASE_SEED + (self.ALL_MODES.index(mode) if mode in self.ALL_MODES else 0)
            self._episode = 0
            self._step = 0

        def update_weights(self, w):
            pass

        def _state_dict(self):
            return {}

        def _load_state_dict(self, s):
            pass

        def reset(self, deck_state_dict=None):
            super().reset(deck_state_dict=deck_state_dict)
            self._episode += 1
            self._step = 0
            if self.RECORD is not None and deck_state_dict is not None:
                self.RECORD.append(deck_state_dict)

        def get_a_probs_for_each_hand(self):
            env = self._internal_env_wrapper.env
            return policy(env, self.SEED, self.env_bldr.rules.RANGE_SIZE, self.env_bldr.N_ACTIONS)

        def get_a_probs(self):
            env = self._internal_env_wrapper.env
            return self.get_a_probs_for_each_hand()[env.get_range_idx(p_id=env.current_player.seat_id)]

        def get_action(self, step_env=True, need_probs=False):
            env = self._internal_env_wrapper.env
            all_p = self.get_a_probs_for_each_hand()
            p = all_p[env.get_range_idx(p_id=env.current_player.seat_id)]
            x = mix32(np.uint64(self.SEED) * np.uint64(0x51ED27) + np.uint64(self._episode) * np.uint64(0x9E3779B1) + np.uint64(self._step))
            u = np.float32(int(x) >> 8) / np.float32(1 << 24)
            self._step += 1
            legal = env.get_legal_actions()
            action, c = legal[-1], np.float32(0.0)
            for a in legal:
                c = np.float32(c + p[a])
                if u < c:
                    action = a
                    break
            if step_env:
                self._internal_env_wrapper.step(action=action)
            return int(action), (all_p if need_probs else None)

    return HashAgent
