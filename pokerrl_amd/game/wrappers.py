"""
Env builders and observation wrappers: the objects through which CFR / BR / LBR and agents obtain a game's rules, LUTs, argument
set and -- for neural agents -- the observation (history) a network is fed.

Reference: PokerRL/game/_/EnvWrapperBuilderBase.py:7-111, PokerRL/game/wrappers.py:18-68 and the three wrapper classes
PokerRL/game/_/wrappers/{Vanilla.py, RecurrentHistoryWrapper.py, FlatHULimitPokerHistoryWrapper.py} (+ _Wrapper.py). Same class
names, constructor arguments, attributes (pub_obs_size, priv_obs_size, obs_*_idxs, action_vector_size, get_vector_idx ...) and
semantics: a wrapper steps its env, pushes every transition into its history, and `set_to_public_tree_node_state(node)` rebuilds
the history a player would have seen on the way from the root to `node` (RecurrentHistoryWrapper.py:57-85,
FlatHULimitPokerHistoryWrapper.py:93-114) -- that is what positions a recurrent / feed-forward agent on a public-tree node for
StrategyFiller._fill_with_agent_policy (StrategyFiller.py:88-116). `history_of_nodes` builds those histories for MANY nodes with
one walk of the tree (shared prefixes are computed once), the host half of batched agent querying.

One deliberate difference: VanillaWrapper.set_to_public_tree_node_state loads the node's public state into the env; the
reference's is a no-op (Vanilla.py:38-39), which leaves an agent's env wherever it was.
"""
import copy

import numpy as np

from pokerrl_amd.game.Poker import Poker
from pokerrl_amd.game.PokerEnvStateDictEnums import EnvDictIdxs


# ---------------------------------------------------------------------------------------------------------------------
# wrappers
# ---------------------------------------------------------------------------------------------------------------------
class Wrapper:
    """_Wrapper.py:7-89"""

    def __init__(self, env, env_bldr_that_built_me):
        self.env = env
        self.env_bldr = env_bldr_that_built_me

    def _return_obs(self, rew_for_all_players, done, info, env_obs=None):
        return self.get_current_obs(env_obs=env_obs), rew_for_all_players, done, info

    def step(self, action):
        env_obs, rew, done, info = self.env.step(action)
        self._pushback(env_obs)
        return self._return_obs(env_obs=env_obs, rew_for_all_players=rew, done=done, info=info)

    def step_from_processed_tuple(self, action):
        env_obs, rew, done, info = self.env.step_from_processed_tuple(action)
        self._pushback(env_obs)
        return self._return_obs(env_obs=env_obs, rew_for_all_players=rew, done=done, info=info)

    def step_raise_pot_frac(self, pot_frac):
        env_obs, rew, done, info = self.env.step_raise_pot_frac(pot_frac=pot_frac)
        self._pushback(env_obs)
        return self._return_obs(env_obs=env_obs, rew_for_all_players=rew, done=done, info=info)

    def reset(self, deck_state_dict=None):
        env_obs, rew, done, info = self.env.reset(deck_state_dict=deck_state_dict)
        self._reset_state()
        self._pushback(env_obs)
        return self._return_obs(env_obs=env_obs, rew_for_all_players=rew, done=done, info=info)

    def state_dict(self):
        return {"env": self.env.state_dict()}

    def load_state_dict(self, state_dict):
        self.env.load_state_dict(state_dict["env"])

    def _reset_state(self):
        raise NotImplementedError

    def _pushback(self, env_obs):
        raise NotImplementedError

    def get_current_obs(self, env_obs):
        raise NotImplementedError

    def set_to_public_tree_node_state(self, node):
        raise NotImplementedError


class VanillaWrapper(Wrapper):
    """No history: the current env observation (Vanilla.py:7-39). Feed-forward agents without recall."""

    def _reset_state(self, **kwargs):
        pass

    def _pushback(self, env_obs):
        pass

    def get_current_obs(self, env_obs=None):
        return env_obs if env_obs is not None else self.env.get_current_obs(is_terminal=False)

    def state_dict(self):
        return {"base": super().state_dict()}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict=state_dict["base"] if "base" in state_dict else state_dict)

    def set_to_public_tree_node_state(self, node):
        self.env.load_state_dict(node.env_state, blank_private_info=True)


def _path_to_root(node):
    """nodes from the root down to `node` at which a PLAYER acts next (chance-pending nodes are skipped: nobody observes them)"""
    seq = []
    n = node
    while n is not None:
        if n.p_id_acting_next != n.tree.CHANCE_ID:
            seq.insert(0, n)
        n = n.parent
    return seq


class RecurrentHistoryWrapper(Wrapper):
    """Perfect recall for recurrent networks: the observation is the SEQUENCE of env observations of the episode, float32
    [T, pub_obs_size] (RecurrentHistoryWrapper.py:11-85)."""

    def __init__(self, env, env_bldr_that_built_me):
        super().__init__(env=env, env_bldr_that_built_me=env_bldr_that_built_me)
        self.invert_history_order = env_bldr_that_built_me.invert_history_order
        self._list_of_obs_this_episode = None

    def _reset_state(self, **kwargs):
        self._list_of_obs_this_episode = []

    def _pushback(self, env_obs):
        if self.invert_history_order:
            self._list_of_obs_this_episode.insert(0, np.copy(env_obs))
        else:
            self._list_of_obs_this_episode.append(np.copy(env_obs))

    def get_current_obs(self, env_obs=None):
        return np.array(self._list_of_obs_this_episode, dtype=np.float32)

    def state_dict(self):
        return {"base": super().state_dict(), "obs_seq": copy.deepcopy(self._list_of_obs_this_episode)}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict=state_dict["base"])
        self._list_of_obs_this_episode = copy.deepcopy(state_dict["obs_seq"])

    def set_to_public_tree_node_state(self, node):
        state_seq = []
        for n in _path_to_root(node):  # RecurrentHistoryWrapper.py:66-73: the obs of every player-node on the path, root first
            self.env.load_state_dict(n.env_state)
            state_seq.append(self.env.get_current_obs(is_terminal=False))
        self.reset()
        self._reset_state()  # .reset() pushed the root obs; the sequence below has it already
        for obs in state_seq:
            self._pushback(env_obs=obs)
        self.env.load_state_dict(node.env_state, blank_private_info=True)
        assert np.array_equal(node.env_state[EnvDictIdxs.board_2d], self.env.board)


class FlatHULimitPokerHistoryWrapper(Wrapper):
    """Heads-up fixed-limit games, feed-forward networks: current env obs + a one-hot action-history vector indexed by
    (round, seat, n-th action of that seat this round, action) as in https://arxiv.org/abs/1603.01121
    (FlatHULimitPokerHistoryWrapper.py:12-114)."""

    def __init__(self, env, env_bldr_that_built_me):
        assert env.N_SEATS == 2
        super().__init__(env=env, env_bldr_that_built_me=env_bldr_that_built_me)
        self._action_vector_size = env_bldr_that_built_me.action_vector_size
        self._action_count_this_round = None
        self._game_round_last_tick = None
        self._action_history_vector = None

    def _reset_state(self, **kwargs):
        self._action_count_this_round = [0, 0]
        self._game_round_last_tick = Poker.PREFLOP
        self._action_history_vector = np.zeros(shape=self._action_vector_size, dtype=np.float32)

    def _pushback(self, env_obs=None):
        # the last action still belongs to the round it was made in; the new-round bookkeeping starts with the NEXT transition
        last_a = self.env.last_action[0]
        if last_a is not None:
            actor = self.env.last_action[2]
            idx = self.env_bldr.get_vector_idx(round_=self._game_round_last_tick, p_id=actor,
                                               nth_action_this_round=self._action_count_this_round[actor], action_idx=last_a)
            self._action_history_vector[idx] = 1
            self._action_count_this_round[actor] += 1
            if self.env.current_round != self._game_round_last_tick:
                self._game_round_last_tick = self.env.current_round
                self._action_count_this_round = [0, 0]

    def get_current_obs(self, env_obs=None):
        if env_obs is None:
            env_obs = self.env.get_current_obs(is_terminal=False)
        return np.concatenate((env_obs, self._action_history_vector), axis=0)

    def state_dict(self):
        return {"base": super().state_dict(), "a_seq": np.copy(self._action_history_vector),
                "game_round_last_tick": self._game_round_last_tick, "action_count_this_round": copy.deepcopy(self._action_count_this_round)}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict=state_dict["base"])
        self._action_history_vector = np.copy(state_dict["a_seq"])
        self._game_round_last_tick = state_dict["game_round_last_tick"]
        self._action_count_this_round = copy.copy(state_dict["action_count_this_round"])

    def set_to_public_tree_node_state(self, node):
        self.reset()
        self._reset_state()
        for n in _path_to_root(node):  # FlatHULimitPokerHistoryWrapper.py:100-114: replay the public states, root first
            self.env.load_state_dict(n.env_state, blank_private_info=True)
            self._pushback()


# ---------------------------------------------------------------------------------------------------------------------
# builders
# ---------------------------------------------------------------------------------------------------------------------
class EnvWrapperBuilderBase:
    WRAPPER_CLS = None

    def __init__(self, env_cls, env_args):
        self.env_cls = env_cls
        self.env_args = env_args
        self.rules = env_cls.RULES
        self.lut_holder = env_cls.get_lut_holder()
        self.N_SEATS = env_args.n_seats
        self.N_ACTIONS = env_args.N_ACTIONS
        # observation layout (EnvWrapperBuilderBase.py:31-38; the heads-up "simple" observation of PokerEnv.py:199-261)
        self.pub_obs_size = self._get_num_public_observation_features()
        self.priv_obs_size = self._get_num_private_observation_features()
        self.complete_obs_size = self.pub_obs_size + self.priv_obs_size
        self.obs_board_idxs, self.obs_players_idxs, self.obs_table_state_idxs = self._get_obs_parts_idxs()
        self.obs_size_board = len(self.obs_board_idxs)
        self.obs_size_player_info_each = len(self.obs_players_idxs[0])
        self.obs_size_table_state = len(self.obs_table_state_idxs)

    def _env_obs_size(self):
        r = self.rules
        n_rounds = max(r.ALL_ROUNDS_LIST) + 1
        return 7 + 3 + self.N_SEATS + self.N_SEATS + n_rounds + 3 * self.N_SEATS + r.N_TOTAL_BOARD_CARDS * (r.N_RANKS + r.N_SUITS)

    def _get_num_public_observation_features(self):
        return self._env_obs_size()

    def _get_num_private_observation_features(self):
        return (self.rules.N_SUITS + self.rules.N_RANKS) * self.rules.N_HOLE_CARDS

    def _get_obs_parts_idxs(self):
        r = self.rules
        n_table = 7 + 3 + self.N_SEATS + self.N_SEATS + max(r.ALL_ROUNDS_LIST) + 1
        players = [list(range(n_table + 3 * p, n_table + 3 * (p + 1))) for p in range(self.N_SEATS)]
        b0 = n_table + 3 * self.N_SEATS
        return list(range(b0, b0 + r.N_TOTAL_BOARD_CARDS * (r.N_RANKS + r.N_SUITS))), players, list(range(n_table))

    def get_new_env(self, is_evaluating, stack_size=None):
        from pokerrl_amd.game.poker_env import PokerEnv
        args = copy.deepcopy(self.env_args)
        if stack_size is not None:
            assert isinstance(stack_size, list)
            args.starting_stack_sizes_list = copy.deepcopy(stack_size)
        return PokerEnv(env_cls=self.env_cls, env_args=args, lut_holder=self.lut_holder, is_evaluating=is_evaluating)

    def get_new_wrapper(self, is_evaluating, init_from_env=None, stack_size=None):
        env = init_from_env if init_from_env is not None else self.get_new_env(is_evaluating=is_evaluating, stack_size=stack_size)
        return self.WRAPPER_CLS(env=env, env_bldr_that_built_me=self)


class VanillaEnvBuilder(EnvWrapperBuilderBase):
    WRAPPER_CLS = VanillaWrapper


class HistoryEnvBuilder(EnvWrapperBuilderBase):
    WRAPPER_CLS = RecurrentHistoryWrapper

    def __init__(self, env_cls, env_args, invert_history_order=False):
        super().__init__(env_cls=env_cls, env_args=env_args)
        self.invert_history_order = invert_history_order


class FlatLimitPokerEnvBuilder(EnvWrapperBuilderBase):
    WRAPPER_CLS = FlatHULimitPokerHistoryWrapper

    def __init__(self, env_cls, env_args):
        assert env_cls.IS_FIXED_LIMIT_GAME
        assert env_args.n_seats == 2
        self._VEC_ROUND_OFFSETS, self._VEC_HALF_ROUND_SIZE = {}, {}
        self.action_vector_size = 0
        for r in env_cls.RULES.ALL_ROUNDS_LIST:  # wrappers.py:38-46
            self._VEC_ROUND_OFFSETS[r] = self.action_vector_size
            self._VEC_HALF_ROUND_SIZE[r] = len([Poker.BET_RAISE, Poker.CHECK_CALL]) * (env_cls.MAX_N_RAISES_PER_ROUND[r] + 2)
            self.action_vector_size += self._VEC_HALF_ROUND_SIZE[r] * env_args.n_seats
        super().__init__(env_cls=env_cls, env_args=env_args)

    def _get_num_public_observation_features(self):
        return self._env_obs_size() + self.action_vector_size

    def get_vector_idx(self, round_, p_id, nth_action_this_round, action_idx):
        # "- 1": fold (action 0) is never recorded -- nobody observes the state after it (wrappers.py:54-61)
        return self._VEC_ROUND_OFFSETS[round_] + p_id * self._VEC_HALF_ROUND_SIZE[round_] + nth_action_this_round * 2 + action_idx - 1


ALL_BUILDERS = [HistoryEnvBuilder, FlatLimitPokerEnvBuilder, VanillaEnvBuilder]


# ---------------------------------------------------------------------------------------------------------------------
# batched positioning: the observation (history) of MANY public-tree nodes in one tree walk
# ---------------------------------------------------------------------------------------------------------------------
def history_of_nodes(env_bldr, nodes, stack_size=None):
    """What `wrapper.set_to_public_tree_node_state(node); wrapper.get_current_obs()` returns, for every node of `nodes` (decision
    nodes of ONE PublicTree), computed with one pass over the union of their root paths instead of one full replay per node.
    Returns a list of arrays: [T_i, pub_obs_size] (HistoryEnvBuilder) or [pub_obs_size] (Flat / Vanilla). Nodes of equal depth in
    the betting tree of one board have equal T, which is what lets an agent stack them into one forward per depth."""
    w = env_bldr.get_new_wrapper(is_evaluating=True, stack_size=stack_size)
    env = w.env
    cache = {}  # node index -> wrapper history state after observing that node

    def obs_of(n):
        env.load_state_dict(n.env_state, blank_private_info=True)
        return env.get_current_obs(is_terminal=False)

    def state_after(n):
        """history state of the wrapper after the player-nodes on the path root .. n have been pushed"""
        key = n._i
        if key in cache:
            return cache[key]
        p = n.parent
        while p is not None and p.p_id_acting_next == p.tree.CHANCE_ID:
            p = p.parent
        base = state_after(p) if p is not None else None
        if isinstance(w, RecurrentHistoryWrapper):
            o = obs_of(n)
            st = ([o] + base) if (base is not None and w.invert_history_order) else ((base or []) + [o])
        elif isinstance(w, FlatHULimitPokerHistoryWrapper):
            w._reset_state()
            if base is not None:
                w._action_history_vector, w._game_round_last_tick, w._action_count_this_round = np.copy(base[0]), base[1], list(base[2])
            env.load_state_dict(n.env_state, blank_private_info=True)
            w._pushback()
            st = (np.copy(w._action_history_vector), w._game_round_last_tick, list(w._action_count_this_round))
        else:
            st = None
        cache[key] = st
        return st

    out = []
    for n in nodes:
        st = state_after(n)
        if isinstance(w, RecurrentHistoryWrapper):
            out.append(np.array(st, dtype=np.float32))
        elif isinstance(w, FlatHULimitPokerHistoryWrapper):
            out.append(np.concatenate((obs_of(n), st[0]), axis=0))
        else:
            out.append(obs_of(n))
    return out
