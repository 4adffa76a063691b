"""
EvalAgentBase: the agent-query API evaluators program against. Method names and semantics follow the reference
(PokerRL/rl/base_cls/EvalAgentBase.py:9-170): an agent wraps its own internal env, can be positioned on a public-tree
node and asked for `[RANGE_SIZE, N_ACTIONS]` action probabilities. Subclass it exactly as with the reference.
"""
import os
import pickle

from pokerrl_amd.rl import rl_util


class EvalAgentBase:
    ALL_MODES = NotImplementedError  # override with the list of modes

    def __init__(self, t_prof, mode=None, device=None):
        self.t_prof = t_prof
        self.env_bldr = rl_util.get_env_builder(t_prof=t_prof)
        self._internal_env_wrapper = self.env_bldr.get_new_wrapper(is_evaluating=True, stack_size=None)
        self._mode = mode
        self.device = t_prof.device_inference if device is None else device

    # ---- to be implemented by the algorithm ---------------------------------------------------------------------------
    def get_a_probs_for_each_hand(self):
        raise NotImplementedError

    def get_a_probs(self):
        raise NotImplementedError

    def get_action(self, step_env=True, need_probs=False):
        raise NotImplementedError

    def get_action_frac_tuple(self, step_env=True):
        raise NotImplementedError

    def _state_dict(self):
        raise NotImplementedError

    def _load_state_dict(self, state):
        raise NotImplementedError

    def update_weights(self, weights_for_eval_agent):
        raise NotImplementedError

    def can_compute_mode(self):
        raise NotImplementedError

    # ---- shared behaviour ---------------------------------------------------------------------------------------------
    def state_dict(self):
        return {"t_prof": self.t_prof, "mode": self._mode, "env": self._internal_env_wrapper.state_dict(), "agent": self._state_dict()}

    def load_state_dict(self, state):
        self._internal_env_wrapper.load_state_dict(state["env"])
        self._mode = state["mode"]
        self._load_state_dict(state["agent"])

    def set_stack_size(self, stack_size):
        self._internal_env_wrapper.env.set_stack_size(stack_size=stack_size)

    def get_mode(self):
        return self._mode

    def set_mode(self, mode):
        assert mode in self.ALL_MODES
        self._mode = mode

    def set_env_wrapper(self, env_wrapper):
        self._internal_env_wrapper = env_wrapper

    def get_env_wrapper(self):
        return self._internal_env_wrapper

    def set_to_public_tree_node_state(self, node):
        self._internal_env_wrapper.set_to_public_tree_node_state(node=node)

    def notify_of_action(self, p_id_acted, action_he_did):
        assert self._internal_env_wrapper.env.current_player.seat_id == p_id_acted
        self._internal_env_wrapper.step(action=action_he_did)

    def notify_of_processed_tuple_action(self, p_id_acted, action_he_did):
        assert self._internal_env_wrapper.env.current_player.seat_id == p_id_acted
        self._internal_env_wrapper.env.step_from_processed_tuple(action_he_did)

    def notify_of_raise_frac_action(self, p_id_acted, frac):
        assert self._internal_env_wrapper.env.current_player.seat_id == p_id_acted
        self._internal_env_wrapper.env.step_raise_pot_frac(pot_frac=frac)

    def notify_of_reset(self):
        self._internal_env_wrapper.reset()

    def reset(self, deck_state_dict=None):
        self._internal_env_wrapper.reset(deck_state_dict=deck_state_dict)

    def env_state_dict(self):
        return self._internal_env_wrapper.state_dict()

    def load_env_state_dict(self, state_dict):
        self._internal_env_wrapper.load_state_dict(state_dict)

    def store_to_disk(self, path, file_name):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, str(file_name) + ".pkl"), "wb") as f:
            pickle.dump(self.state_dict(), f, protocol=pickle.HIGHEST_PROTOCOL)

    @classmethod
    def load_from_disk(cls, path_to_eval_agent):
        with open(path_to_eval_agent, "rb") as f:
            state = pickle.load(f)
        agent = cls(t_prof=state["t_prof"])
        agent.load_state_dict(state=state)
        return agent
