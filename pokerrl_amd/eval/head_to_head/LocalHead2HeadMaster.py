"""LocalHead2HeadMaster (PokerRL/eval/head_to_head/LocalHead2HeadMaster.py:10-135): two modes of one
EvalAgent class play heads-up against each other on the native-backed PokerEnv; n_hands with the reference agent in seat 0,
then n_hands with it in seat 1; per-hand winnings of the reference agent in the game's WIN_METRIC, mean +- 95 % confidence
logged under the reference's experiment names. Same constructor / set_modes / update_weights / evaluate; the episodes (deck
draws from np.random, agent notifications, env steps) happen in the reference's order, so for the same seed and agents the
winnings are identical (tests/golden/h2h_*.npz)."""
import numpy as np

from pokerrl_amd.eval._.EvaluatorMasterBase import EvaluatorMasterBase
from pokerrl_amd.rl import rl_util


class LocalHead2HeadMaster(EvaluatorMasterBase):
    def __init__(self, t_prof, chief_handle, eval_agent_cls):
        bldr = rl_util.get_env_builder(t_prof=t_prof)
        assert bldr.N_SEATS == 2, "Only HU supported!"
        EvaluatorMasterBase.__init__(self, t_prof=t_prof, eval_env_bldr=bldr, chief_handle=chief_handle,
                                     eval_type="Head2Head_Winnings", log_conf_interval=True)
        self._args = t_prof.module_args["h2h"]
        self._env_bldr = bldr
        self._eval_agents = [eval_agent_cls(t_prof=t_prof) for _ in range(2)]
        self._REFERENCE_AGENT = 0  # the agent whose winnings are reported

    def set_modes(self, modes):
        for agent, mode in zip(self._eval_agents, modes):
            agent.set_mode(mode)

    def update_weights(self):
        w = self.pull_current_strat_from_chief()
        for agent in self._eval_agents:  # each agent picks what its mode needs out of the dict
            agent.update_weights(w)

    def evaluate(self, iter_nr):
        per_stack = []
        for stack_size_idx, stack_size in enumerate(self._t_prof.eval_stack_sizes):
            for agent in self._eval_agents:
                agent.set_stack_size(stack_size=stack_size)
            if all(agent.can_compute_mode() for agent in self._eval_agents):
                mean, d = self._get_95confidence(self.play(stack_size=stack_size))
                self._log_results(iter_nr=iter_nr, agent_mode=self._eval_agents[self._REFERENCE_AGENT].get_mode(),
                                  stack_size_idx=stack_size_idx, score=mean, upper_conf95=mean + d, lower_conf95=mean - d)
                per_stack.append((mean, d))
        if self._is_multi_stack and per_stack:
            m = sum(x[0] for x in per_stack) / float(len(per_stack))
            d = sum(x[1] for x in per_stack) / float(len(per_stack))
            self._log_multi_stack(agent_mode="Head2Head", iter_nr=iter_nr, score_total=m, lower_conf95=m - d, upper_conf95=m + d)

    def play(self, stack_size):
        """float32 [2 * n_hands]: winnings of the reference agent per hand (_run_eval, :82-126, before the confidence interval)"""
        n = self._args.n_hands
        winnings = np.empty(2 * n, dtype=np.float32)
        env = self._eval_env_bldr.get_new_env(is_evaluating=True, stack_size=stack_size)
        ref, other = self._eval_agents[self._REFERENCE_AGENT], self._eval_agents[1 - self._REFERENCE_AGENT]
        for ref_seat in range(2):
            by_seat = {ref_seat: (ref, other), 1 - ref_seat: (other, ref)}  # seat -> (who acts, who is told)
            for hand in range(n):
                _obs, rewards, done, _info = env.reset()
                deck = env.cards_state_dict()
                for agent in self._eval_agents:
                    agent.reset(deck_state_dict=deck)
                while not done:
                    seat = env.current_player.seat_id
                    actor, listener = by_seat[seat]
                    action, _ = actor.get_action(step_env=True, need_probs=False)
                    listener.notify_of_action(p_id_acted=seat, action_he_did=action)
                    _obs, rewards, done, _info = env.step(action)
                winnings[ref_seat * n + hand] = rewards[ref_seat] * env.REWARD_SCALAR * env.EV_NORMALIZER
        return winnings

    def _run_eval(self, stack_size):
        return self._get_95confidence(self.play(stack_size=stack_size))
