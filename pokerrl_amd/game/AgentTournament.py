"""AgentTournament (PokerRL/game/AgentTournament.py:7-73): two EvalAgents play heads-up on the native-backed
PokerEnv -- n_games_per_seat hands with agent 1 in seat 0, then as many with it in seat 1; the episodes (deck draws from np.random,
agent notifications, env steps) happen in the reference's order, so for the same seed and agents `run` returns the reference's numbers
(tests/golden/tournament_*.npz). Same constructor and `run(n_games_per_seat) -> (mean, upper_conf95, lower_conf95)` of agent 1's
winnings in the game's WIN_METRIC; `play` returns the per-hand winnings the three numbers are computed from."""
import numpy as np

from pokerrl_amd.game.poker_env import PokerEnv


class AgentTournament:

    def __init__(self, env_cls, env_args, eval_agent_1, eval_agent_2):
        assert env_args.n_seats == 2
        self._eval_agents = [eval_agent_1, eval_agent_2]
        self._env_cls = env_cls
        self._env_args = env_args
        self._lut_holder = env_cls.get_lut_holder()

    def play(self, n_games_per_seat):
        """float32 [2 * n_games_per_seat]: agent 1's winnings per hand (AgentTournament.py:22-58)"""
        env = PokerEnv(env_cls=self._env_cls, env_args=self._env_args, lut_holder=self._lut_holder, is_evaluating=True)
        first, second = self._eval_agents
        winnings = np.empty(n_games_per_seat * env.N_SEATS, dtype=np.float32)
        for seat_first in range(env.N_SEATS):
            by_seat = {seat_first: (first, second), 1 - seat_first: (second, first)}  # seat -> (who acts, who is told)
            for hand in range(n_games_per_seat):
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
                winnings[hand + seat_first * n_games_per_seat] = rewards[seat_first] * env.REWARD_SCALAR * env.EV_NORMALIZER
        return winnings

    def run(self, n_games_per_seat):
        winnings = self.play(n_games_per_seat)
        mean = np.mean(winnings).item()
        std = np.std(winnings).item()
        d = 1.96 * std / np.sqrt(n_games_per_seat * 2)
        print()
        print("Played", n_games_per_seat * 2, "hands of poker.")
        print("Player ", self._eval_agents[0].get_mode() + ":", mean, "+/-", d)
        print("Player ", self._eval_agents[1].get_mode() + ":", (-mean), "+/-", d)
        return float(mean), float(mean + d), float(mean - d)
