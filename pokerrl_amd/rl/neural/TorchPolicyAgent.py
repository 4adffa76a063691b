"""
A PyTorch(-ROCm) policy agent behind the EvalAgent API, with BATCHED querying over public-tree nodes.

The reference positions an agent on one node at a time and asks for `[RANGE_SIZE, N_ACTIONS]` probabilities
(StrategyFiller.py:88-116, EvalAgentBase.py:39-44,129-130): one network forward per decision node, each preceded by a replay of
the node's observation history (RecurrentHistoryWrapper.py:57-85). Once the tree arithmetic is on the GPU that loop is the whole
cost of a best-response evaluation. Here
  * `get_a_probs_for_each_hand()` is the reference protocol (the agent's internal wrapper was positioned by the caller), and
  * `get_a_probs_for_each_hand_in_nodes(nodes)` answers for ALL given decision nodes: their histories come from one walk of the
    tree (wrappers.history_of_nodes), nodes with the same history length are stacked, and the network runs ONE forward per
    length: public trunk [n_nodes, T, pub_obs] -> [n_nodes, H], private trunk [RANGE_SIZE, priv_obs] -> [R, H] once, head on the
    [n_nodes, R] cross product. PublicTree.fill_with_agent_policy picks this up automatically.
The network is a plain module (GRU or MLP public trunk + linear private trunk + MLP head, softmax over the legal actions); weights
come from `update_weights(state_dict)` / the agent pickle. It stands for the neural agents of the reference's downstream projects
(Deep CFR, NFSP: PokerRL/rl/neural/*): those are out of this repository's scope, this class is what exercises the path.
"""
import numpy as np
import torch
import torch.nn as nn

from pokerrl_amd.game import wrappers as W
from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase


class PolicyNet(nn.Module):
    def __init__(self, pub_obs_size, priv_obs_size, n_actions, recurrent, hidden=64):
        super().__init__()
        self.recurrent = recurrent
        self.pub = nn.GRU(pub_obs_size, hidden, batch_first=True) if recurrent else nn.Sequential(nn.Linear(pub_obs_size, hidden), nn.ReLU())
        self.priv = nn.Sequential(nn.Linear(priv_obs_size, hidden), nn.ReLU())
        self.head = nn.Sequential(nn.Linear(2 * hidden, hidden), nn.ReLU(), nn.Linear(hidden, n_actions))

    def forward(self, pub_obs, priv_obs_all, legal_mask):
        """pub_obs [B, T, pub] (recurrent) or [B, pub]; priv_obs_all [R, priv]; legal_mask bool [B, A] -> probabilities [B, R, A]"""
        hp = self.pub(pub_obs)[1][-1] if self.recurrent else self.pub(pub_obs)  # [B, H]
        hr = self.priv(priv_obs_all)                                             # [R, H]
        B, R = hp.shape[0], hr.shape[0]
        x = torch.cat((hp[:, None, :].expand(B, R, -1), hr[None, :, :].expand(B, R, -1)), dim=-1)
        logits = self.head(x).masked_fill(~legal_mask[:, None, :], float("-inf"))
        return torch.softmax(logits, dim=-1)


class TorchPolicyAgent(EvalAgentBase):
    ALL_MODES = ["POLICY"]
    HIDDEN = 64
    SEED = 0
    MAX_ROWS_PER_FORWARD = 1 << 22  # n_nodes x RANGE_SIZE rows of one forward (the head's activations are rows x 128 floats)

    def __init__(self, t_prof, mode=None, device=None):
        super().__init__(t_prof=t_prof, mode=mode, device=device)
        b = self.env_bldr
        self._recurrent = isinstance(b, W.HistoryEnvBuilder)
        gen = torch.Generator().manual_seed(self.SEED)
        with torch.random.fork_rng():
            torch.manual_seed(int(gen.initial_seed()))
            self._net = PolicyNet(b.pub_obs_size, b.priv_obs_size, b.N_ACTIONS, self._recurrent, self.HIDDEN)
        self._net.to(self.device).eval()
        self._priv = torch.from_numpy(b.lut_holder.LUT_RANGE_IDX_TO_PRIVATE_OBS).to(self.device)
        self.n_forwards = 0

    # ---- EvalAgentBase protocol ---------------------------------------------------------------------------------------
    def can_compute_mode(self):
        return True

    def update_weights(self, weights_for_eval_agent):
        if weights_for_eval_agent is not None:
            self._net.load_state_dict({k: torch.as_tensor(v) for k, v in weights_for_eval_agent.items()})
            self._net.to(self.device).eval()

    def _state_dict(self):
        return {"net": {k: v.detach().cpu().numpy() for k, v in self._net.state_dict().items()}}

    def _load_state_dict(self, state):
        self.update_weights(state["net"])

    def _legal_mask(self, legal_lists):
        m = torch.zeros((len(legal_lists), self.env_bldr.N_ACTIONS), dtype=torch.bool)
        for i, legal in enumerate(legal_lists):
            m[i, legal] = True
        return m.to(self.device)

    @torch.no_grad()
    def _forward_t(self, pub_obs, legal_lists):
        """the forward's output where it is computed: float32 tensor [B, R, A] on self.device"""
        self.n_forwards += 1
        pub = torch.as_tensor(np.ascontiguousarray(pub_obs, dtype=np.float32)).to(self.device)
        return self._net(pub, self._priv, self._legal_mask(legal_lists)).float()

    def _forward(self, pub_obs, legal_lists):
        return self._forward_t(pub_obs, legal_lists).cpu().numpy()

    def get_a_probs_for_each_hand(self):
        """reference protocol: the internal wrapper has been positioned (set_to_public_tree_node_state / steps)"""
        w = self._internal_env_wrapper
        obs = w.get_current_obs()
        return self._forward(obs[None], [w.env.get_legal_actions()])[0]

    def get_a_probs(self):
        env = self._internal_env_wrapper.env
        return self.get_a_probs_for_each_hand()[env.get_range_idx(p_id=env.current_player.seat_id)]

    def get_action(self, step_env=True, need_probs=False):
        env = self._internal_env_wrapper.env
        all_p = self.get_a_probs_for_each_hand()
        p = all_p[env.get_range_idx(p_id=env.current_player.seat_id)].astype(np.float64)
        action = int(np.random.choice(len(p), p=p / p.sum()))
        if step_env:
            self._internal_env_wrapper.step(action=action)
        return action, (all_p if need_probs else None)

    # ---- batched protocol (PublicTree.fill_with_agent_policy) -----------------------------------------------------------
    DEVICE_RESIDENT_FILL = True  # PublicTree.fill_with_agent_policy keeps this agent's probabilities in HBM (False: the host path, for comparison)

    def get_a_probs_for_each_hand_in_nodes_device(self, nodes):
        """the same numbers as get_a_probs_for_each_hand_in_nodes, left where the network put them: a float32 tensor [len(nodes), R, A] on self.device
        (one forward per history length writes its rows of the output tensor), synchronised -- PublicTree hands its address to the solver, which
        scatters it into its strategy columns on the GPU. None (declined) when the agent is told to use the host path."""
        if not self.DEVICE_RESIDENT_FILL or not nodes:
            return None
        R, A = self.env_bldr.rules.RANGE_SIZE, self.env_bldr.N_ACTIONS
        out = torch.zeros((len(nodes), R, A), dtype=torch.float32, device=self.device)
        self._fill_nodes(nodes, lambda part, probs: out.index_copy_(0, torch.as_tensor(part, device=self.device), probs), self._forward_t)
        if out.is_cuda:
            torch.cuda.synchronize(out.device)
        return out

    def get_a_probs_for_each_hand_in_nodes(self, nodes):
        """float32 [len(nodes), RANGE_SIZE, N_ACTIONS] for decision nodes of one PublicTree: one forward per history length"""
        R, A = self.env_bldr.rules.RANGE_SIZE, self.env_bldr.N_ACTIONS
        out = np.zeros((len(nodes), R, A), dtype=np.float32)
        if not nodes:
            return out

        def put(part, probs):
            out[part] = probs

        self._fill_nodes(nodes, put, self._forward)
        return out

    def _fill_nodes(self, nodes, put, forward):
        R = self.env_bldr.rules.RANGE_SIZE
        hist = W.history_of_nodes(self.env_bldr, nodes, stack_size=list(nodes[0].tree.stack_size))
        groups = {}
        for i, h in enumerate(hist):
            groups.setdefault(h.shape, []).append(i)
        chunk = max(1, self.MAX_ROWS_PER_FORWARD // R)
        for idxs in groups.values():
            for lo in range(0, len(idxs), chunk):
                part = idxs[lo:lo + chunk]
                put(part, forward(np.stack([hist[i] for i in part]), [nodes[i].allowed_actions for i in part]))
