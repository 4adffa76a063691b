"""
PublicTree facade over the device-resident solver.

Same constructor, methods and node attribute protocol as the reference (PokerRL/game/_/tree/PublicTree.py:30-158,
PokerRL/game/_/tree/_/nodes.py:17-41), but the tree is a flat struct-of-arrays built natively (csrc/prl_tree.cpp) and every
pass -- strategy fill, reach push-down, EV / best-response pull-up -- is a HIP kernel. Node objects are thin views: reading
`node.ev`, `node.reach_probs`, `node.strategy` ... copies the corresponding slice out of HBM on demand (lazy host
mirrors); assigning `node.strategy` stages the value and uploads all staged strategies before the next pass.

`stop_at_street` builds the reference's partial tree (nodes of a round >= the limit are not expanded): structure, env states and
legal actions only -- the passes (fill / reach / ev) need the whole tree and raise on a partial one.
For 2-hole-card games the chance outcomes must be given (`boards=`): the reference cannot enumerate them at all
 and the full C(52,5) set does not fit one GPU.
"""
import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.game.Poker import Poker

from pokerrl_amd.game.PokerEnvStateDictEnums import EnvDictIdxs, PlayerDictIdxs

KIND_DECISION, KIND_CHANCE, KIND_FOLD, KIND_SHOWDOWN = 0, 1, 2, 3


class TreeNode:
    """View of one node of the flat tree; attribute names as in the reference's NodeBase (nodes.py:8-41)."""

    def __init__(self, tree, idx):
        self.tree = tree
        self._i = idx
        self.data = None  # algorithms may hang a dict here (nodes.py:40-41); CFR state itself lives in the solver
        self._children = None

    # ---- structure ----------------------------------------------------------------------------------------------------
    @property
    def is_terminal(self):
        return self.tree._kind[self._i] >= KIND_FOLD

    @property
    def p_id_acting_next(self):
        k = self.tree._kind[self._i]
        if k == KIND_DECISION:
            return int(self.tree._actor[self._i])
        return PublicTree.CHANCE_ID if k == KIND_CHANCE else None

    @property
    def p_id_acted_last(self):
        a = int(self.tree._acted_last[self._i])
        return None if a == -1 else (PublicTree.CHANCE_ID if a == -2 else a)

    @property
    def action(self):
        a = int(self.tree._action[self._i])
        return "CHANCE" if a == -1 else a

    @property
    def depth(self):
        return int(self.tree._depth[self._i])

    @property
    def parent(self):
        p = int(self.tree._parent[self._i])
        return None if p < 0 else self.tree.node(p)

    @property
    def children(self):
        if self._children is None:
            t = self.tree
            lo, hi = t._child_start[self._i], t._child_start[self._i + 1]
            self._children = [t.node(int(c)) for c in t._child_list[lo:hi]]
        return self._children

    @property
    def allowed_actions(self):
        t = self.tree
        if t._kind[self._i] != KIND_DECISION:
            return []
        if t._n_children[self._i] == 0:  # unexpanded node of a partial tree (stop_at_street): ask the env, as the reference does
            env = t._get_replay_env()
            env.load_state_dict(self.env_state, blank_private_info=True)
            return [int(a) for a in env.get_legal_actions()]
        c0 = t._first_col[self._i]
        return [int(a) for a in t._col_action[c0:c0 + t._n_children[self._i]]]

    @property
    def env_state(self):
        return self.tree._env_state_of(self._i)

    # ---- per-node vectors (lazy host mirrors of the HBM arrays) -------------------------------------------------------
    @property
    def reach_probs(self):
        return self.tree._vec("reach")[self._i]

    @property
    def ev(self):
        return self.tree._vec("ev")[self._i]

    @property
    def ev_br(self):
        return self.tree._vec("ev_br")[self._i]

    @property
    def ev_weighted(self):
        return self.ev * self.reach_probs

    @property
    def ev_br_weighted(self):
        return self.ev_br * self.reach_probs

    @property
    def epsilon(self):
        return self.ev_br_weighted - self.ev_weighted

    @property
    def exploitability(self):
        if self._i == 0:
            return self.tree._solver.exploitability()  # computed on the GPU in the reference's summation order
        return np.sum(self.epsilon, axis=1)

    @property
    def br_a_idx_in_child_arr_for_each_hand(self):
        return self.tree._vec("br_idx")[self._i]

    @property
    def strategy(self):
        t = self.tree
        if t._kind[self._i] == KIND_CHANCE:
            return t._chance_strategy(self._i)
        if t._kind[self._i] != KIND_DECISION:
            return None
        c0, a = t._first_col[self._i], t._n_children[self._i]
        cols = t._vec("strategy")[c0:c0 + a]
        out = cols.T
        return out if t._vec("strat_f64")[self._i] else out.astype(np.float32)

    @strategy.setter
    def strategy(self, value):
        self.tree._stage_strategy(self._i, np.asarray(value))


class PublicTree:
    CHANCE_ID = "Ch"

    def __init__(self, env_bldr, stack_size, stop_at_street, put_out_new_round_after_limit=False, is_debugging=False, boards=None,
                 engine="levels", n_boards=None, max_outcomes=None, board_seed=None, suit_isomorphism=None, board_mult=None):
        """boards=None: the builder deals the chance outcomes from the deck itself as the reference does (PublicTree.py:188-210) -- every board
        of the game (Leduc: the 6 cards; Flop5Holdem: all C(52,5) five-card boards), or a subset: n_boards= (games that deal once) /
        max_outcomes=(flops, turns, rivers) (games that deal on several streets), the first ones in combinatorial order or, with board_seed=,
        a seeded choice (pokerrl_amd/game/board_enum.py). boards=[[c1..c5], ...] lists them explicitly (run-outs in deal order).
        board_mult= (with explicit boards): weighted boards, board i standing for board_mult[i] boards of the game. They are only treated as SUIT CLASSES
        (orbit-mean chance values: prl_solver_create_weighted's `symmetrize`) when suit_isomorphism says so -- True: the representatives of ALL classes,
        "subset": some of them; the library checks either claim. Any other weighting (sampled boards with importance weights ...) leaves it None / False."""
        self._env_bldr = env_bldr
        self._stack_size = stack_size
        self._is_debugging = is_debugging
        self._put_out_new_round_after_limit = put_out_new_round_after_limit
        # PublicTree.py:72,173: nodes of a round >= stop_at_street stay unexpanded. A partial tree has structure and states only
        # (node.children / env_state / allowed_actions ...): the device solver needs the whole tree, every pass raises.
        self._stop_at_street = max(env_bldr.rules.ALL_ROUNDS_LIST) + 1 if stop_at_street is None else int(stop_at_street)
        self._is_partial = self._stop_at_street <= max(env_bldr.rules.ALL_ROUNDS_LIST)
        self._boards = boards
        self._board_caps = (n_boards, max_outcomes, board_seed)
        # suit isomorphism (board_enum.default_boards_or_classes): class representatives + multiplicities instead of every board; board_mult comes
        # with explicit `boards` that are such representatives
        self._suit_iso, self._board_mult = suit_isomorphism, board_mult
        self._engine = engine
        self._n_seats = env_bldr.N_SEATS
        self.dir_tree_vis_data = None
        self.root = None
        self._native_tree = None
        self._solver = None
        self._nodes = {}
        self._cache = {}
        self._staged = {}
        self._env_states = {}
        self._replay_env = None

    # ---- reference properties -----------------------------------------------------------------------------------------
    stack_size = property(lambda s: s._stack_size)
    is_debugging = property(lambda s: s._is_debugging)
    n_seats = property(lambda s: s._n_seats)
    stop_at_street = property(lambda s: s._stop_at_street)
    put_out_new_round_after_limit = property(lambda s: s._put_out_new_round_after_limit)
    env_bldr = property(lambda s: s._env_bldr)

    @property
    def n_nodes(self):  # the reference's counter excludes the root (PublicTree.py:60,163)
        return self._native_tree.n_nodes - 1

    @property
    def n_nonterm(self):
        return int(np.sum(self._kind[1:] < KIND_FOLD))

    # ---- construction -------------------------------------------------------------------------------------------------
    def build_tree(self, variant="vanilla", delay=0):
        import copy
        self._variant, self._delay = variant, delay  # copy() builds its solver the same way
        env_cls, rules = self._env_bldr.env_cls, self._env_bldr.rules
        args = copy.deepcopy(self._env_bldr.env_args)
        args.starting_stack_sizes_list = copy.deepcopy(self._stack_size)
        boards = self._boards
        if boards is None:  # PublicTree.py:188-210: the chance outcomes come from the deck, cards ascending
            from pokerrl_amd.game import board_enum
            n_boards, max_outcomes, seed = self._board_caps
            boards, self._board_mult = board_enum.default_boards_or_classes(env_cls, n_boards=n_boards, max_outcomes=max_outcomes, seed=seed,
                                                                                suit_isomorphism=self._suit_iso)
            self._boards = boards
            if self._board_mult is not None:
                self._suit_iso = True  # the enumeration above produced every class of the game
        self._native_tree = _native.NativeTree(env_cls.native_game(args), env_cls.native_rules(), boards,
                                               stop_at_round=self._stop_at_street if self._is_partial else None)
        t = self._native_tree
        for f in ("kind", "actor", "parent", "action", "acted_last", "depth", "n_children", "first_col", "child_start", "child_list",
                  "col_action", "board_id", "main_pot", "round", "child_idx"):
            setattr(self, "_" + f, t.field(f))
        if self._is_partial:
            self._solver = None
        elif self._board_mult is not None:  # the whole game through its suit classes: fused engine, prl_solver_create_weighted
            self._solver = _native.NativeSolver(t, variant, delay, board_mult=self._board_mult, symmetrize=self._suit_iso or False)
        else:
            self._solver = _native.NativeSolver(t, variant, delay, engine=self._engine)
        self.root = self.node(0)
        self._invalidate()

    def node(self, idx):
        n = self._nodes.get(idx)
        if n is None:
            n = self._nodes[idx] = TreeNode(self, idx)
        return n

    def nodes(self):
        return (self.node(i) for i in range(self._native_tree.n_nodes))

    @property
    def solver(self):
        if self._solver is None:
            raise RuntimeError("partial tree (stop_at_street=%d): it has structure and states only; the solver needs the whole tree" % self._stop_at_street)
        return self._solver

    @property
    def native_tree(self):
        return self._native_tree

    # ---- passes (PublicTree.py:128-141) -------------------------------------------------------------------------------
    def compute_ev(self):
        self._flush()
        self.solver.compute_ev()
        self._invalidate()

    def fill_uniform_random(self):
        self._staged.clear()
        self.solver.fill_uniform()
        self._invalidate()

    def fill_random_random(self):
        """row-normalised np.random.random per decision node (StrategyFiller.py:67-86), then reach"""
        t = self._native_tree
        strat = np.zeros((t.n_cols, t.range_size), np.float64)
        for n in np.where(self._kind == KIND_DECISION)[0]:
            a = self._n_children[n]
            x = np.random.random(size=(t.range_size, a))
            x /= np.expand_dims(np.sum(x, axis=1), axis=-1)
            strat[self._first_col[n]:self._first_col[n] + a] = x.T
        self._staged.clear()
        self.solver.set_strategy(strat)
        self._invalidate()

    def fill_with_agent_policy(self, agent):
        """one query per decision node (StrategyFiller.py:88-116): strategy = agent probs restricted to the legal actions.

        An agent that defines ``get_a_probs_for_each_hand_in_nodes_device(nodes)`` -> a float32 device tensor ``[len(nodes), RANGE_SIZE, N_ACTIONS]``
        (or None to decline) keeps the probabilities in HBM: no host copy of the strategy at all (prl_solver_set_strategy_device).
        SURVEY section 8f-1 (batched agent querying): an agent that defines
        ``get_a_probs_for_each_hand_in_nodes(nodes) -> [len(nodes), RANGE_SIZE, N_ACTIONS]`` is asked ONCE for all decision
        nodes (DFS pre-order, the order the reference visits them in) instead of once per node, so a neural agent can run
        one batched forward; it positions its own env copies from ``node.env_state`` / the node's action history."""
        t = self._native_tree
        decision = np.where(self._kind == KIND_DECISION)[0]
        on_device = getattr(agent, "get_a_probs_for_each_hand_in_nodes_device", None)
        if callable(on_device):
            # round 6: the probabilities never leave HBM -- a float32 tensor [n_decision_nodes, R, N_ACTIONS] on the GPU (the network's output) is
            # scattered into the solver's columns by the library (prl_solver_set_strategy_device); equal to the host path below bit for bit
            probs = on_device([self.node(int(n)) for n in decision])
            if probs is not None:
                assert tuple(probs.shape) == (len(decision), t.range_size, int(self._env_bldr.N_ACTIONS)) and probs.is_contiguous(), tuple(probs.shape)
                self._staged.clear()
                self.solver.set_strategy_device(probs.data_ptr(), int(self._env_bldr.N_ACTIONS))
                self._invalidate()
                return
        strat, dtype = np.zeros((t.n_cols, t.range_size), np.float64), None
        batched = getattr(agent, "get_a_probs_for_each_hand_in_nodes", None)
        if callable(batched):
            nodes = [self.node(int(n)) for n in decision]
            probs = np.asarray(batched(nodes))
            assert probs.shape[:2] == (len(nodes), t.range_size), probs.shape
            for n, node, pr in zip(decision, nodes, probs):
                strat[self._first_col[n]:self._first_col[n] + self._n_children[n]] = pr[:, node.allowed_actions].T
            self._staged.clear()
            self.solver.set_strategy(strat if probs.dtype == np.float64 else strat.astype(np.float32))
            self._invalidate()
            return
        for n in decision:
            node = self.node(int(n))
            agent.set_to_public_tree_node_state(node=node)
            assert node.p_id_acting_next == agent._internal_env_wrapper.env.current_player.seat_id, node.p_id_acting_next
            probs = np.asarray(agent.get_a_probs_for_each_hand())
            dtype = probs.dtype if dtype is None else np.promote_types(dtype, probs.dtype)
            sel = probs[:, node.allowed_actions]
            strat[self._first_col[n]:self._first_col[n] + self._n_children[n]] = sel.T
        self._staged.clear()
        self.solver.set_strategy(strat if dtype == np.float64 else strat.astype(np.float32))
        self._invalidate()

    def update_reach_probs(self):
        self._flush()
        self.solver.update_reach()
        self._invalidate()

    def copy(self):
        n_boards, max_outcomes, board_seed = self._board_caps  # (a copy made before build_tree deals the same capped / seeded boards)
        c = PublicTree(self._env_bldr, self._stack_size, self._stop_at_street if self._is_partial else None,
                       self._put_out_new_round_after_limit, self._is_debugging, self._boards, self._engine,
                       n_boards=n_boards, max_outcomes=max_outcomes, board_seed=board_seed, suit_isomorphism=self._suit_iso, board_mult=self._board_mult)
        c.build_tree(variant=getattr(self, "_variant", "vanilla"), delay=getattr(self, "_delay", 0))
        if self._is_partial:  # structure and states only: there is no solver state to move over
            return c
        self._flush()
        try:  # a solver in a CFR run: the whole persistent state (regrets, averages, iteration counter) moves over
            c._solver.load_state(self.solver.save_state())
        except _native.NativeError as e:
            if e.status != _native.ERR_STATE:  # ERR_STATE = an explicit strategy is loaded (fill_with_agent_policy ...): copy that
                raise
            strat = self._vec("strategy")
            if self.solver.engine == "fused":  # ONE explicit strategy array whose dtype the solver remembers (no per-node flags)
                f64 = int(self.solver.get("explicit_strategy")[0]) == 1
            else:
                f64 = bool(self.solver.get("strat_f64").any())
            c._solver.set_strategy(strat if f64 else strat.astype(np.float32))
        c._invalidate()
        return c

    def get_tree_as_dict(self):
        """PublicTree.py:143-144: the nested PokerViz dictionary (pokerrl_amd/game/_tree_export.py)"""
        from pokerrl_amd.game import _tree_export
        self._flush()
        return _tree_export.tree_as_dict(self)

    def export_to_file(self, name="data"):
        """PublicTree.py:146-149: writes <dir_tree_vis_data>/<name>.js ("const data=" + JSON) if a directory is set"""
        if self.dir_tree_vis_data is not None:
            from pokerrl_amd.game import _tree_export
            return _tree_export.write_js(self.dir_tree_vis_data, name, self.get_tree_as_dict())
        return None

    # ---- internals ----------------------------------------------------------------------------------------------------
    def _invalidate(self):
        self._cache = {}

    def _vec(self, name):
        if name not in self._cache:
            self._flush()
            self._cache[name] = self.solver.get(name)
        return self._cache[name]

    def _stage_strategy(self, idx, value):
        assert self._kind[idx] == KIND_DECISION and value.shape == (self._native_tree.range_size, self._n_children[idx])
        self._staged[idx] = value
        self._cache.pop("strategy", None)

    def _flush(self):
        if not self._staged:
            return
        staged, self._staged = self._staged, {}
        cur = self.solver.get("strategy")
        flags = np.array(self.solver.get("strat_f64"), dtype=np.uint8)
        for idx, v in staged.items():  # a node keeps the dtype of the array assigned to it, like the reference's node.strategy
            cur[self._first_col[idx]:self._first_col[idx] + self._n_children[idx]] = v.T
            flags[idx] = 1 if v.dtype == np.float64 else 0
        if self.solver.engine == "fused":
            f64 = bool(flags.any())
            self.solver.set_strategy(cur if f64 else cur.astype(np.float32))
        elif flags.any():
            self.solver.set_strategy_mixed(cur, flags)
        else:
            self.solver.set_strategy(cur.astype(np.float32))
        self._invalidate()

    def _chance_strategy(self, idx):
        """[R, n_boards] float32: board probability for hands not blocked by the board (StrategyFiller.py:148-169)"""
        t, lut = self._native_tree, self._env_bldr.lut_holder.LUT_IDX_2_HOLE_CARDS
        kids = self.node(idx).children
        before = t.board_rows[self._board_id[idx]] if self._board_id[idx] >= 0 else np.zeros(0, np.int8)
        n_before = int(np.sum(before >= 0))
        rows = [t.board_rows[self._board_id[c._i]] for c in kids]
        k = int(np.sum(rows[0] >= 0)) - n_before
        from math import comb
        n = self._env_bldr.rules.N_CARDS_IN_DECK - n_before
        h = self._env_bldr.rules.N_HOLE_CARDS
        p = np.float32(1.0 / (float(len(kids)) * float(comb(n - 2 * h, k)) / float(comb(n, k))))  # generalised StrategyFiller.py:166
        out = np.zeros((t.range_size, len(kids)), np.float32)
        for b, row in enumerate(rows):
            blocked = np.isin(lut, row[row >= 0]).any(axis=1)
            out[~blocked, b] = p
        return out

    def _get_replay_env(self):
        if self._replay_env is None:
            self._replay_env = self._env_bldr.get_new_env(is_evaluating=True, stack_size=self._stack_size)
            a = self._replay_env.get_args()
            a.RETURN_PRE_TRANSITION_STATE_IN_INFO = True
            self._replay_env.set_args(a)
        return self._replay_env

    def _env_state_of(self, idx):
        """public env state of a node (PokerEnv.state_dict layout) with the reference's conventions (PublicTree.py:205-293):
        decision nodes hold the env's state after the action; terminal and chance-pending nodes hold the state after the action
        but BEFORE the money moves (bets not swept, nothing paid, parent's round / board / deck); a chance outcome holds the
        post-transition state of its parent with this node's board on the table and those cards out of the deck."""
        if idx in self._env_states:
            return self._env_states[idx]
        import copy
        env = self._get_replay_env()
        if idx == 0:
            env.reset()
            st = copy.deepcopy(env.state_dict())
        elif int(self._action[idx]) == -1:
            pending = int(self._parent[idx])
            before = self._env_state_of(int(self._parent[pending]))
            env.load_state_dict(copy.deepcopy(before))
            env.step(int(self._action[pending]))  # sweeps the bets and opens the next round (its random deal is replaced below)
            board_1d = np.asarray(self._native_tree.board_rows[self._board_id[idx]])
            board_1d = board_1d[board_1d >= 0]  # a row of the board table is a prefix: the cards dealt so far
            env.board[:] = before[EnvDictIdxs.board_2d]
            env.board[:len(board_1d)] = self._env_bldr.lut_holder.get_2d_cards(board_1d)
            env.deck.load_state_dict(copy.deepcopy(before[EnvDictIdxs.deck]))
            env.deck.remove_cards(env.board[:len(board_1d)])
            st = copy.deepcopy(env.state_dict())
        else:
            before = self._env_state_of(int(self._parent[idx]))
            env.load_state_dict(copy.deepcopy(before))
            _o, _r, _done, info = env.step(int(self._action[idx]))
            if self._kind[idx] == KIND_DECISION:
                st = copy.deepcopy(env.state_dict())
            elif self._kind[idx] == KIND_CHANCE:
                # PokerEnv.py:761-766: the state after the (check / call) action and before _next_round -- bets still in front
                # of the players, old pot, old round, the actor still "current": what the env hands out in its step info
                st = {k: v for k, v in copy.deepcopy(info["state_dict_before_money_move"]).items() if not (isinstance(k, str) and k.startswith("_"))}
            else:
                st = {k: v for k, v in copy.deepcopy(info["state_dict_before_money_move"]).items() if not (isinstance(k, str) and k.startswith("_"))}
                st[EnvDictIdxs.current_round] = before[EnvDictIdxs.current_round]
                st[EnvDictIdxs.board_2d] = np.copy(before[EnvDictIdxs.board_2d])
                st[EnvDictIdxs.deck] = copy.deepcopy(before[EnvDictIdxs.deck])
        self._env_states[idx] = st
        return st
