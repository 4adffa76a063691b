"""
Generates tests/golden/wrappers.npz by RUNNING THE REFERENCE (/root/reference): observation wrappers of PokerRL/game/wrappers.py
  <case>_attrs         builder attributes [pub_obs_size, priv_obs_size, complete_obs_size, obs_size_board, obs_size_player_info_each,
                       obs_size_table_state, action_vector_size or 0]
  <case>_node_obs      for every decision node of the reference's PublicTree (DFS pre-order): the wrapped observation after
                       wrapper.set_to_public_tree_node_state(node), rows concatenated; <case>_node_off = row offsets
  <case>_play_obs/_off the wrapped observation after every step of seeded random play (reset included), same packing;
                       <case>_play_act = the actions
Cases: History (RecurrentHistoryWrapper, both history orders) and Flat (FlatHULimitPokerHistoryWrapper) on StandardLeduc,
History on DiscretizedNLLeduc.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.game import bet_sets  # noqa: E402
from PokerRL.game.games import DiscretizedNLLeduc, StandardLeduc  # noqa: E402
from PokerRL.game.wrappers import FlatLimitPokerEnvBuilder, HistoryEnvBuilder, VanillaEnvBuilder  # noqa: E402

CASES = {
    "History_StandardLeduc": (HistoryEnvBuilder, {}, StandardLeduc, 13, None),
    "HistoryInv_StandardLeduc": (HistoryEnvBuilder, {"invert_history_order": True}, StandardLeduc, 13, None),
    "Flat_StandardLeduc": (FlatLimitPokerEnvBuilder, {}, StandardLeduc, 13, None),
    "History_DiscretizedNLLeduc": (HistoryEnvBuilder, {}, DiscretizedNLLeduc, 1500, bet_sets.B_3),
    "Vanilla_StandardLeduc": (VanillaEnvBuilder, {}, StandardLeduc, 13, None),
}


def pack(list_of_arrays):
    rows = [np.atleast_2d(np.asarray(a, np.float32)) for a in list_of_arrays]
    off = np.cumsum([0] + [r.shape[0] for r in rows])
    return np.concatenate(rows, axis=0), off.astype(np.int64)


def main():
    from PokerRL.game._.tree.PublicTree import PublicTree
    out = {}
    for name, (bcls, kw, game, stack, bets) in CASES.items():
        args = game.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bets) if bets is not None \
            else game.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack])
        bldr = bcls(env_cls=game, env_args=args, **kw)
        out[name + "_attrs"] = np.array([bldr.pub_obs_size, bldr.priv_obs_size, bldr.complete_obs_size, bldr.obs_size_board,
                                         bldr.obs_size_player_info_each, bldr.obs_size_table_state, getattr(bldr, "action_vector_size", 0)], np.int64)
        if bcls is not VanillaEnvBuilder:  # the reference's Vanilla wrapper does not position its env on a node (Vanilla.py:38-39)
            tree = PublicTree(env_bldr=bldr, stack_size=args.starting_stack_sizes_list, stop_at_street=None)
            tree.build_tree()
            w = bldr.get_new_wrapper(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
            obs = []

            def visit(n):
                if (not n.is_terminal) and n.p_id_acting_next != tree.CHANCE_ID:
                    w.set_to_public_tree_node_state(n)
                    obs.append(np.copy(w.get_current_obs()))
                for c in n.children:
                    visit(c)

            visit(tree.root)
            out[name + "_node_obs"], out[name + "_node_off"] = pack(obs)
        w = bldr.get_new_wrapper(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
        rng = np.random.RandomState(5)
        np.random.seed(5)
        play, acts = [], []
        for ep in range(25):
            o, _, done, _ = w.reset()
            play.append(np.copy(o)); acts.append(-1)
            while not done:
                legal = w.env.get_legal_actions()
                a = int(legal[rng.randint(len(legal))])
                o, _, done, _ = w.step(a)
                play.append(np.copy(o)); acts.append(a)
        out[name + "_play_obs"], out[name + "_play_off"] = pack(play)
        out[name + "_play_act"] = np.array(acts, np.int64)
        print(name, out[name + "_attrs"].tolist(), out.get(name + "_node_obs", np.zeros(0)).shape, out[name + "_play_obs"].shape)
    np.savez_compressed(os.path.join(HERE, "wrappers.npz"), numpy=np.__version__, **out)


if __name__ == "__main__":
    main()
