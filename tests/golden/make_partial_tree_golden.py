import os, sys
sys.path.insert(0, "/root/repo/tests/golden")
import ref_harness
np = ref_harness.setup()
import make_golden as mg
from PokerRL.game._.tree.PublicTree import PublicTree
out = {}
for name, (cls, kw) in mg.GAMES.items():
    if name not in ("StandardLeduc", "DiscretizedNLLeduc_POT"):
        continue
    for stop in (0, 1):
        bldr, args = mg.make_bldr(cls, kw["stack"], kw["bets"])
        tree = PublicTree(env_bldr=bldr, stack_size=args.starting_stack_sizes_list, stop_at_street=stop)
        tree.build_tree()
        # flatten tolerating leaves: is_terminal False with no children
        rec = dict(kind=[], actor=[], parent=[], round=[], n_children=[], depth=[])
        def visit(node, parent_id):
            my = len(rec["kind"])
            if node.is_terminal: kind = 2 if node.action == 0 else 3
            elif node.p_id_acting_next == tree.CHANCE_ID: kind = 1
            else: kind = 0
            rec["kind"].append(kind); rec["actor"].append(node.p_id_acting_next if kind == 0 else -1); rec["parent"].append(parent_id)
            rec["round"].append(int(node.env_state[mg.EnvDictIdxs.current_round])); rec["n_children"].append(len(node.children)); rec["depth"].append(node.depth)
            for c in node.children: visit(c, my)
        visit(tree.root, -1)
        for k, v in rec.items(): out["%s_stop%d_%s" % (name, stop, k)] = np.array(v, np.int32)
        print(name, stop, len(rec["kind"]), tree.n_nodes, tree.n_nonterm)
        out["%s_stop%d_counters" % (name, stop)] = np.array([tree.n_nodes, tree.n_nonterm], np.int32)
np.savez_compressed("/root/repo/tests/golden/tree_partial.npz", **out)
