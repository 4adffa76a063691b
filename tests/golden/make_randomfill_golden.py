"""Golden vector for PublicTree.fill_random_random: the REFERENCE's tree, np.random.seed(7), random fill,
compute_ev -> every decision node's float64 strategy (DFS pre-order, one [R, A] block per node, hashed and sampled), the root's
exploitability / ev / ev_br and a few nodes' reach. Usage: python tests/golden/make_randomfill_golden.py -> tests/golden/randomfill.npz"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.game._.tree.PublicTree import PublicTree  # noqa: E402
from PokerRL.game.games import StandardLeduc  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402

args = StandardLeduc.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[13, 13])
tree = PublicTree(env_bldr=HistoryEnvBuilder(env_cls=StandardLeduc, env_args=args), stack_size=[13, 13], stop_at_street=None)
tree.build_tree()
np.random.seed(7)
tree.fill_random_random()
tree.compute_ev()


def walk(n, out):
    out.append(n)
    for c in n.children:
        walk(c, out)
    return out


nodes = walk(tree.root, [])
dec = [n for n in nodes if not n.is_terminal and n.p_id_acting_next != tree.CHANCE_ID]
h = hashlib.sha256()
for n in dec:
    assert n.strategy.dtype == np.float64
    h.update(np.ascontiguousarray(n.strategy).tobytes())
np.savez_compressed(os.path.join(HERE, "randomfill.npz"), n_decision=np.int64(len(dec)), sha256=np.array(h.hexdigest()),
                    first=np.array(dec[0].strategy), last=np.array(dec[-1].strategy), exploitability=np.array(tree.root.exploitability),
                    root_ev=np.array(tree.root.ev), root_ev_br=np.array(tree.root.ev_br), reach_25=np.array(nodes[25].reach_probs),
                    ev_25=np.array(nodes[25].ev))
print(len(dec), h.hexdigest()[:16], tree.root.exploitability, tree.root.ev.dtype)
