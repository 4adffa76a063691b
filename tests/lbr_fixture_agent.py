"""The fixture EvalAgent of the LBR / head-to-head tests = the host mirror of the library's synthetic hash agent; it lives in the
package (pokerrl_amd/rl/hash_agent.py) because bench_lbr.py / bench_h2h.py time the host drop-ins with it. Re-exported here for
the tests and the golden generators (which bind it to the REFERENCE's EvalAgentBase)."""
from pokerrl_amd.rl.hash_agent import M32, make_agent_cls, mix32, policy, state_key  # noqa: F401
