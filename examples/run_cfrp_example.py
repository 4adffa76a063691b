"""150 iterations of CFR+ on DiscretizedNLLeduc (reference: examples/run_cfrp_example.py, BASELINE.json config 1)."""
from _common import run

from pokerrl_amd.cfr.CFRPlus import CFRPlus

if __name__ == "__main__":
    run(CFRPlus, "CFRp_EXAMPLE", delay=0)
