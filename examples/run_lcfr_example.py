"""150 iterations of Linear CFR on DiscretizedNLLeduc (reference: examples/run_lcfr_example.py)."""
from _common import run

from pokerrl_amd.cfr.LinearCFR import LinearCFR

if __name__ == "__main__":
    run(LinearCFR, "LCFR_EXAMPLE")
