"""Local Best Response (arXiv:1612.07547) -- mirrors PokerRL/eval/lbr: LBRArgs, LocalLBRMaster, LocalLBRWorker."""
from pokerrl_amd.eval.lbr.BatchedLBR import BatchedLBR, BatchedLBRWorker
from pokerrl_amd.eval.lbr.LBRArgs import LBRArgs
from pokerrl_amd.eval.lbr.LocalLBRMaster import LocalLBRMaster
from pokerrl_amd.eval.lbr.LocalLBRWorker import LocalLBRWorker

__all__ = ["BatchedLBR", "BatchedLBRWorker", "LBRArgs", "LocalLBRMaster", "LocalLBRWorker"]
