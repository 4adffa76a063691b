from pokerrl_amd.eval.head_to_head.BatchedHead2Head import BatchedHead2Head, BatchedHead2HeadMaster
from pokerrl_amd.eval.head_to_head.H2HArgs import H2HArgs
from pokerrl_amd.eval.head_to_head.LocalHead2HeadMaster import LocalHead2HeadMaster

__all__ = ["BatchedHead2Head", "BatchedHead2HeadMaster", "H2HArgs", "LocalHead2HeadMaster"]
