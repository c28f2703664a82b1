"""TEST INFRASTRUCTURE ONLY — CPU oracle for Tucker-factorised FNO weights (TFNO).  PARITY UNPINNED: tltorch /
tensorly (third-party, absent) define the reference behaviour; this restates their published algorithm
(SURVEY.md App. A-1): complex Tucker reconstruction and the float-rank rule of validate_tucker_rank."""
import numpy as np
import torch
from scipy.optimize import brentq


def tucker_rank(shape, rank):
    n = len(shape)
    target = np.prod(shape) * rank
    sq = np.sum([s ** 2 for s in shape])
    fun = lambda x: target - sq * x - np.prod(shape) * x ** n
    c = brentq(fun, 0.0, max(rank, 1.0))
    return [max(int(round(s * c)), 1) for s in shape]


def tucker_dense(core, factors):
    """core complex [a,b,c,d]; factors complex [dim_k, r_k] -> dense complex [i,o,x,y]"""
    return torch.einsum("abcd,ia,ob,xc,yd->ioxy", core, *factors)
