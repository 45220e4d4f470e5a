"""factors.utils — factor classification used by the clique sampler (reference: src/factors/utils.py:8-28)."""
from typing import List

from factors.Factors import AmbiguousDataAssociationFactor, BinaryFactor, BinaryFactorWithNullHypo, PriorFactor


def classify_factors(factors: List, ranked_classes: List):
    """Split `factors` by the first class of `ranked_classes` each one is an instance of."""
    groups = [[] for _ in ranked_classes]
    for f in factors:
        for g, klass in zip(groups, ranked_classes):
            if isinstance(f, klass):
                g.append(f)
                break
        else:
            raise ValueError("Unknown factor classes: " + str(f))
    return groups


def unpack_prior_binary_nh_da_factors(factors: List):
    pr, nh, da, bf = classify_factors(factors, [PriorFactor, BinaryFactorWithNullHypo,
                                                AmbiguousDataAssociationFactor, BinaryFactor])
    return pr, bf, nh, da
