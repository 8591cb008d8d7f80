from .amp_gcn import AMPGCN, FeatureTokens

__all__ = ['AMPGCN', 'FeatureTokens']
