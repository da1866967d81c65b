"""Host-side mirror of the reference's ``src.models`` package (same public names)."""
from .cab import ContextAggregationBlock, DWConv, GlobalContextAttention, LocalAttention, PSPModule
from .cabinet import (AttentionBranch, CABiNet, CABiNetOutput, ConvBNReLU, FeatureFusionModule,
                      SpatialBranch)
from .constants import MODEL_CONFIG, MOBILENETV3_CFGS
from .mobilenetv3 import MobileNetV3

__all__ = ["ContextAggregationBlock", "DWConv", "GlobalContextAttention", "LocalAttention", "PSPModule",
           "AttentionBranch", "CABiNet", "CABiNetOutput", "ConvBNReLU", "FeatureFusionModule", "SpatialBranch",
           "MODEL_CONFIG", "MOBILENETV3_CFGS", "MobileNetV3"]
