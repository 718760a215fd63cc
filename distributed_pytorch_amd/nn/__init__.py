from .modules import (AdaptiveAvgPool2d, AvgPool2d, BatchNorm2d, Conv2d, CrossEntropyLoss, Dropout, Flatten, GELU,
                      Hardsigmoid, Hardswish, LayerNorm, LayerNorm2d, Linear, MaxPool2d, ReLU, ReLU6, Sigmoid, SiLU, SqueezeExcitation,
                      StochasticDepth, SyncBatchNorm, conv_bn_act, conv_bn_relu_maxpool)

__all__ = ["AdaptiveAvgPool2d", "AvgPool2d", "BatchNorm2d", "Conv2d", "CrossEntropyLoss", "Dropout", "Flatten", "GELU",
           "Hardsigmoid", "Hardswish", "LayerNorm", "LayerNorm2d", "Linear", "MaxPool2d", "ReLU", "ReLU6", "Sigmoid", "SiLU", "SqueezeExcitation",
           "StochasticDepth", "SyncBatchNorm", "conv_bn_act", "conv_bn_relu_maxpool"]
