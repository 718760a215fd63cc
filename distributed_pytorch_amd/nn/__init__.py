from .modules import (AdaptiveAvgPool2d, AvgPool2d, BatchNorm2d, Conv2d, CrossEntropyLoss, Dropout, Flatten,
                      Linear, MaxPool2d, ReLU, ReLU6, SyncBatchNorm, conv_bn_act, conv_bn_relu_maxpool)

__all__ = ["AdaptiveAvgPool2d", "AvgPool2d", "BatchNorm2d", "Conv2d", "CrossEntropyLoss", "Dropout", "Flatten",
           "Linear", "MaxPool2d", "ReLU", "ReLU6", "SyncBatchNorm", "conv_bn_act", "conv_bn_relu_maxpool"]
