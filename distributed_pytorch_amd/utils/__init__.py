from .meters import (AverageMeter, DeviceMeter, PlainMeter, ProgressMeter, Summary, accuracy,
                     format_hms)

__all__ = ["AverageMeter", "DeviceMeter", "PlainMeter", "ProgressMeter", "Summary", "accuracy",
           "format_hms"]
