#!/usr/bin/env python
"""imagenet_ddp_apex.py — same command line as the reference's imagenet_ddp_apex.py, running on the MI355X-native engine.

See distributed_pytorch_amd/cli.py for the flag table (every reference flag / alias / default is
kept) and distributed_pytorch_amd/engine.py for the trainer.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_pytorch_amd import cli, engine  # noqa: E402

if __name__ == "__main__":
    engine.main(cli.parse("apex"))
