#!/usr/bin/env python
# coding=utf-8
"""Drop-in for the reference's code/train.py on the MI355X engine
(same command line; see multiverse_amd/cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from multiverse_amd import cli  # noqa: E402

if __name__ == "__main__":
  cli.train_main()
