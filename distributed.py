#!/usr/bin/env python3
"""Drop-in counterpart of the reference launcher: same command lines as
/root/reference/README.md:7-15, e.g.

    python distributed.py --job_name=ps --task_index=0
    python distributed.py --job_name=worker --task_index=0
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from distributed_tensorflow_amd.train import run  # noqa: E402

if __name__ == '__main__':
    run()
