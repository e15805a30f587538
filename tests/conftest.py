import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # before anything in this process touches the GPU: the hardware queues a four-handle pool wants (asr_amd/hwqueues.py),
    # so that the GPU tests run the product's default (Model: four engine handles per GPU)
    from asr_amd import hwqueues
    hwqueues.ensure()


@pytest.fixture(scope='session')
def repo_root():
    return ROOT
