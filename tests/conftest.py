import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kandinsky-5_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle on a many-core host: torch's default of one thread per logical CPU (256 on the GPU box) runs the oracle's
    # medium-size matmuls ~100x slower than 16 threads do (measured, bench.py cpu_baseline) — cap it for every test
    import torch
    if torch.get_num_threads() > 32:
        torch.set_num_threads(32)


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file
    return load_file(os.path.join(GOLDEN, "dit_tiny.safetensors"))


@pytest.fixture(scope="session")
def golden_meta():
    import json
    with open(os.path.join(GOLDEN, "dit_tiny_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def tiny_sd(golden):
    return {k[2:]: v for k, v in golden.items() if k.startswith("w.")}
