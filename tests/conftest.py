import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the product library and the oracle once per session (hipcc cross-compiles without a GPU)."""
    from eirgrid_amd import build as b
    b.build()
    from oracle import api as O
    O.build()
    return True


@pytest.fixture(scope="session")
def world(built):
    from eirgrid_amd import synthetic_world
    return synthetic_world()


@pytest.fixture(scope="session")
def oracle_world(world):
    from oracle import api as O
    return O.OracleWorld(world)


@pytest.fixture(scope="session")
def engine(world):
    from eirgrid_amd.engine import Engine
    eng = Engine(world, device=0)
    yield eng
    eng.close()
