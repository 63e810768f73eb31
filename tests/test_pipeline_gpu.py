"""-m gpu: the whole accelerated path in service order (tests/pipeline_demo.py): leaves -> tree -> proofs -> CEX / batch
commitments -> compressed key -> resident R1CS -> proof -> pairing verification, every step checked against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.isolated
def test_pipeline_demo():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_demo.py")
    spec = importlib.util.spec_from_file_location("pipeline_demo", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(n_acc=3000, verbose=False)
