import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Every GPU test gets a deadline of its own (pytest-timeout is in the image): a hung kernel or a wedged box fails ONE test with a
    traceback instead of eating the whole run (round 3 saw a box, fresh from seven rocprof PMC passes, hang in its first test for 600 s)."""
    try:
        import pytest_timeout  # noqa: F401
    except Exception:
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """On a GPU box bring torch's HIP runtime up before libpr_amd.so loads its own copy (the documented order:
    torch owns device memory and streams, the library is loaded next to it)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


def _ref_sequence_names():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sequences")
    return sorted(n for n in os.listdir(d) if os.path.isdir(os.path.join(d, n))) if os.path.isdir(d) else []


REF_SEQUENCES = _ref_sequence_names()


@pytest.fixture(scope="session")
def ref_sequence(tmp_path_factory):
    """name -> (poses_history_file.txt, incoming_id_file.txt) of one of the 13 sequences the reference holds
    (tests/golden/ref_sequences, written by tests/make_ref_sequences.py; the pose files are stored gzip-compressed)."""
    import gzip
    import shutil
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sequences")
    cache = {}

    def get(name):
        if name not in cache:
            d = tmp_path_factory.mktemp("ref_" + name.replace("-", "_"))
            poses = str(d / "poses_history_file.txt")
            with gzip.open(os.path.join(root, name, "poses_history_file.txt.gz"), "rb") as f, open(poses, "wb") as g:
                shutil.copyfileobj(f, g)
            cache[name] = (poses, os.path.join(root, name, "incoming_id_file.txt"))
        return cache[name]
    return get
