import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if HERE not in sys.path:
    sys.path.insert(0, HERE)

DATA = os.path.join(HERE, "golden", "data")
# the kernel's tuning / test knobs (PFAC_FORCE_L2, PFAC_FAULT, ...) are honoured only by a process that opts in
os.environ.setdefault("PFAC_ENABLE_KNOBS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # product libs + oracle must exist (both are built in-tree by __graft_entry__.build())
    need = [os.path.join(REPO, "phfpfac_amd", "lib", "libpfac_host.so"),
            os.path.join(REPO, "phfpfac_amd", "lib", "libpfac_hip.so"),
            os.path.join(REPO, "phfpfac_amd", "lib", "libpfac_seam.so"),
            os.path.join(REPO, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=REPO)


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def work_dir(tmp_path_factory):
    """Materialised fixture files: the 1 MiB input `1M`, the concatenated dictionary, the gunzipped title list."""
    d = tmp_path_factory.mktemp("pfac")
    para = open(os.path.join(DATA, "paragraph402"), "rb").read()
    one_m = (para * (1048576 // 402 + 1))[:1048576]
    (d / "1M").write_bytes(one_m)
    with open(d / "all.pat", "wb") as f:
        for p in ("xaa", "xab", "xac", "xad"):
            f.write(open(os.path.join(DATA, p), "rb").read())
    with gzip.open(os.path.join(DATA, "bytefile_1000000byte.gz"), "rb") as g:
        (d / "bytefile_1000000byte").write_bytes(g.read())
    return d


@pytest.fixture(scope="session")
def resolve(work_dir):
    """Map the names used in golden/fingerprints.json to files."""
    def _r(name):
        m = {"1M": work_dir / "1M", "xaa+xab+xac+xad": work_dir / "all.pat",
             "bytefile/1000000byte": work_dir / "bytefile_1000000byte",
             "bytefile/10000byte": os.path.join(DATA, "bytefile_10000byte"),
             "bytefile/100000byte": os.path.join(DATA, "bytefile_100000byte")}
        return str(m.get(name, os.path.join(DATA, name)))
    return _r
