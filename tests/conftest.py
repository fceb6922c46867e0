import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """The oracle is test infrastructure: make sure its C restatement is compiled."""
    import oraclelib
    if not os.path.exists(oraclelib.ORACLE_SO):
        oraclelib.build_oracle()


# Es/N0 (dB) at which each mode is exercised: FER<0.1 threshold of include/common/common_defines.h:130-147
# plus 2 dB; the two zero-forcing modes need more in the baseband loop.
OPERATING_ESN0 = {0: -8.0, 1: -6.0, 2: -4.5, 3: -3.0, 4: -1.5, 5: -0.5, 6: 1.0, 7: 1.5, 8: 2.5, 9: 4.0,
                  10: 5.5, 11: 7.0, 12: 8.5, 13: 9.5, 14: 11.5, 15: 16.0, 16: 20.0,
                  # ROBUST_0..2 (MFSK): waterfalls -13 / -11 / -8 dB (telecom_system.cc:2970-2972) plus ~3 dB
                  100: -10.0, 101: -8.0, 102: -5.0}
MFSK_CFGS = (100, 101, 102)
SEED = 0x4D455243

# Explicit (constellation, LDPC rate, preamble, estimator) combinations outside load_configuration's 17 rows
# (include/mercury_gpu.h MGPU_CFG_EXPLICIT), with an Es/N0 at which they decode: (M, rate16, preamble_nsymb, estimator, Es/N0 dB)
EXPLICIT_COMBOS = [(4, 1, 4, 1, -5.0), (16, 5, 2, 1, 8.0), (32, 8, 1, 0, 20.0), (8, 3, 3, 1, 3.0), (2, 14, 2, 0, 12.0), (4, 4, 8, 1, 1.0)]
