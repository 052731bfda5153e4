"""CPU: run the real kernel sources through the fiber interpreter and compare with plain PyTorch."""
import pytest
import torch

import emu
import kernel_cases as KC


@pytest.fixture(scope="module", autouse=True)
def _emu():
    if torch.cuda.is_available():
        pytest.skip("GPU present: the -m gpu suite exercises the real library instead")
    emu.install()


@pytest.mark.parametrize("case", KC.CONV_CASES, ids=[c[0] for c in KC.CONV_CASES])
def test_conv(case):
    KC.run_conv_case(case, "cpu")
