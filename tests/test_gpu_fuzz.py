"""A bounded slice of the randomised differential run (tools/tools_fuzz.py): every
product of the build and every search form against the oracle on texts of six
kinds, sizes 1 .. 200,000."""
import pytest

pytestmark = pytest.mark.gpu


def test_randomised_texts_against_the_oracle(gpu):
    import os, sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tools_fuzz

    cases = tools_fuzz.main(budget=20.0, seed0=7000, nmax=200000, max_cases=250, quiet=True)
    assert cases >= 50
