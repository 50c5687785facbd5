"""The synthetic signal families of the test suite live in the package (flac_amd/signals.py: bench.py and flac_amd/corpus.py
generate their workloads from them); the tests import them under the old name."""
from flac_amd.signals import *          # noqa: F401,F403
from flac_amd.signals import FAMILIES, _clip      # noqa: F401
