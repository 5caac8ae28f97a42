"""The constant extractor of the frozen physRNN exports lives in the package (climsim_amd/frozen_extract.py: it is what
`physical_RNN_wrapped.from_export` uses); the golden scripts import it from here."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from climsim_amd.frozen_extract import *          # noqa: F401,F403,E402
from climsim_amd.frozen_extract import Extractor  # noqa: F401,E402
