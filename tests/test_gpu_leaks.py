"""Device memory is given back: contexts created and destroyed, trees re-uploaded in every bounds mode,
queries, frames, and frames on more streams than there are scratch slots (tools/leak_probe.py) -- free device
memory must stop changing once the runtime's own pools are warm."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_no_device_memory_is_lost_across_context_lifetimes(gpu):
    import leak_probe

    marks, growth = leak_probe.measure(blocks=3, per_block=6)
    assert max(growth) < 2.0, (marks, growth)
