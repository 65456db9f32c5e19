"""The allocation-free entry point is for callers under HIP graph capture (sdfgpu.h): one
sdfgpu_nearest_dev_static call -- curve keys, radix sort, walk -- is captured into a graph (capture mode
"global": an allocation or a synchronisation inside it would abort the capture), replayed on fresh inputs and
compared with sdfgpu_nearest_dev bit for bit (tools/graph_probe.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_static_entry_point_replays_from_a_graph(gpu):
    import graph_probe

    msgs = []
    assert graph_probe.run(log=msgs.append), msgs


def test_a_frame_replays_from_a_graph(gpu):
    # sdfgpu_render_dev allocates its per-stream scratch on the first call on a stream; after that warm-up the
    # whole frame (tile schedule, eye walk, march) can be captured and replayed (tools/graph_frame_probe.py)
    import graph_frame_probe

    msgs = []
    assert graph_frame_probe.run(log=msgs.append, config=2, frames=5), msgs
