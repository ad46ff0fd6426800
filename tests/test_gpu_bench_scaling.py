"""GPU: the db_build_scaling leg of bench.py -- the code the driver runs with --gpus N -- at a small size: one rank, and the
N > 1 path (communicator warm-up, reservation for gathering, ShardedBuilder sending runs on the way, max over ranks) with
thread ranks on this GPU.  Same corpus, so the same table whatever the number of ranks."""
import importlib.util
import os
import types

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scaling_leg_one_rank_and_thread_ranks_build_the_same_table():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from shazam_amd import _ffi
    bs = _load(os.path.join(ROOT, "scripts", "build_scaling.py"), "build_scaling_script")
    songs, seconds = 600, 60.0
    ctx = _ffi.Context(0)
    one = bench.db_build_scaling(types.SimpleNamespace(scaling_songs=songs, scaling_seconds=seconds), ctx, None, None, 0, 1)
    ctx.close()
    assert one["n_gpus"] == 1 and one["rows"] > 0 and one["seconds"] > 0 and one["config"].startswith("BASELINE configs[2]")
    for world in (2, 4):
        outs = bs.run_local(songs, seconds, world)
        assert [o["rows"] for o in outs] == [one["rows"]] * world           # every rank holds the whole table
        assert all(o["n_gpus"] == world and o["allgather_bytes_received"] > 0 for o in outs)
        assert all(o["runs_sent_on_the_way_rank0"] >= 1 for o in outs[:1])  # runs travelled before the final call
