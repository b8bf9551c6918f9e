import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


BENCH_PATH_TEST = "test_bench_path_at_320x512_T16_matches_oracle_on_every_step"
FULL_BATCH_TEST = "test_full_batch_train_forward_at_bs32_320x512_matches_the_oracle"
RL_EVAL_TEST = "test_eval_mode_backward_matches_oracle"


@pytest.hookimpl(trylast=True)
def pytest_collection_finish(session):
    """The bench-path parity test needs ~4-5 minutes of CPU oracle work (fp64 + fp32 at 320x512, T = 16) and seconds of GPU work: when it
    is part of a GPU session, its oracle runs are started NOW in two worker processes and the test itself is moved to the end of the
    session, so that the host work overlaps with the other GPU tests (the suite at 751 s of the driver's 1200 s)."""
    items = session.items
    live = lambda nm: [it for it in items if it.name == nm and not any(m.name == "skip" for m in it.iter_markers())]
    late, full, rl = live(BENCH_PATH_TEST), live(FULL_BATCH_TEST), live(RL_EVAL_TEST)
    if not (late or full or rl) or session.config.option.collectonly:
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    # the full-batch leg's oracle (fp32 on 32 samples, fp64 encoder + 4 decoded samples: 2-4 minutes of host work) runs beside the
    # bench-path one; its test comes second to last
    # (the RL eval-backward test's oracle, ~35 s of host work, likewise: started now, the test after the other cheap ones)
    # (starting the worker processes of test_ddp_gpu.py here as well was tried: nine more processes on the box at session start made the
    # first tests five times slower and the session no shorter)
    moved = late + full + rl
    items[:] = [it for it in items if it not in moved] + rl + full + late
    if len(items) > 1:                         # (a run of one of these tests alone starts its workers itself)
        from helpers import start_bench_oracle, start_full_oracle, start_rl_eval_oracle
        if late:
            session.config._bench_oracle = start_bench_oracle(background=True)
        if full:
            session.config._full_oracle = start_full_oracle(background=True)
        if rl:
            session.config._rl_oracle = start_rl_eval_oracle(background=True)


def pytest_sessionfinish(session, exitstatus):
    for attr in ("_bench_oracle", "_full_oracle", "_rl_oracle"):
        bo = getattr(session.config, attr, None)
        if bo is not None:
            bo[0].shutdown(wait=False, cancel_futures=True)
