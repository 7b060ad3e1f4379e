"""Helpers for the multi-process tests."""


def join_or_kill(procs, timeout):
    """join every child; a child that outlives the timeout is terminated (then killed) BEFORE the assert, so a stalled
    rank never stays on the GPU behind a failed test."""
    import time
    deadline = time.time() + timeout
    for p in procs:
        p.join(max(0.0, deadline - time.time()))
    stuck = [p for p in procs if p.is_alive()]
    for p in stuck:
        p.terminate()
    for p in stuck:
        p.join(10)
        if p.is_alive():
            p.kill()
            p.join(10)
    assert not stuck, f"{len(stuck)} rank(s) still running after {timeout} s (terminated)"
    for p in procs:
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
