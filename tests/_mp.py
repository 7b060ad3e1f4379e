"""Helpers shared by the test modules: child-process joins and THE float bar."""


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


def dtype_factor() -> float:
    """Every float bar in the GPU tests is stated for the benchmarked mode -- f16 MFMA operands, 2^-11 per rounding.  The
    documented switch SLIMMOE_COMPUTE_DTYPE=bf16 (8-bit mantissas, 2^-8 per rounding) is held to 8 x the same bars (the README's
    8e-3 for the operator); f32 to the f16 bars."""
    import os
    return 8.0 if os.environ.get("SLIMMOE_COMPUTE_DTYPE", "f16") == "bf16" else 1.0


def float_bar(got, ref, tol=1e-3):
    """THE float bar of the operator tests (north_star: "fp tolerance <= 1e-3 on expert outputs"), stated once, in
    the convention test_grouped_gemm_matches_fp64_reference uses: max |diff| <= tol * max(1, max |ref|) AND
    relative L2 <= tol (``tol`` as stated for f16 operands; x dtype_factor() under SLIMMOE_COMPUTE_DTYPE=bf16).  Returns the
    numbers so that a failure prints the scale it was judged at."""
    tol = tol * dtype_factor()
    diff = got.double() - ref.double()
    scale = max(1.0, float(ref.abs().max()))
    max_abs, rel_l2 = float(diff.abs().max()), float(diff.norm() / ref.double().norm().clamp(min=1e-30))
    assert max_abs <= tol * scale and rel_l2 <= tol, dict(max_abs=max_abs, ref_abs_max=scale, rel_l2=rel_l2, tol=tol)
    return max_abs, scale, rel_l2
