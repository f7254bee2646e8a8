"""Executable model of the streaming kernels' LDS schedule (gnn_fused_x3.hip, gnn_fused_c6.hip): who writes and who reads
which LDS region in which barrier phase, for both roles of a workgroup (4 matrix waves, 4 helper waves).

A workgroup barrier is the ONLY thing that orders two different waves here, so the rule the model checks is: inside one phase
(the code between two consecutive workgroup barriers) no region may be written by one role and touched by the other, and no
region written by SOME helper wave may be read by the helper waves in the same phase unless the reader is the writing thread
itself.  Every read must also see the version (step index) of the data it expects.  The model is transcribed by hand from the
kernel's step loop (the table in DESIGN.md section 4.3); its purpose is to pin the reasoning, and to show that the round-2
schedule — pair rows written at the top of a step and gathered by other waves in the same phase — violates the rule while
this round's (written between B3 and B4 of the previous step into a parity double buffer) does not.
"""
import pytest

HELPER_CROSS_WAVE = "other helper waves"      # marks a helper-side access whose counterpart may be ANOTHER helper wave


def schedule(round2_pair_rows: bool, steps: int = 6):
    """Yield phases; a phase is a list of (role, op, region, version, cross_wave) with op in {"r", "w"}.
    Regions: bufX rows (x1), bufX carry, bufY rows (x2 / x3 share them), bufY carry, prow0 / prow1 (pair-row buffers)."""
    def prow_region(s):            # which buffer holds the pair rows of step s
        return "prow0" if round2_pair_rows else f"prow{s & 1}"

    # prologue: all threads write the pair rows of steps 0 (and, this round, 1); barrier; helpers gather x1(0); barrier
    pro = [("all", "w", prow_region(0), ("prow", 0), True)]
    if not round2_pair_rows:
        pro.append(("all", "w", prow_region(1), ("prow", 1), True))
    yield pro
    yield [("helper", "r", prow_region(0), ("prow", 0), True), ("helper", "w", "bufX.rows", ("x1", 0), True)]
    for s in range(steps):
        # ---- phase A: after B4(s-1) [or the prologue barrier], before B1(s)
        a = [("matrix", "r", "bufX.rows", ("x1", s), True), ("matrix", "r", "bufX.carry", ("x1c", s - 1), True),     # w_v A(s), conv2(s)
             ("helper", "r", "bufX.rows", ("x1", s), True)]                                                         # pairs A(s), carry read
        if s > 0:
            a += [("matrix", "r", "bufY.rows", ("x3", s - 1), True),      # w_v B(s-1): issued by the matrix waves after B4(s-1)
                  ("helper", "r", "bufY.rows", ("x3", s - 1), True)]      # pairs B(s-1)
        if round2_pair_rows:
            a += [("helper", "w", "prow0", ("prow", s + 1), True),        # thread h writes prow[h] at the top of the step ...
                  ("helper", "r", "prow0", ("prow", s + 1), True)]        # ... and the OTHER waves' gathers read it later in the same phase
        else:
            a += [("helper", "r", prow_region(s + 1), ("prow", s + 1), True)]      # gather loads of x1(s+1)
        yield a
        # ---- phase B: B1(s) .. B2(s)
        yield [("matrix", "w", "bufY.rows", ("x2", s), True),                       # conv2 epilogue
               ("helper", "w", "bufX.carry", ("x1c", s), False)]                    # own registers -> carry rows (each thread its chunk)
        # ---- phase C: B2(s) .. B3(s)
        yield [("matrix", "r", "bufY.rows", ("x2", s), True), ("matrix", "r", "bufY.carry", ("x2c", s - 1), True),   # conv3(s)
               ("helper", "w", "bufX.rows", ("x1", s + 1), True),                   # x1(s+1) (f16c6: its first half already in phase B)
               ("helper", "r", "bufY.rows", ("x2", s), True)]                       # x2 carry rows into registers
        # ---- phase D: B3(s) .. B4(s)
        d = [("matrix", "w", "bufY.rows", ("x3", s), True),                         # conv3 epilogue overwrites x2 with x3
             ("helper", "w", "bufY.carry", ("x2c", s), False)]
        if not round2_pair_rows:
            d.append(("helper", "w", prow_region(s + 2), ("prow", s + 2), True))
        yield d


def check(phases):
    """Return a list of violations: (phase index, description)."""
    bad, version = [], {"bufX.carry": ("x1c", -1), "bufY.carry": ("x2c", -1)}
    for i, phase in enumerate(phases):
        writes = [(role, region, ver, cross) for role, op, region, ver, cross in phase if op == "w"]
        reads = [(role, region, ver, cross) for role, op, region, ver, cross in phase if op == "r"]
        for wrole, wreg, wver, wcross in writes:
            for rrole, rreg, rver, rcross in reads:
                if rreg != wreg:
                    continue
                if rrole != wrole and "all" not in (rrole, wrole):
                    bad.append((i, f"{wrole} writes {wreg} while {rrole} reads it in the same phase"))
                elif wcross and rcross:
                    bad.append((i, f"{wreg} written and read by different {wrole} waves in the same phase (no barrier in between)"))
            for orole, oreg, over, _ in writes:
                if oreg == wreg and orole != wrole and "all" not in (orole, wrole):
                    bad.append((i, f"{wrole} and {orole} both write {wreg} in the same phase"))
        for rrole, rreg, rver, _ in reads:
            have = version.get(rreg)
            same_phase_writer = any(w[1] == rreg for w in writes)
            if not same_phase_writer and have != rver and not (rver[1] < 0):
                bad.append((i, f"{rrole} reads {rreg} expecting {rver}, it holds {have}"))
        for _, wreg, wver, _ in writes:
            version[wreg] = wver
    return bad


def test_this_rounds_schedule_orders_every_lds_producer_and_consumer_with_a_barrier():
    assert check(list(schedule(round2_pair_rows=False))) == []


def test_round_2_pair_rows_are_the_one_unordered_pair():
    bad = check(list(schedule(round2_pair_rows=True)))
    assert bad and all("prow0" in msg for _, msg in bad)
    assert any("different helper waves" in msg for _, msg in bad)


def test_the_model_notices_a_missing_barrier():
    """Sanity of the checker itself: merging the phases B3..B4 and B4..B1 (as if B4 were dropped) must be flagged — the
    matrix waves' conv3 epilogue would overwrite rows the helpers' pair products still read."""
    phases = list(schedule(round2_pair_rows=False))
    merged = phases[:5] + [phases[5] + phases[6]] + phases[7:]
    assert any("bufY.rows" in msg for _, msg in check(merged))


@pytest.mark.parametrize("steps", [1, 2, 47])
def test_schedule_holds_for_short_and_full_windows(steps):
    assert check(list(schedule(False, steps))) == []
