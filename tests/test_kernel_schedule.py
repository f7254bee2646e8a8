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


# ------------------------------------------------------------------ the Toom-Cook kernel (gnn_fused_tc.hip): 18 barriers per step, a ring of 3 slots
def tc_schedule(steps: int = 4, ahead: int = 2):
    """Phases of the default kernel's step loop, transcribed from gnn_fused_tc.hip.  Regions: bufX.rows / bufX.carry (x1), bufY.rows
    (x2 as f32, then x3) / bufY.carry, ring0..2 (one k16 unit of transformed activations each), prow0 / prow1.  A chunk's version is
    (conv, step, chunk index).  `ahead` = how many chunks the helpers run ahead of the matrix waves (the kernel: 2; 3 would overwrite
    the slot the matrix waves are reading)."""
    def prow(s):
        return f"prow{s & 1}"

    def ring(c):
        return f"ring{c % 3}"

    yield [("all", "w", prow(0), ("prow", 0), True), ("all", "w", prow(1), ("prow", 1), True)]
    yield [("helper", "r", prow(0), ("prow", 0), True), ("helper", "w", "bufX.rows", ("x1", 0), True)]       # gather of the first step
    yield [("helper", "r", "bufX.rows", ("x1", 0), True), ("helper", "r", "bufX.carry", ("x1c", -1), True),   # V2 chunks 0, 1 of the first step
           ("helper", "w", ring(0), ("c2", 0, 0), True), ("helper", "w", ring(1), ("c2", 0, 1), True)]

    def conv_phase(conv, s, c, src_rows, src_ver, carry_reg, carry_ver):
        """interval behind barrier b_c of a conv loop: the matrix waves consume chunk c (and fetch the first fragment of chunk c + 1),
        the helpers produce chunk c + ahead"""
        ph = [("matrix", "r", ring(c), (conv, s, c), True)]
        if c + 1 < 8:
            ph.append(("matrix", "r", ring(c + 1), (conv, s, c + 1), True))
        if c + ahead < 8:
            ph += [("helper", "r", src_rows, src_ver, True), ("helper", "r", carry_reg, carry_ver, True),
                   ("helper", "w", ring(c + ahead), (conv, s, c + ahead), True)]
        return ph

    for s in range(steps):
        for c in range(8):                                                   # ---- conv2: b_0 .. b_7
            ph = conv_phase("c2", s, c, "bufX.rows", ("x1", s), "bufX.carry", ("x1c", s - 1))
            if c == 5:
                ph.append(("helper", "r", prow(s + 1), ("prow", s + 1), True))          # gather round 0: table loads
            if c >= 6:
                ph.append(("helper", "r", "bufX.rows", ("x1", s), True))                # head A's pair products; x1 carry rows into registers
            if c == 7:
                ph += [("helper", "r", prow(s + 1), ("prow", s + 1), True),             # gather round 1: table loads
                       ("matrix", "w", "bufY.rows", ("x2", s), True)]                   # conv2 epilogue (the helpers touch bufX only)
            yield ph
        # ---- B1 .. b'_0.  Round 6: head A's y @ w_v comes from the 9-mer table (no LDS access), the matrix waves make V3 chunk 1 while
        # their table rows travel, the helpers chunk 0 and head A's last pair pass (x1(s) stays in bufX until the gather behind b'_0)
        yield [("matrix", "r", "bufY.rows", ("x2", s), True), ("matrix", "r", "bufY.carry", ("x2c", s - 1), True),
               ("matrix", "w", ring(1), ("c3", s, 1), True),
               ("matrix", "r", prow(s + 1), ("prow", s + 1), True),                     # row indices of the next step's table rows
               ("helper", "r", "bufX.rows", ("x1", s), True),
               ("helper", "r", "bufY.rows", ("x2", s), True), ("helper", "r", "bufY.carry", ("x2c", s - 1), True),
               ("helper", "w", ring(0), ("c3", s, 0), True)]
        for c in range(8):                                                   # ---- conv3: b'_0 .. b'_7
            ph = conv_phase("c3", s, c, "bufY.rows", ("x2", s), "bufY.carry", ("x2c", s - 1))
            if c == 0:
                ph += [("helper", "w", "bufX.carry", ("x1c", s), False),                # own registers -> carry rows
                       ("helper", "w", "bufX.rows", ("x1", s + 1), True)]               # gather rounds 0, 1 (held in registers since b_7 / B1)
            if c == 4:
                ph.append(("helper", "r", prow(s + 1), ("prow", s + 1), True))          # gather round 2: table loads
            if c == 6:
                ph += [("helper", "w", "bufX.rows", ("x1", s + 1), True),               # gather round 2
                       ("helper", "r", "bufY.rows", ("x2", s), True), ("helper", "w", "bufY.carry", ("x2c", s), False)]
            if c == 7:
                ph += [("helper", "w", prow(s + 2), ("prow", s + 2), True),
                       ("matrix", "w", "bufY.rows", ("x3", s), True)]                   # conv3 epilogue overwrites x2 with x3
            yield ph
        yield [("matrix", "r", "bufY.rows", ("x3", s), True),                           # ---- B0 .. b_0: w_v B(s), then V2(s+1) chunk 1
               ("matrix", "r", "bufX.rows", ("x1", s + 1), True), ("matrix", "r", "bufX.carry", ("x1c", s), True),
               ("matrix", "w", ring(1), ("c2", s + 1, 1), True),
               ("helper", "r", "bufY.rows", ("x3", s), True),                           # head B's pair products
               ("helper", "r", "bufX.rows", ("x1", s + 1), True), ("helper", "r", "bufX.carry", ("x1c", s), True),
               ("helper", "w", ring(0), ("c2", s + 1, 0), True)]


def test_toomcook_kernel_schedule_orders_every_lds_producer_and_consumer_with_a_barrier():
    for steps in (1, 2, 63):
        assert check(list(tc_schedule(steps))) == []


def test_toomcook_model_notices_a_dropped_barrier_and_a_ring_overrun():
    phases = list(tc_schedule(3))
    # B1 dropped: the conv2 epilogue (last conv2 interval) and the helpers' V3 chunks 0, 1 would share a phase
    k = 3 + 7
    merged = phases[:k] + [phases[k] + phases[k + 1]] + phases[k + 2:]
    assert any("bufY.rows" in msg for _, msg in check(merged))
    # helpers three chunks ahead: chunk c + 3 lands in the slot the matrix waves read in the same interval
    assert any("ring" in msg for _, msg in check(list(tc_schedule(3, ahead=3))))
    # one chunk ahead is too little: the matrix waves fetch the first fragment of chunk c + 1 while they finish chunk c
    assert any("ring" in msg for _, msg in check(list(tc_schedule(3, ahead=1))))


# ------------------------------------------------------------------ the model against the source: barrier count and order parsed from the .hip
def _tc_source():
    """gnn_tc_dev.h + gnn_fused_tc.hip as the shipped library compiles them: every conditional on a TC_* macro (#ifdef / #ifndef / #if defined(..) /
    #elif defined(..) chains; all TC_* macros are undefined in the product build) is resolved, other conditionals are kept."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, stack = [], []          # stack of [keep this branch?, is a TC_ switch?, has a branch of the chain been taken?]
    csrc = os.path.join(root, "genomad_amd", "csrc")
    lines = []
    for name in ("gnn_tc_dev.h", "gnn_fused_tc.hip"):     # the device helpers (conv_tc and its barriers) live in the header since round 6
        if os.path.exists(os.path.join(csrc, name)):
            lines += open(os.path.join(csrc, name)).readlines()
    for line in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|elif|else|endif|if)\b\s*(.*)", line)
        if not m:
            if all(f[0] for f in stack):
                out.append(line)
            continue
        d, rest = m.group(1), re.sub(r"//.*", "", m.group(2)).strip()
        sw = re.fullmatch(r"defined\((TC_\w+)\)", rest) if d in ("if", "elif") else None
        if d in ("ifdef", "ifndef") and rest.split()[0].startswith("TC_"):
            stack.append([d == "ifndef", True, d == "ifndef"])
        elif d == "if" and sw:
            stack.append([False, True, False])
        elif d in ("ifdef", "ifndef", "if"):
            stack.append([True, False, True])
            if all(f[0] for f in stack):
                out.append(line)
        elif d == "elif" and stack and stack[-1][1]:
            assert sw, "an #elif of a TC_ chain that is not a TC_ switch: " + line
            stack[-1][0] = False                      # defined(TC_...) is false in the product build
        elif d == "else" and stack and stack[-1][1]:
            stack[-1][0] = not stack[-1][2]
            stack[-1][2] = True
        elif d == "endif" and stack:
            top = stack.pop()
            if not top[1] and all(f[0] for f in stack):
                out.append(line)
        elif all(f[0] for f in stack):
            out.append(line)
    assert not stack
    return "".join(out)


def test_tc_source_filter_resolves_every_form_of_a_switch(tmp_path, monkeypatch):
    """`#if defined(TC_X) / #elif defined(TC_Y) / #else` chains and plain #ifdef / #ifndef switches all resolve to the
    product build (every TC_* macro undefined); a non-TC conditional stays as it is."""
    import os
    text = ("a\n#ifdef TC_A\nb\n#else\nc\n#endif\n#if defined(TC_B)\nd\n#elif defined(TC_C)\ne\n#else\nf\n#endif\n#ifndef TC_D\ng\n#else\nh\n#endif\n"
            "#ifdef OTHER\ni\n#endif\n#if defined(TC_E)\nj\n#endif\nk\n")
    d = tmp_path / "genomad_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "gnn_fused_tc.hip").write_text(text)
    real = os.path.dirname
    monkeypatch.setattr(os.path, "dirname", lambda p_: str(tmp_path) if p_ == real(os.path.abspath(__file__)) else real(p_))
    assert _tc_source().split() == ["a", "c", "f", "g", "#ifdef", "OTHER", "i", "#endif", "k"]


def _loop_body(src, start):
    """text of the `for (int step = ...) {` loop that starts at or after `start` (brace matched)"""
    i = src.index("for (int step = s_begin; step < s_hi; ++step) {", start)
    depth, j = 0, src.index("{", i)
    while True:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        if depth == 0:
            return src[i:j + 1], j
        j += 1


def test_toomcook_model_has_the_barriers_the_source_has():
    """The hand-written model above can drift from the kernel.  This parses gnn_fused_tc.hip: both roles of a workgroup must pass the
    same number of workgroup barriers per step, that number must be the number of phases the model yields per step, and the helpers'
    barriers must come in the order the model assumes (b_0 .. b_7, B1, b'_0 .. b'_7, B0 - the labels in the source's comments)."""
    import re
    src = _tc_source()
    matrix, end = _loop_body(src, src.index("if (!helper) {"))
    helper, _ = _loop_body(src, end)
    # matrix waves: two conv loops (each: one barrier in front of unit 0 + one per further unit) + B1 + B0
    conv = src[src.index("__device__ __forceinline__ void conv_tc("):src.index("template <bool F16>")]
    assert len(re.findall(r"\bTC_BARRIER\(\);", conv)) == 2 and "k % 8 == 0 && k > 0" in conv and "make_integer_sequence<int, 64>" in conv
    per_conv = 1 + (64 // 8 - 1)
    n_matrix = matrix.count("conv_tc(") * per_conv + len(re.findall(r"\bTC_BARRIER_W\(\);", matrix)) + len(re.findall(r"\bTC_BARRIER\(\);", matrix))
    assert matrix.count("conv_tc(") == 2
    # helper waves: every barrier is an HBAR / HBAR_W macro call with its label in the trailing comment
    calls = re.findall(r"\bHBAR(?:_W)?\(\d+, \d+\);\s*//\s*(?:-+\s*)?([bB]'?_?\d)", helper)
    assert len(re.findall(r"\bHBAR(?:_W)?\(", helper)) == len(calls), "a helper barrier without a label comment"
    want = [f"b_{c}" for c in range(8)] + ["B1"] + [f"b'_{c}" for c in range(8)] + ["B0"]
    assert calls == want, calls
    per_step_model = len(list(tc_schedule(2))) - len(list(tc_schedule(1)))
    assert n_matrix == len(calls) == per_step_model == 18


# ------------------------------------------------------------------ the k-mer-table kernel (gnn_fused_tk.hip): 9 barriers per step, two alternating row buffers
def tk_schedule(steps: int = 4, store_at: int = 6, chunk0_at: int = 7):
    """Phases of gnn_fused_tk.hip's step loop.  Regions: buf0 / buf1 (rows of x2(s), overwritten by x3(s); x2(s+1) lands in the other),
    ring0..2, dirty (the list of rows no 14-mer indexes + its counter).  `store_at` = the conv interval in which the helpers store the
    gathered x2(s+1) rows (the kernel: behind c_6, when V3(s) is complete ... and x3(s-1) long dead), `chunk0_at` = the interval in which
    they make V3(s+1) chunk 0 (the kernel: behind c_7, ring slot 0 = chunk 6's slot)."""
    def buf(s):
        return f"buf{s & 1}"

    def ring(c):
        return f"ring{c % 3}"

    # prologue: all waves gather x2(0) (+ the rows the tap tables fill), barrier; matrix: V3(0) chunk 1, helpers: chunk 0
    yield [("all", "w", buf(0), ("x2", 0), True), ("all", "w", "dirty", ("d", 0), True)]
    yield [("all", "r", "dirty", ("d", 0), True), ("all", "w", buf(0), ("x2", 0), False)]          # dirty_rows_fill: each thread its own channel of a listed row
    yield [("all", "w", "dirty", ("d", -1), False)]                                               # one thread resets the list
    yield [("matrix", "r", buf(0), ("x2", 0), True), ("matrix", "w", ring(1), ("c3", 0, 1), True),
           ("helper", "r", buf(0), ("x2", 0), True), ("helper", "w", ring(0), ("c3", 0, 0), True)]
    for s in range(steps):
        last = s + 1 >= steps
        for c in range(8):                                                   # ---- conv3: c_0 .. c_7
            ph = [("matrix", "r", ring(c), ("c3", s, c), True)]
            if c + 1 < 8:
                ph.append(("matrix", "r", ring(c + 1), ("c3", s, c + 1), True))
            if c + 2 < 8:
                ph += [("helper", "r", buf(s), ("x2", s), True), ("helper", "w", ring(c + 2), ("c3", s, c + 2), True)]
            if c == 0 and not last:
                ph.append(("helper", "w", "dirty", ("d", s + 1), True))      # row indices of x2(s+1): rows no table holds go onto the list
            if c == store_at and not last:
                ph += [("helper", "w", buf(s + 1), ("x2", s + 1), True),     # a wave stores the rows it requested (other waves read them); listed rows are not stored ...
                       ("helper", "r", "dirty", ("d", s + 1), False)]        # ... but filled from the tap tables (list written behind c_0)
            if c == chunk0_at:
                ph.append(("helper", "w", "dirty", ("d", -1), False))        # one thread resets the list
                if not last:
                    ph += [("helper", "r", buf(s + 1), ("x2", s + 1), True), ("helper", "w", ring(0), ("c3", s + 1, 0), True)]
            if c == 7:
                ph.append(("matrix", "w", buf(s), ("x3", s), True))          # conv3 epilogue overwrites x2(s) with x3(s)
            yield ph
        ph = [("matrix", "r", buf(s), ("x3", s), True),                      # ---- E .. c_0: w_v B(s), then V3(s+1) chunk 1
              ("helper", "r", buf(s), ("x3", s), True)]                      # head B's pair products (three passes)
        if not last:
            ph += [("matrix", "r", buf(s + 1), ("x2", s + 1), True), ("matrix", "w", ring(1), ("c3", s + 1, 1), True)]
        yield ph


def test_kmer_table_kernel_schedule_orders_every_lds_producer_and_consumer_with_a_barrier():
    for steps in (1, 2, 3, 63):
        assert check(list(tk_schedule(steps))) == []


def test_kmer_table_model_notices_wrong_placements():
    # x2(s+1) stored while the helpers still transform x2(s)?  No: other buffer.  But stored BEFORE head B's last pass of step s-1 would
    # be wrong in the real kernel only if that pass ran beside the conv loop; the model's invariants: rows stored behind c_7 (same
    # interval as chunk 0 of the next step reads them) ...
    assert any("buf" in msg for _, msg in check(list(tk_schedule(3, store_at=7))))
    # ... and chunk 0 of the next step written behind c_6 lands in the slot the matrix waves read chunk 6 from
    assert any("ring0" in msg for _, msg in check(list(tk_schedule(3, chunk0_at=6))))


def test_kmer_table_model_has_the_barriers_the_source_has():
    """Both roles of gnn_fused_tk.hip pass the same number of workgroup barriers per step, in the order the model assumes."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "genomad_amd", "csrc", "gnn_fused_tk.hip")).read()
    i = src.index("if (!helper) {")
    matrix, end = _loop_body_from(src, "for (int step = s_lo; step < s_hi; ++step) {", i)
    helper, _ = _loop_body_from(src, "for (int step = s_lo; step < s_hi; ++step) {", end)
    n_matrix = matrix.count("conv_tc(") * 8 + len(re.findall(r"\bTC_BARRIER_W\(\);", matrix)) + len(re.findall(r"\bTC_BARRIER\(\);", matrix))
    assert matrix.count("conv_tc(") == 1
    calls = re.findall(r"\bHBAR(?:_W)?\(\d+, \d+\);\s*//\s*(?:-+\s*)?([cE]_?\d?)", helper)
    assert len(re.findall(r"\bHBAR(?:_W)?\(", helper)) == len(calls), "a helper barrier without a label comment"
    assert calls == [f"c_{c}" for c in range(8)] + ["E"], calls
    per_step_model = len(list(tk_schedule(3))) - len(list(tk_schedule(2)))
    assert n_matrix == len(calls) == per_step_model == 9
    assert "__syncthreads" not in matrix and "__syncthreads" not in helper      # a __syncthreads() would drain the weight loads in flight


def test_kmer_table_helpers_wait_once_behind_E_and_gather_two_rows_per_request():
    """Two facts the step loop's timing rests on (profiles/r06/ab/ab12..14): (1) everything the helpers requested in front of E is
    waited for ONCE, right behind E, with the builtin the compiler's wait-count pass sees - every pair pass sits under a condition, and
    without it the compiler waits with vmcnt(0) at the top of each pass, behind the store of the pass before; (2) table rows travel in
    16-byte requests, two 512-byte rows per request (the vector memory pipe is bound by the number of requests behind E)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "genomad_amd", "csrc", "gnn_fused_tk.hip")).read()
    hdr = open(os.path.join(root, "genomad_amd", "csrc", "gnn_tc_dev.h")).read()
    i = src.index("if (!helper) {")
    _, end = _loop_body_from(src, "for (int step = s_lo; step < s_hi; ++step) {", i)
    helper, _ = _loop_body_from(src, "for (int step = s_lo; step < s_hi; ++step) {", end)
    code = re.sub(r"//[^\n]*", "", helper)
    # the loop is rotated: E and the wait are the LAST statements of an iteration, the next one starts with the located entry, the burst of
    # row requests (requested and stored inside one iteration) and only then the pair products of the step before
    m = re.search(r"HBAR\(13, 8\);\s*__builtin_amdgcn_s_waitcnt\(0x0F70\);\s*\}\s*$", code)
    assert m, "the helpers' one wait behind E (vmcnt(0), lgkmcnt / expcnt untouched: 0x0F70) is gone or moved"
    order = [code.index(t) for t in ("locate();", "x2_rows_issue<0, X2_PER_WAVE>", "pass_compute(p0, jp", "HBAR_W(14, 8)", "x2_rows_store(xr", "pass_issue(p0, jb", "HBAR(13, 8)")]
    assert order == sorted(order), order
    assert 'asm volatile("s_waitcnt vmcnt' not in helper                     # an inline-asm wait is invisible to the compiler's pass
    assert re.search(r"struct X2Rows \{\s*u32x4 v\[X2_PER_WAVE / 2\];", src)
    assert re.search(r"struct WvaRows \{\s*u32x4 v\[WVA_ROWS_PER_WAVE / 2\];", hdr)
    assert "raw_buffer_load_b64(tbl" not in hdr and "u32x2*>(p + lane * 8)" not in src


def _loop_body_from(src, head, start):
    i = src.index(head, start)
    depth, j = 0, src.index("{", i)
    while True:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        if depth == 0:
            return src[i:j + 1], j
        j += 1
