"""Remove measurement switches from a source file: every conditional on one of the named macros is resolved as if the macro were
UNDEFINED (the product build), the other conditionals are kept.  Round 6 used it once to take the TC_ABL_* / TC_PROBE_* / TC_EMU_* ...
branches (several of them wrong by construction) out of genomad_amd/csrc/gnn_fused_tc.hip; the switch builds of
rounds 4 and 5 are reproducible from the commits named in profiles/HISTORY.md.

    python scripts/strip_switches.py FILE MACRO_PREFIX_OR_NAME [...]  > out
"""
import re
import sys


def strip(text, names):
    def hit(macro):
        return any(macro == n or (n.endswith("*") and macro.startswith(n[:-1])) for n in names)

    out = []
    # stack entries: [kind, keep_now, resolved]   kind: "sw" (a switch: directive lines dropped) or "other" (kept verbatim)
    stack = []
    for line in text.splitlines(keepends=True):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        active = all(k for _, k, _ in stack)
        if not m:
            if active:
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2).strip()
        rest = re.sub(r"//.*", "", rest).strip()
        if d in ("ifdef", "ifndef"):
            macro = rest.split()[0]
            if hit(macro):
                stack.append(["sw", d == "ifndef", d == "ifndef"])
                continue
            stack.append(["other", True, True])
        elif d == "if":
            mm = re.fullmatch(r"defined\((\w+)\)", rest)
            if mm and hit(mm.group(1)):
                stack.append(["sw", False, False])
                continue
            stack.append(["other", True, True])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "sw":
                mm = re.fullmatch(r"defined\((\w+)\)", rest)
                assert mm and hit(mm.group(1)), "an #elif of a switch chain that is not itself a switch: " + line
                top[1] = False
                continue
        elif d == "else":
            top = stack[-1]
            if top[0] == "sw":
                top[1] = not top[2]
                top[2] = True
                continue
        elif d == "endif":
            top = stack.pop()
            if top[0] == "sw":
                continue
            if all(k for _, k, _ in stack):
                out.append(line)
            continue
        if active:
            out.append(line)
    assert not stack
    return "".join(out)


if __name__ == "__main__":
    sys.stdout.write(strip(open(sys.argv[1]).read(), sys.argv[2:]))
