#!/usr/bin/env python3
"""Static check of the generated code of csrc/spmm_quad.hip's pipelined loop (q_units_fast).

The loop issues its loads and stores from inline asm and waits for them with a hand-counted `s_waitcnt vmcnt(n)`; with
Q_DEPTH >= 2 requests are still in flight at the loop's back edge.  The compiler does not know that an asm load's destination
is not there yet, so ANY instruction of its own that touches such a register before the wait that covers it (a copy the
register allocator inserted to split a live range, a rotation at the back edge) would read or clobber data in flight.
This walks the loop twice in layout order, tracks which registers are destinations of loads in flight (a load issued in
sub-iteration e has landed after the wait that ends sub-iteration e + depth - 1) and reports every compiler-emitted
instruction that names one of them.  Used by tests/test_abi.py; `python scripts/check_quad_isa.py quad.s` prints the findings.
"""
import re
import sys

STORES, OPS_PER_ITERATION = 4, 15  # per super-unit: 4 stores; 8 index chunks + 1 scale + 2 (extents, rows) loads + 4 stores
KERNEL_RE = r"^(_ZN\S*spmm_quad_kernel%s[^:\s]*):[^\n]*\n(.*?)s_endpgm"


def _regs(tok):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out += list(range(int(a), int(b) + 1))
    out += [int(x) for x in re.findall(r"\bv(\d+)\b", tok)]
    return out


MARK_RE = re.compile(r"s_waitcnt vmcnt\((\d+)\)\s*; q_units_fast set (\d+) of (\d+)")


def check_kernel(text, variant):
    """-> dict(depth, in_flight, sets, asm_loads, asm_stores, loops, blocks, violations=[(line, text, regs)])

    Forward dataflow over the kernel's basic blocks.  A fact is (phase, {register: waits still to pass}); phase = the register
    set named by the last hand-written wait passed ("P": none since a wait for everything).  The source guarantees that the
    loop's waits are passed in the order 0, 1, .., sets - 1, 0, ..: paths that would pass them in another order exist in the
    CFG (an early exit shares its block with the back edge) but are never executed, and are not followed."""
    m = re.search(KERNEL_RE % variant, text, re.M | re.S)
    if not m:
        raise ValueError(f"spmm_quad_kernel<{variant}> not found")
    lines = [l.strip() for l in m.group(2).split("\n")]
    marks = [MARK_RE.match(t) for t in lines]
    marks = [mm for mm in marks if mm]
    if not marks:
        raise ValueError("the pipelined loop's waits were not found")
    in_flight, n_sets = int(marks[0].group(1)), int(marks[0].group(3))
    assert all(int(mm.group(1)) == in_flight and int(mm.group(3)) == n_sets for mm in marks)
    depth = n_sets - 1
    assert in_flight == STORES + (depth - 1) * OPS_PER_ITERATION, (in_flight, depth)
    # ---- basic blocks: [label?] instructions ... terminator; successors = branch target + fall-through
    blocks, cur, in_asm_now = [], {"label": None, "ins": []}, False
    for i, t in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", t)
        if mm:
            blocks.append(cur)
            cur = {"label": mm.group(1), "ins": []}
            continue
        if not t or (t.startswith(";") and not t.startswith(";;#ASM")) or t.startswith("."):
            continue
        if in_asm_now and t.startswith("s_waitcnt vmcnt"):  # a hand-written wait starts a block of its own
            blocks.append(cur)
            cur = {"label": None, "ins": [(-1, ";;#ASMSTART")]}
        if t.startswith(";;#ASMSTART"):
            in_asm_now = True
        elif t.startswith(";;#ASMEND"):
            in_asm_now = False
        cur["ins"].append((i, t))
        if re.match(r"s_c?branch", t) or t.startswith("s_setpc") or t.startswith("s_endpgm"):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    blocks.append(cur)
    blocks = [b for b in blocks if b["label"] or b["ins"]]
    by_label = {b["label"]: k for k, b in enumerate(blocks) if b["label"]}
    succ = []
    for k, b in enumerate(blocks):
        out, last = [], b["ins"][-1][1] if b["ins"] else ""
        mm = re.match(r"(s_c?branch\S*)\s+(\.LBB\d+_\d+)", last)
        if mm:
            out.append(by_label[mm.group(2)])
        if not (last.startswith("s_branch") or last.startswith("s_setpc") or last.startswith("s_endpgm")) and k + 1 < len(blocks):
            out.append(k + 1)
        succ.append(out)

    def transfer(phase, state, b, report=None):
        """-> (phase, state) at the block's end, or None when the path is infeasible"""
        st, in_asm = dict(state), False
        for i, t in b["ins"]:
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm:
                if t.startswith("global_load"):
                    for r in _regs(t.split(",")[0]):
                        st[r] = depth
                elif t.startswith("s_waitcnt vmcnt"):
                    mm = MARK_RE.match(t)
                    if mm:
                        k = int(mm.group(2))
                        if not ((phase == "P" and k == 0) or (phase != "P" and (phase + 1) % n_sets == k)):
                            return None
                        phase, st = k, {r: w - 1 for r, w in st.items() if w > 1}
                    elif re.match(r"s_waitcnt vmcnt\(0\)", t):
                        phase, st = "P", {}
                    else:
                        raise ValueError(f"unexpected hand-written wait: {t}")
                continue
            if re.match(r"s_waitcnt\s+vmcnt\(0\)", t):
                st = {}  # (a compiler wait for everything: wasteful, not wrong; the phase stays)
                continue
            ops = t.split(None, 1)[1] if " " in t else ""
            touched = [r for r in _regs(ops) if r in st]
            if touched and report is not None:
                report.append((i, t, touched))
        return phase, st

    # ---- feasibility of (block, phase): some path reaches the next wait the source allows, or the end of the pipelined region
    #      (a wait for everything / the kernel's end), through blocks without hand-written waits
    def first_wait(b):
        for _i, t in b["ins"]:
            if t.startswith(";;#ASM"):
                continue
            return t if t.startswith("s_waitcnt vmcnt") else None
        return None
    kind = []  # per block: None | "end" | set index
    for k, b in enumerate(blocks):
        w = first_wait(b)
        mm = MARK_RE.match(w) if w else None
        ends = any(t.startswith("s_endpgm") for _i, t in b["ins"]) or not succ[k]
        kind.append(int(mm.group(2)) if mm else ("end" if (w or ends) else None))
    pred = [[] for _ in blocks]
    for k, out in enumerate(succ):
        for s_ in out:
            pred[s_].append(k)
    feasible = {}
    for ph in ["P"] + list(range(n_sets)):
        want = 0 if ph == "P" else (ph + 1) % n_sets
        seen = set(k for k in range(len(blocks)) if kind[k] == "end" or kind[k] == want)
        todo = list(seen)
        while todo:
            k = todo.pop()
            for q in pred[k]:
                if q not in seen and (kind[q] is None or True):
                    # q's own leading wait (if any) is passed BEFORE its body: the body's phase is what reaches k
                    seen.add(q)
                    if kind[q] is None:
                        todo.append(q)
        feasible[ph] = seen
    facts = {(0, "P"): {}}  # (block, phase at entry) -> state at entry
    work = [(0, "P")]
    while work:
        k, ph = work.pop()
        res = transfer(ph, facts[(k, ph)], blocks[k])
        if res is None:
            continue
        ph2, out = res
        for s_ in succ[k]:
            key = (s_, ph2)
            if key not in facts:
                facts[key] = dict(out)
                work.append(key)
            else:
                merged = dict(facts[key])
                for r, w in out.items():
                    if merged.get(r, 0) < w:
                        merged[r] = w
                if merged != facts[key]:
                    facts[key] = merged
                    work.append(key)
    violations = []
    for (k, ph), st in facts.items():
        # the phase of the block's BODY (behind its leading wait, if it has one) decides whether it is ever executed this way
        res = transfer(ph, st, blocks[k])
        if res is not None and k in feasible[res[0]]:
            transfer(ph, st, blocks[k], violations)
    violations = sorted(set((i, t, tuple(r)) for i, t, r in violations))
    asm_loads = asm_stores = 0
    in_asm = False
    for t in lines:
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            asm_loads += t.startswith("global_load")
            asm_stores += t.startswith("global_store")
    return {"depth": depth, "in_flight": in_flight, "sets": n_sets, "asm_loads": asm_loads, "asm_stores": asm_stores,
            "loops": len(marks) // n_sets, "blocks": len(blocks), "violations": violations}


def main():
    text = open(sys.argv[1]).read()
    bad = 0
    for variant in ("IfLb0ELi0E", "ItLb0ELi0E", "IfLb0ELi2E", "ItLb0ELi2E"):
        res = check_kernel(text, variant)
        print(variant, {k: v for k, v in res.items() if k != "violations"}, "violations:", len(res["violations"]))
        for v in res["violations"][:20]:
            print("   ", v)
        bad += len(res["violations"])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
