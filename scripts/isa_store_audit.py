#!/usr/bin/env python3
"""Compile-only audit of the device code for the mis-compiled store of round 4 (commit 08f2e7a).

What happened there (profiles/r05_fault_08f2e7a_isa.txt): the interior-node rows of segment_residual were a three-way
divergent branch with an `emit(row, .)` / `emit(row + D, .)` pair of stores in every arm.  hipcc 7.2 sank the second store of
an unrolled component to the join -- ONE flat_store whose address register is a phi of the arms -- and left that register
UNDEFINED on one arm (the listing says `implicit-def`).  A wave that takes only that arm stores through whatever the
register held last.

This script builds every device translation unit with `--cuda-device-only -S` (no GPU needed), walks each kernel's control
flow graph and reports every store whose ADDRESS register can reach the store from an `implicit-def` marker in divergent
context with no definition in between (a may-be-undefined address).  What it cannot see: exec masks (a definition under one
mask and a use under another look alike; a masked write counts as a definition, so the walk under-reports) and uniform flags
(markers that belong to a uniform skip are ignored: the compiler re-tests the same scalar flag before every use).  A report is
therefore a candidate to READ, not a proof; `--expect-clean` makes any candidate an error.  On the headers of 08f2e7a it
reports exactly the store that faulted, in all eight residual_lane_kernel<Lqr1D, ...> instantiations.

    python scripts/isa_store_audit.py                       # all TUs of socp_amd/csrc + tests/plugin
    python scripts/isa_store_audit.py --rev 08f2e7a --only plugin   # the historical headers, the example plugin
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# translation unit -> flags, as socp_amd/csrc/Makefile and socp_amd/host/Makefile build them
UNITS = {
    "kernels_exact": ("socp_amd/csrc/kernels_exact.hip", ["-ffp-contract=off"]),
    "kernels_fast": ("socp_amd/csrc/kernels_fast.hip", ["-ffp-contract=fast"]),
    "kernels_interceptor": ("socp_amd/csrc/kernels_interceptor.hip", ["-ffp-contract=off"]),
    "kernels_interceptor_fast": ("socp_amd/csrc/kernels_interceptor_fast.hip", ["-ffp-contract=fast"]),
    "kernels_solver": ("socp_amd/csrc/kernels_solver.hip", ["-ffp-contract=off"]),
    "kernels_factor_fast": ("socp_amd/csrc/kernels_factor_fast.hip", ["-ffp-contract=fast"]),
    "capi": ("socp_amd/csrc/capi.cpp", ["-ffp-contract=off", "-x", "hip"]),
    "plugin": ("tests/plugin/lqr1d_plugin.hip", ["-ffp-contract=off"]),
}

REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]")
STORE = re.compile(r"^\s*(flat_store|global_store|scratch_store|buffer_store)_\w+\s+(.*)$")
IMPDEF = re.compile(r"implicit-def: \$(vgpr|sgpr)(\d+)(?:_\w*?(\d+))?\s*$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)")


def regs_of(tok):
    """registers named by one operand token, as a set of ('v'|'s', index)"""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def split_kernels(text):
    """name -> list of lines, for every function of the listing"""
    kernels, name, cur = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*; @", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if name is not None:
            cur.append(line)
            if line.strip().startswith(".Lfunc_end"):
                name = None
    return kernels


def audit_kernel(lines):
    """candidates: (line number in kernel, store text, register) for stores whose address may be undefined"""
    # basic blocks: a block starts at a label or after a branch
    blocks, starts, cur = [], {}, []
    for idx, line in enumerate(lines):
        m = LABEL.match(line)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            starts[m.group(1)] = len(blocks)
        cur.append((idx, line))
        if BRANCH.match(line) or line.strip().startswith("s_endpgm") or line.strip().startswith("s_setpc"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for bi, blk in enumerate(blocks):
        s = set()
        last = blk[-1][1] if blk else ""
        m = BRANCH.match(last)
        if m:
            if m.group(2) in starts:
                s.add(starts[m.group(2)])
            if m.group(1) != "s_branch" and bi + 1 < len(blocks):
                s.add(bi + 1)
        elif not (last.strip().startswith("s_endpgm") or last.strip().startswith("s_setpc")) and bi + 1 < len(blocks):
            s.add(bi + 1)
        succ.append(s)

    # Which implicit-def markers count.  The compiler also writes the marker where a UNIFORM branch skips a definition whose
    # uses sit behind the same uniform condition (a flag in a scalar pair, re-tested later: `if (rows_per_block) ... `): every
    # value of the other arm then reads as possibly undefined to a walk that cannot follow the flag.  The mis-compiled store
    # came out of a DIVERGENT three-way branch, so a marker counts only in divergent context: its block is the target or the
    # fall-through of an exec-mask branch, or the block changes exec before the marker.
    EXECBR = ("s_cbranch_execz", "s_cbranch_execnz")
    div_entry = [False] * len(blocks)
    for bi, blk in enumerate(blocks):
        m = BRANCH.match(blk[-1][1]) if blk else None
        if m and m.group(1) in EXECBR:
            if m.group(2) in starts:
                div_entry[starts[m.group(2)]] = True
            if bi + 1 < len(blocks):
                div_entry[bi + 1] = True

    # forward data flow: the set of registers that MAY be undefined (marked implicit-def in divergent context and not written
    # since).  A write under a partial exec mask counts as a definition: the walk under-reports rather than drowning the reader.
    def transfer(bi, undef, report):
        undef = set(undef)
        divergent = div_entry[bi]
        term = BRANCH.match(blocks[bi][-1][1]) if blocks[bi] else None
        if term and term.group(1) in EXECBR:
            divergent = True
        elif term and term.group(1) != "s_branch":
            divergent = False        # the marker belongs to a UNIFORM skip (s_cbranch_vcc* / scc*): the flag pattern above
        for idx, line in blocks[bi]:
            m = IMPDEF.search(line)
            if m and line.lstrip().startswith(";"):
                if divergent:
                    lo = int(m.group(2))
                    hi = int(m.group(3)) if m.group(3) else lo
                    for i in range(lo, hi + 1):
                        undef.add(("v" if m.group(1) == "vgpr" else "s", i))
                continue
            code = line.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            parts = code.split(None, 1)
            op = parts[0]
            ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
            if STORE.match(line):
                addr = regs_of(ops[0]) if ops else set()
                if op.startswith("global_store") and len(ops) > 2:
                    addr |= regs_of(ops[2])              # saddr form: global_store v_off, v_data, s[base]
                bad = addr & undef
                if bad and report is not None:
                    report.append((idx, code, sorted(bad)))
                continue
            if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "ds_write", "ds_store", "s_endpgm")):
                continue
            if ops and (ops[0] == "exec" or "saveexec" in op) and not (term and term.group(1) not in EXECBR and term.group(1) != "s_branch"):
                divergent = True
            if ops:
                undef -= regs_of(ops[0])
                if op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
                    undef -= regs_of(ops[1])
        return undef

    n = len(blocks)
    inn = [set() for _ in range(n)]
    work = list(range(n))
    while work:
        b = work.pop()
        out = transfer(b, inn[b], None)
        for s in succ[b]:
            if not out <= inn[s]:
                inn[s] |= out
                if s not in work:
                    work.append(s)
    report = []
    for b in range(n):
        transfer(b, inn[b], report)
    return report


def build_listing(unit, tree, outdir):
    src, flags = UNITS[unit]
    out = os.path.join(outdir, unit + ".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(tree, "socp_amd/csrc"),
           "-I" + os.path.join(tree, "include")] + flags + ["--cuda-device-only", "-S", os.path.join(tree, src), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def checkout(rev, dest):
    """the headers and sources of `rev` under dest (git archive: nothing of the working tree is touched)"""
    p = subprocess.Popen(["git", "-C", ROOT, "archive", rev, "socp_amd/csrc", "include", "tests/plugin"], stdout=subprocess.PIPE)
    subprocess.check_call(["tar", "-x", "-C", dest], stdin=p.stdout)
    p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rev", help="audit the sources of this commit instead of the working tree")
    ap.add_argument("--only", action="append", help="unit name(s): " + ", ".join(UNITS))
    ap.add_argument("--keep", help="directory to keep the listings in")
    ap.add_argument("--expect-clean", action="store_true", help="exit 1 when any candidate is found")
    a = ap.parse_args()
    units = a.only or list(UNITS)
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    tree = ROOT
    if a.rev:
        tree = os.path.join(tmp, "tree")
        os.makedirs(tree)
        checkout(a.rev, tree)
    outdir = a.keep or tmp
    os.makedirs(outdir, exist_ok=True)
    total = 0
    for u in units:
        if not os.path.exists(os.path.join(tree, UNITS[u][0])):
            print("%-26s (not in this revision)" % u)
            continue
        listing = build_listing(u, tree, outdir)
        kernels = split_kernels(open(listing).read())
        found = 0
        stores = 0
        for name, lines in kernels.items():
            stores += sum(1 for l in lines if STORE.match(l))
            for idx, code, bad in audit_kernel(lines):
                found += 1
                print("  CANDIDATE %s\n    +%d: %s   <- may be undefined: %s" % (name, idx, code, ", ".join("%s%d" % r for r in bad)))
        total += found
        print("%-26s %4d functions, %6d global/flat/scratch stores, %d with a possibly undefined address" % (u, len(kernels), stores, found))
    print("total candidates: %d" % total)
    shutil.rmtree(tmp, ignore_errors=True)                    # (the listings stay only where --keep named a directory)
    sys.exit(1 if (a.expect_clean and total) else 0)


if __name__ == "__main__":
    main()
