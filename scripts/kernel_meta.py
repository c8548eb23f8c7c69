#!/usr/bin/env python3
"""Code-object hygiene report: VGPRs, spills, scratch (private segment) and LDS of every kernel in the built objects
(socp_amd/_build/kernels_*.o), read from the AMDGPU metadata note of the gfx950 code objects.

    python scripts/kernel_meta.py [--all] [--json out.json]

Default: prints the kernels with spills or scratch, plus a per-family summary.  CPU-only (no GPU needed)."""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def kernels_of(obj, tmp):
    base = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, base)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", base], capture_output=True, text=True, check=True)
    recs = []
    for co in glob.glob(base + ".*gfx950*"):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            blk = ".agpr_count:" + blk
            f = {}
            for key in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                        "group_segment_fixed_size", "agpr_count"):
                m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                if m:
                    f[key] = m.group(1)
            if "name" in f:
                recs.append(f)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--json")
    ap.add_argument("--build", default=os.path.join(ROOT, "socp_amd", "_build"), help="directory with the kernels_*.o to read")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp()
    rows = []
    try:
        for obj in sorted(glob.glob(os.path.join(args.build, "kernels_*.o"))):
            recs = kernels_of(obj, tmp)
            names = demangle([r["name"] for r in recs])
            for r, nm in zip(recs, names):
                nm = re.sub(r"^void ", "", nm)
                nm = nm.replace("(anonymous namespace)::", "")
                nm = re.sub(r"\(.*$", "", nm)
                rows.append({"object": os.path.basename(obj), "kernel": nm.replace("socp::", ""),
                             "vgpr": int(r.get("vgpr_count", 0)), "agpr": int(r.get("agpr_count", 0)), "sgpr": int(r.get("sgpr_count", 0)),
                             "vgpr_spill": int(r.get("vgpr_spill_count", 0)), "sgpr_spill": int(r.get("sgpr_spill_count", 0)),
                             "scratch_bytes": int(r.get("private_segment_fixed_size", 0)), "lds_bytes": int(r.get("group_segment_fixed_size", 0))})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    bad = [r for r in rows if r["vgpr_spill"] or r["scratch_bytes"]]
    for r in (rows if args.all else bad):
        print("%-28s vgpr %3d agpr %3d spill %3d scratch %5d lds %6d  %s" % (r["object"], r["vgpr"], r["agpr"], r["vgpr_spill"], r["scratch_bytes"],
                                                                    r["lds_bytes"], r["kernel"]))
    print("%d kernels, %d with spills or scratch" % (len(rows), len(bad)), file=sys.stderr)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=0)


if __name__ == "__main__":
    main()
