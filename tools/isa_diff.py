#!/usr/bin/env python3
"""Which kernels' gfx950 instruction streams differ between two builds?  For changes that must not touch the device code
of the default path (a refactoring, an opt-in variant added as a template parameter) when no GPU is at hand:
    python tools/isa_diff.py <git-rev-A> [<git-rev-B> | WORKTREE]
compiles kimera_semantics_amd/csrc/ks_hip.hip of both revisions with `hipcc --cuda-device-only -S` (a temporary
checkout of csrc/ and include/ for a revision) and compares every kernel's instructions, comments and local label
numbers aside.  Template kernels are matched by mangled name; a kernel whose template parameter list changed shows up as
removed + added."""
import difflib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(rev, out):
    if rev == "WORKTREE":
        src = ROOT
    else:
        src = tempfile.mkdtemp(prefix="isa_")
        for d in ("kimera_semantics_amd/csrc", "include"):
            tar = subprocess.run(["git", "-C", ROOT, "archive", rev, d], check=True, capture_output=True).stdout
            subprocess.run(["tar", "-x", "-C", src], input=tar, check=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S", "-o", out,
                    "ks_hip.hip"], cwd=os.path.join(src, "kimera_semantics_amd", "csrc"), check=True, stderr=subprocess.DEVNULL)


def kernels(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M):
        name = m.group(1)
        body = re.sub(r';[^\n]*', '', m.group(2))
        body = re.sub(r'\.LBB\d+_\d+', 'L', body)
        out[name] = [l.strip() for l in body.split('\n') if l.strip() and name not in l]
    return out


def main():
    a_rev = sys.argv[1]
    b_rev = sys.argv[2] if len(sys.argv) > 2 else "WORKTREE"
    with tempfile.TemporaryDirectory() as tmp:
        fa, fb = os.path.join(tmp, "a.s"), os.path.join(tmp, "b.s")
        asm_of(a_rev, fa)
        asm_of(b_rev, fb)
        a, b = kernels(fa), kernels(fb)
    same = [k for k in a if k in b and a[k] == b[k]]
    changed = [k for k in a if k in b and a[k] != b[k]]
    print(f"{len(same)} kernels identical, {len(changed)} changed, {len(set(a) - set(b))} only in {a_rev}, {len(set(b) - set(a))} only in {b_rev}")
    for k in changed:
        d = [l for l in difflib.unified_diff(a[k], b[k], lineterm='', n=0) if not l.startswith(('@@', '---', '+++'))]
        print(f"  changed: {k}  ({len(d)} differing lines)")
    for k in sorted(set(a) - set(b)):
        print(f"  only in {a_rev}: {k}")
    for k in sorted(set(b) - set(a)):
        print(f"  only in {b_rev}: {k}")
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
