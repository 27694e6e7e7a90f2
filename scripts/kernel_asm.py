"""Device assembly of the traversal kernels, normalised, and its hash.

    python scripts/kernel_asm.py sha [tree]           -> one sha256 over the normalised gfx950 assembly of the kernels the counter files
                                                          describe (PROFILED: production launch of k_horizon, its follow-up launch,
                                                          k_shadow_refill) of `tree` (default: this repository)
    python scripts/kernel_asm.py diff treeA treeB      -> per source file: identical, or the kernels whose bodies differ (exit status 1)
    python scripts/kernel_asm.py stamp                 -> writes profiles/kernel_asm.sha (the stamp profiles/valu_model.json, traffic.json and
                                                          valu_class_mix.json carry: they describe THIS machine code, not the source text)

Normalisation: comment-only changes, renamed probes that compile to nothing, moved lines -- none of them change the assembly; what does
differ between two compilations of the same code is the `__hip_cuid_<hash>` symbol (a hash of the source path and text), debug / ident
directives and `;` comments, which are dropped.  Used by scripts/asm_diff.sh, tests/test_boundary.py and bench.py (the stamp check).
"""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("hz_horizon.hip", "hz_shadow.hip", "hz_locations.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only"]


def assembly(tree, src, extra=()):
    d = os.path.join(tree, "horayzon_amd", "csrc")
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", "-", src], cwd=d, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("hipcc failed on %s/%s:\n%s" % (d, src, out.stderr[-2000:]))
    return out.stdout


def normalise(text):
    keep = []
    for line in text.split("\n"):
        s = line.split(";", 1)[0].rstrip()
        if not s.strip():
            continue
        t = s.strip()
        if "__hip_cuid_" in t or t.startswith((".ident", ".file", ".loc", ".section\t.debug", ".amdgcn_target", ".addrsig")):
            continue
        # (labels are numbered by the function's position in the file: `.LBB12_3` -> `.LBB_3`)
        t = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", t)
        t = re.sub(r"\.L(tmp|func_begin|func_end|JTI)\d+(_\d+)?", r".L\1", t)
        keep.append(t)
    return keep


def kernels(lines):
    """{symbol: body lines} of every function of a normalised listing (from `sym:` to `.Lfunc_end`)."""
    out, cur, name = {}, None, None
    for t in lines:
        m = re.match(r"^(_Z\w+):$", t)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if t.startswith(".Lfunc_end"):     # (normalised: no number)
                out[name] = cur
                cur = None
            else:
                cur.append(t)
    return out


def assemblies(tree):
    """{source: normalised listing}, the three compilations side by side."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(SOURCES)) as ex:
        return dict(zip(SOURCES, ex.map(lambda s_: normalise(assembly(tree, s_)), SOURCES)))


# the kernels the counter files under profiles/ describe: production launch of k_horizon, its follow-up launch, the shadow kernel
PROFILED = ("_ZN2hz9k_horizonILi2ELb0ELb1ELb0ELb0ELb0EEEvNS_13HorizonParamsE", "_ZN2hz9k_horizonILi2ELb0ELb1ELb0ELb0ELb1EEEvNS_13HorizonParamsE",
            "_ZN2hz15k_shadow_refillILb0ELb1EEEvNS_12ShadowParamsE")


def sha(tree=ROOT):
    """sha256 over the normalised bodies of the PROFILED kernels (a change of any other kernel of these files does not move it)."""
    h = hashlib.sha256()
    asm = assemblies(tree)
    found = {}
    for src in SOURCES:
        found.update(kernels(asm[src]))
    for name in PROFILED:
        if name not in found:
            raise RuntimeError("kernel %s not found in the assembly of %s" % (name, ", ".join(SOURCES)))
        h.update(name.encode()); h.update(b"\n")
        for t in found[name]:
            h.update(t.encode()); h.update(b"\n")
    return h.hexdigest()


def diff(a, b):
    rc = 0
    asm_a, asm_b = assemblies(a), assemblies(b)
    for src in SOURCES:
        ka, kb = kernels(asm_a[src]), kernels(asm_b[src])
        bad = sorted(k for k in set(ka) | set(kb) if ka.get(k) != kb.get(k))
        if not bad:
            print("%-18s identical device code (%d kernels)" % (src, len(ka)))
            continue
        rc = 1
        print("%-18s %d of %d kernels differ:" % (src, len(bad), len(set(ka) | set(kb))))
        for k in bad:
            la, lb = ka.get(k), kb.get(k)
            print("   %s  (%s -> %s instructions+directives)" % (k, len(la) if la is not None else "absent", len(lb) if lb is not None else "absent"))
    return rc


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "sha"
    if cmd == "sha":
        print(sha(sys.argv[2] if len(sys.argv) > 2 else ROOT))
    elif cmd == "diff":
        sys.exit(diff(sys.argv[2], sys.argv[3]))
    elif cmd == "stamp":
        v = sha()
        with open(os.path.join(ROOT, "profiles", "kernel_asm.sha"), "w") as f:
            f.write(v + "\n")
        print(v)
    else:
        sys.exit(__doc__)
