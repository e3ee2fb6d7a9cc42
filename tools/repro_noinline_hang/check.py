#!/usr/bin/env python3
"""Static check for the long-branch / return-address clobber described in README.md (no GPU needed).
  python tools/repro_noinline_hang/check.py            # the non-inlined variant: shows the two bodies that hang
  python tools/repro_noinline_hang/check.py --product  # the product build: exit code 1 if any non-entry function is at risk
  python tools/repro_noinline_hang/check.py --run      # GPU box: run the hanging variant under `timeout 60` (expected: killed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "boundplanner_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast"]


def asm(defs):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "p.s")
        subprocess.check_call(["hipcc", *FLAGS, *defs, "--cuda-device-only", "-S", os.path.join(CS, "bmpc_pipeline.hip"), "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def functions(txt):
    """(name, body, is_kernel) of every function in the assembly"""
    kernels = set(re.findall(r"\.amdhsa_kernel\s+(\S+)", txt))
    for m in re.finditer(r"^([_A-Za-z][\w.$]*):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        yield m.group(1), m.group(2), m.group(1) in kernels


def report(txt):
    bad = []
    for name, body, is_kernel in functions(txt):
        pairs = re.findall(r"s_getpc_b64\s+(s\[\d+:\d+\])\s*\n\.Lpost_getpc", body)
        if is_kernel or not pairs:
            continue
        saved = bool(re.search(r"v_writelane_b32\s+v\d+,\s*s30\b", body))
        risk = "s[30:31]" in pairs and not saved
        short = re.sub(r"^_ZN?\d*(?:bmpc)?L?\d*", "", name)[:40]
        print(f"{short:42s} long branches via {sorted(set(pairs))}  return address saved: {saved}  {'<-- CLOBBERS ITS RETURN ADDRESS' if risk else 'ok'}")
        if risk:
            bad.append(name)
    return bad


if "--run" in sys.argv:
    sys.exit(subprocess.call(["bash", os.path.join(os.path.dirname(__file__), "run_on_gpu.sh")]))
product = "--product" in sys.argv
bad = report(asm([] if product else ["-DBMPC_KBODY_CALL"]))
print(("product build: " if product else "non-inlined bodies: ") + (f"{len(bad)} function(s) at risk" if bad else "no function at risk"))
sys.exit(1 if (product and bad) else 0)
