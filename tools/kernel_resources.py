"""Register / spill / scratch / LDS figures of the pipeline kernels from the compiler's metadata (no GPU needed):
    python tools/kernel_resources.py [-D...]   (compiles boundplanner_amd/csrc/bmpc_pipeline.hip for gfx950 to assembly with the
                                                 product build's flags plus the given defines, ~1 min)"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "p.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", *sys.argv[1:], "--cuda-device-only", "-S",
                           os.path.join(R, "boundplanner_amd", "csrc", "bmpc_pipeline.hip"), "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
pat = r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)"
print(f"{'kernel':20s} {'VGPRs (incl. AGPRs)':>20s} {'AGPRs':>6s} {'spilled VGPRs':>14s} {'scratch B/lane':>15s} {'static LDS B':>13s}")
for ag, lds, name, priv, vg, sp in re.findall(pat, txt, re.S):
    n = re.sub(r"^_Z\d+", "", name).split("N4bmpc")[0]
    print(f"{n:20s} {vg:>20s} {ag:>6s} {sp:>14s} {priv:>15s} {lds:>13s}")
