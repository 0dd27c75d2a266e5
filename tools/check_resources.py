#!/usr/bin/env python3
"""Build gate for EVERY shipped kernel instantiation (ukf_kernel16 and the generic ukf_kernel): no scratch and no AGPRs, and
the tuned kernels keep the wavefronts per SIMD their design counts on (fp64: three, fp32: five) -- a register regression
that costs a wavefront fails the build instead of showing up as a slower bench line.

ROCm 7.2's hipcc places VGPR spill / live-range-split copies at the join label of a divergent `if`
in front of the EXEC restore; reached through s_cbranch_execz they save nothing (see the note in
slam-pose_estimation_amd/csrc/ukf_kernel16.hpp).  A kernel that needs spills is therefore
rejected at build time instead of being trusted.  Usage: check_resources.py <hipcc remark log>..."""
import re
import sys


def parse(text):
    out, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        for key, pat in (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("agpr", r"remark:\s+AGPRs: (\d+)"),
                         ("scratch", r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"remark:\s+Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"remark:\s+LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


def main(paths):
    # --allow-agpr: diagnostic builds with the fp64 one-wavefront-per-filter kernels (make GENERIC_F64=1)
    allow_agpr = "--allow-agpr" in paths
    paths = [p for p in paths if not p.startswith("--")]
    bad, rows = [], []
    for p in paths:
        for k in parse(open(p).read()):
            if "ukf_kernel" not in k["name"]:
                continue
            rows.append(k)
            if k.get("scratch", 0) or (k.get("agpr", 0) and not allow_agpr):
                bad.append(k)
    for k in rows:
        print(f"{k['name'][:70]:70s} vgpr={k.get('vgpr')} agpr={k.get('agpr')} scratch={k.get('scratch')} "
              f"occ={k.get('occupancy')}")
    if bad:
        print("ERROR: kernels with scratch or AGPR spills (unsafe with this toolchain):", [b["name"] for b in bad])
        return 1
    # occupancy floor of the tuned kernels (the register side; LDS is checked by the layout's static_asserts):
    # ukf_kernel16<double, ...> three wavefronts per SIMD (<= 168 VGPRs), ukf_kernel16<float, ...> five (<= 96)
    slow = [k for k in rows if "ukf_kernel16I" in k["name"] and
            k.get("occupancy", 0) < (3 if "ukf_kernel16Id" in k["name"] else 5)]
    if slow and "--no-occupancy-floor" not in sys.argv:
        print("ERROR: tuned kernels below their occupancy floor (fp64: 3, fp32: 5 wavefronts per SIMD):",
              [(k["name"], k.get("vgpr"), k.get("occupancy")) for k in slow])
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
