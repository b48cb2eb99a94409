"""Developer: registers / occupancy / scratch of every kernel before and after a change, from two device assembly dumps:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S -o before.s bear_amd/csrc/bear_hip.hip   (each side)
    python scripts/dev/kernel_resources.py before.s after.s
Prints the kernels whose numbers differ, with a mark where the waves per SIMD changed (round 6: two more vector registers in a
shared helper moved dm_ref_items_kernel from 6 waves per SIMD to 5 and configs[3] from 128 to 164 us per step)."""
import re
import sys


def parse(path):
    out, name = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
        for key, pat in (("vgpr", r"; NumVgprs: (\d+)"), ("waves_per_simd", r"; Occupancy: (\d+)"), ("scratch", r"; ScratchSize: (\d+)")):
            m = re.match(pat, line)
            if m and name:
                out.setdefault(name, {})[key] = int(m.group(1))
    return out


a, b = parse(sys.argv[1]), parse(sys.argv[2])
for k in sorted(a):
    if k in b and a[k] != b[k] and "waves_per_simd" in a[k]:
        print(k[:90], a[k], "->", b[k], "  <<< waves per SIMD" if a[k].get("waves_per_simd") != b[k].get("waves_per_simd") else "")
