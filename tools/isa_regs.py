"""Register / spill summary of every opp_gemm_kernel instantiation.
    cd /tmp/isa && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<csrc> -I<include> -c gemm_mfma.hip -save-temps
    python tools/isa_regs.py /tmp/isa/gemm_mfma-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys

s = open(sys.argv[1]).read()
i = s.find("amdhsa.kernels:")
for b in s[i:].split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    ag = int(b.split()[0])
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1))
    sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1))
    m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELi(\d+)ELb(\d)", name)
    if m:
        bm, bn, wm, wn, conv, abl, depth, h2 = m.groups()
        print("%sx%s waves %sx%s %s abl %s depth %s %s  agpr %3d vgpr %3d spill %d" %
              (bm, bn, wm, wn, "conv " if conv == "1" else "dense", abl, depth, "fp16x2" if h2 == "1" else "fp32  ", ag, vg, sp))
