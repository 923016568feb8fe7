"""Static audit of every kernel of the library (CPU only: hipcc cross-compiles gfx950).  Per kernel: VGPRs, scratch bytes, and the two code-generation
accidents round 4 found the hard way in conv_wgrad_kernel --
  * waterfall loops: a buffer descriptor (or any SGPR operand) that is a per-lane value makes the compiler wrap the instruction in a
    v_readfirstlane / v_cmp_eq / s_and_saveexec loop (one trip per distinct value, and the load pipeline drains every trip);
  * packed fp32 VALU (v_pk_add/mul/fma_f32) inside kernels that issue MFMAs: the SLP vectorizer's pairs cost more beside MFMAs than the scalar
    instructions they replace (the sources with split arithmetic in their MFMA loops are built with -fno-slp-vectorize, build.py).
Exit status 1 if any kernel has a waterfall loop or scratch that is not on the allow list.
    python tools/isa_audit.py [--tuning] > profiles/rNN_isa_audit.txt"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onepose_plus_plus_amd import build  # noqa: E402

# kernels known to use scratch: the PnP solvers keep per-thread 3x4 / 6x6 systems in indexed local arrays (304-400 B), the focal-loss
# forward indexes a 6-float weight struct (24 B) -- none of them is on the matching forward's path
# the 128 x 224 ring tile (gemm_mfma.hip, config 27) sits at the 256-register limit of a two-waves-per-SIMD kernel: 104-124 B of spills, all of them
# outside the K loops except the division constant of the K-tail cursor (2 dword reloads per chunk) -- located with the MFMA index of every
# scratch instruction when the tile was written
ALLOW_SCRATCH = ("pnp_", "focal_fwd_kernel", "opp_gemm_kernelILi128ELi224E")


def audit_source(src, tuning, tmp):
    out = os.path.join(tmp, src + ".s")
    flags = build.FLAGS + (["-DOPP_TUNING"] if tuning else []) + build.SOURCE_FLAGS.get(src, [])
    flags = [f for f in flags if f != "-fPIC"]
    extra = ['-DOPP_SRC_HASH="audit"'] if src == "version.hip" else []
    cmd = [build._hipcc()] + flags + extra + ["-S", "--cuda-device-only", "-I", build.CSRC, "-I", os.path.join(ROOT, "include"),
                                               os.path.join(build.CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return src, None, r.stderr[-2000:]
    text = open(out).read()
    rows = []
    # kernel bodies: from "<name>:" (a .type ...,@function symbol that has an .amdhsa_kernel block) to its s_endpgm / .Lfunc_end
    kernels = re.findall(r"\.amdhsa_kernel (\S+)", text)
    for k in kernels:
        m = re.search(r"^%s:.*?^\.Lfunc_end\d+:" % re.escape(k), text, re.S | re.M)
        body = m.group(0) if m else ""
        vg = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(k), text)
        ag = re.search(r"\.set %s\.num_agpr, (\d+)" % re.escape(k), text)
        sc = re.search(r"\.set %s\.private_seg_size, (\d+)" % re.escape(k), text)
        lines = body.split("\n")
        water = 0
        for i, ln in enumerate(lines):
            if "v_readfirstlane_b32" in ln:
                win = "\n".join(lines[i:i + 12])
                if "v_cmp_eq_u" in win and "s_and_saveexec_b64" in win and ("buffer_" in win or "image_" in win or "s_cbranch_execnz" in win):
                    water += 1
        # consecutive readfirstlanes of one descriptor count once
        water = len(re.findall(r"(?:v_readfirstlane_b32[^\n]*\n\s*){2,}[^\n]*\n?(?:[^\n]*\n){0,6}?[^\n]*s_and_saveexec_b64", body)) if water else 0
        mfma = len(re.findall(r"\bv_mfma_", body))
        pk = len(re.findall(r"\bv_pk_(?:add|mul|fma)_f32", body))
        rows.append((k, int(vg.group(1)) if vg else -1, int(ag.group(1)) if ag else 0, int(sc.group(1)) if sc else 0, water, mfma, pk))
    return src, rows, ""


def main():
    tuning = "--tuning" in sys.argv
    bad = 0
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda s: audit_source(s, tuning, tmp), build.SOURCES))
    print("%-96s %5s %5s %8s %10s %6s %8s" % ("kernel (demangled prefix stripped)", "vgpr", "agpr", "scratch", "waterfall", "mfma", "pk_f32"))
    for src, rows, err in results:
        if rows is None:
            print("%s: COMPILE FAILED\n%s" % (src, err))
            bad = 1
            continue
        print("# %s (%d kernels)%s" % (src, len(rows), "  [-fno-slp-vectorize]" if build.SOURCE_FLAGS.get(src) else ""))
        for k, vg, ag, sc, water, mfma, pk in rows:
            short = re.sub(r"^_ZN\d+_GLOBAL__N_1", "", k)[:96]
            flag = ""
            if water or (sc and not any(a in k for a in ALLOW_SCRATCH)):
                flag = "   <-- CHECK"
                bad = 1
            print("%-96s %5d %5d %8d %10d %6d %8d%s" % (short, vg, ag, sc, water, mfma, pk if mfma else 0, flag))
    sys.exit(bad)


if __name__ == "__main__":
    main()
