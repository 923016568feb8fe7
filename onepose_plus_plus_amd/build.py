"""Builds libopp_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m onepose_plus_plus_amd.build [--force] [--tuning]

`--tuning` builds libopp_hip_tuning.so with -DOPP_TUNING (phase-stamped / ablated kernel variants and the
environment knobs used by tools/; never loaded by the product path -- point OPP_HIP_LIB at it explicitly).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repository snapshot (git-ignored, not gpurun-ignored).  The library carries the
sha256 of the sources it was built from (`opp_source_hash()`, csrc/version.hip); `_lib.load()` compares it with the sources next
to it and refuses a stale binary.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_build")
LIB = os.path.join(HERE, "libopp_hip.so")
SOURCES = ["gemm_mfma.hip", "gemm_ss.hip", "enc_chain.hip", "enc_layer64.hip", "stem_direct.hip", "attention.hip", "backbone.hip", "kpt.hip",
           "coarse_match.hip", "fine.hip", "pnp.hip", "ingest.hip", "profile.hip", "bn_train.hip", "bankbuild.hip", "loss.hip", "linear_bwd.hip",
           "linattn_train.hip", "conv_bwd.hip", "train_misc.hip", "conv_tail.hip", "version.hip", "api.hip"]
HEADERS = ["opp_common.h", "opp_internal.h", "enc_frag.h", "pnp_math.h", os.path.join("..", "..", "include", "opp_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source extras.  The two files whose kernels split fp32 values into bf16 triples beside MFMAs: the SLP vectorizer pairs the
# residual subtractions of the split into v_pk_add_f32, which costs more there than the two scalar adds it replaces
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs"; +0.4 % images/s, profiles/r04_ab_slp_vectorizer.txt).  The scalar and
# the packed forms contract into FMAs differently, so arithmetic that two files must round identically (the LayerNorm of the fused
# encoder chain against the launch-per-Linear path) is written with explicit FMAs (opp_common.h: opp_ln_*), not left to the flags.
# (Every source without the vectorizer, `--variant noslp_all`: +0.2 % at most and the coarse-level chain loses its bit-identity through the
# attention arithmetic -- profiles/r04_ab_slp_all_sources.txt -- so the other files keep it.)
SOURCE_FLAGS = {"conv_bwd.hip": ["-fno-slp-vectorize"], "gemm_mfma.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def source_hash():
    """sha256 over the names and contents of every source / header of the library (None if the sources are not there)"""
    h = hashlib.sha256()
    for rel in sorted(SOURCES + HEADERS):
        path = os.path.join(CSRC, rel)
        if not os.path.exists(path):
            return None
        h.update(os.path.basename(rel).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, tuning=False, variant=None):
    """variant: A/B builds for tools/ (libopp_hip_<variant>.so): "slp" = gemm_mfma.hip with the SLP vectorizer left on; "noslp_all" = no source with it"""
    suffix = "_tuning" if tuning else ("_" + variant if variant else "")
    obj_dir = OBJ + suffix
    lib_path = LIB.replace(".so", suffix + ".so")
    flags = FLAGS + (["-DOPP_TUNING"] if tuning else [])
    extra = dict(SOURCE_FLAGS)
    if variant == "slp":
        extra.pop("gemm_mfma.hip", None)
    if variant == "noslp_all":
        flags = flags + ["-fno-slp-vectorize"]
    if variant and variant.startswith("def:"):          # tools: one extra macro definition, e.g. --variant def:R3_DMA_EVERY=2
        flags = flags + ["-D" + variant[4:]]
        suffix = "_" + variant[4:].replace("=", "").lower()
        obj_dir = OBJ + suffix
        lib_path = LIB.replace(".so", suffix + ".so")
    return _build(force, verbose, obj_dir, lib_path, flags, extra)


def _build(force, verbose, OBJ, LIB, FLAGS, SOURCE_FLAGS):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    digest = source_hash()
    stamp = os.path.join(OBJ, "src_hash.txt")
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        extra = list(SOURCE_FLAGS.get(src, []))
        if src == "version.hip":
            extra = ['-DOPP_SRC_HASH="%s"' % digest]
            if old != digest:                      # any source changed: the hash baked into this object changes with it
                jobs.append((s, o, extra))
                continue
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o, extra))

    def compile_one(job):
        s, o, extra = job
        cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for (s, o, _), r in ex.map(compile_one, jobs):
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed on %s:\n%s" % (s, r.stderr))
    objs = [os.path.join(OBJ, src.replace(".hip", ".o")) for src in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
    with open(stamp, "w") as f:
        f.write(digest or "")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, tuning="--tuning" in sys.argv,
                variant=sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None))
