"""Build libnefnet_hip.so (gfx950) in-tree with hipcc.  `python -m electrocardio_panorama_amd.csrc.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# conv_mfma.hip is compiled as four objects (-DNEF_MFMA_PART=1..4: see the top of that file), in parallel with the other sources
MFMA_PARTS = 4
SOURCES = ["conv_mfma.hip", "conv_h2.hip", "conv_h2w.hip", "conv_bww_glds.hip", "stem.hip", "elementwise.hip", "roi.hip", "convt_theta.hip", "pano_h.hip", "metrics.hip"]
LIB = os.path.join(HERE, "libnefnet_hip.so")
# per-source extra flags.  Every source that issues matrix instructions is built WITHOUT SLP vectorisation: the packed-fp32
# instructions it creates (v_pk_fma_f32 with op_sel on registers a ds_read_b128 has just returned, in conv_h2.hip's epilogue)
# intermittently produced 0.0 in lanes 48..63 on a loaded chip (profiles/r05_pk_fp32_hazard.md: not reproduced
# in isolation, no root cause) -- and the matrix-core guide lists packed fp32 beside MFMAs as an anti-lever anyway.  Round 5
# extended the flag from the two split-fp16 files to all of them (conv_mfma.hip alone had 10 k such instructions).
# Round 6: the four files without matrix instructions (elementwise / roi / convt_theta / metrics) are built the same way -- they read
# LDS-returned pairs too (block reductions), and one rule for the whole library is easier to audit than a per-file argument.
_NO_SLP = ["-fno-slp-vectorize"]
# Experimental kernel forms, NOT in the default library (`python -m electrocardio_panorama_amd.csrc.build --with-experiments`, or
# NEF_BUILD_EXPERIMENTS=1): sources under tools/experiments/ that the default path can never reach.  conv_h2p.hip = the producer /
# consumer form of conv_h2_kernel (bit-identical, measured slower); conv_h2.hip reaches it through weak hooks.
EXPERIMENTS = ["conv_h2p.hip"]
EXP_DIR = os.path.join(ROOT, "tools", "experiments")
EXTRA_FLAGS = {s: _NO_SLP for s in SOURCES + EXPERIMENTS}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
         "-I", os.path.join(ROOT, "include"), "-I", HERE]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _with_experiments(flag=None):
    return bool(flag) if flag is not None else os.environ.get("NEF_BUILD_EXPERIMENTS") == "1"


def _src(s):
    return os.path.join(EXP_DIR if s in EXPERIMENTS else HERE, s)


def _units(sources):
    """(source, extra -D flags, object path) for every object of the library."""
    out = []
    for s in sources:
        if s == "conv_mfma.hip":
            out += [(s, [f"-DNEF_MFMA_PART={i}"], os.path.join(HERE, f"conv_mfma_p{i}.o")) for i in range(1, MFMA_PARTS + 1)]
        else:
            out.append((s, [], os.path.join(HERE, s.replace(".hip", ".o"))))
    return out


def _stamp():
    return os.path.join(HERE, ".experiments")      # present <=> the library on disk was linked with the experimental forms


def needs_build(experiments=None):
    if not os.path.exists(LIB):
        return True
    exp = _with_experiments(experiments)
    if exp != os.path.exists(_stamp()):
        return True
    t = os.path.getmtime(LIB)
    deps = [_src(s) for s in SOURCES + (EXPERIMENTS if exp else [])] + [os.path.join(HERE, "nef_common.h"),
                                                                        os.path.join(ROOT, "include", "nefnet_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines, sources=("conv_mfma.hip",)):
    """A/B builds for kernel experiments: csrc/variants/lib<name>.so with extra -D flags on `sources` (the other objects
    are taken from the regular build).  Select one at run time with NEF_LIB=<path>."""
    build(force=False, verbose=False)
    vdir = os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    objs = []
    srcs = SOURCES + (EXPERIMENTS if (os.path.exists(_stamp()) or any(s in EXPERIMENTS for s in sources)) else [])
    procs = []
    for s, dflags, o in _units(srcs):
        if s in sources or (s in EXPERIMENTS and not os.path.exists(o)):
            o = os.path.join(vdir, f"{name}_{os.path.basename(o)}")
            procs.append(subprocess.Popen([hipcc()] + FLAGS + ["-w"] + EXTRA_FLAGS.get(s, []) + dflags + [f"-D{d}" for d in defines] + ["-c", _src(s), "-o", o]))
        objs.append(o)
    if any(p_.wait() != 0 for p_ in procs):
        raise RuntimeError("hipcc failed")
    lib = os.path.join(vdir, f"lib{name}.so")
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=True, experiments=None):
    exp = _with_experiments(experiments)
    if not force and not needs_build(exp):
        return LIB
    objs = []
    procs = []
    for s, dflags, o in _units(SOURCES + (EXPERIMENTS if exp else [])):
        cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(s, []) + dflags + ["-c", _src(s), "-o", o]
        procs.append((s + " " + " ".join(dflags), subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out.strip()):
            sys.stderr.write(f"--- hipcc {s} ---\n{out}\n")
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    if exp:
        open(_stamp(), "w").close()
    elif os.path.exists(_stamp()):
        os.remove(_stamp())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments=True if "--with-experiments" in sys.argv else None))
