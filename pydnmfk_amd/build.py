"""Build libdnmf_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

`python -m pydnmfk_amd.build [--report]` or `pydnmfk_amd.build.build_lib()`.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libdnmf_hip.so")
TUNE_LIB = os.path.join(ROOT, "tools", "_build", "libdnmf_hip_tune.so")   # -DDNMF_TUNING: experiment switches (tools only)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# per-unit compiler flags.  dnmf_kl16: MFMA accumulators in VGPRs (no v_accvgpr copies around the division of the KL
# products -- on gfx950 every fp32 vector instruction costs matrix-pipe time, csrc/dnmf_kl16.h)
UNIT_FLAGS = {"dnmf_kl16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale():
    if not os.path.exists(LIB):
        return True
    import glob
    deps = glob.glob(os.path.join(HERE, "csrc", "*")) + [os.path.join(ROOT, "include", "dnmf.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build_lib(force=False, report=False, tuning=False):
    """Compile the translation units csrc/*.hip (dnmf, dnmf_kl, dnmf_hals, dnmf_split; kernels in csrc/dnmf_*.h) side by side and link them into
    libdnmf_hip.so.  Returns the library path.
    `tuning=True` builds tools/_build/libdnmf_hip_tune.so instead: the same sources with -DDNMF_TUNING, in which the
    DNMF_* environment switches and the extra kernel variants of the A/B runs exist (tools/README.md); the shipped
    library reads no environment."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    out = TUNE_LIB if tuning else LIB
    if not tuning and not force and not report and not _stale():
        return LIB
    objdir = os.path.join(ROOT, "tools", "_build", "obj_tune" if tuning else "obj")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(HERE, "csrc")]
    if tuning:
        flags.append("-DDNMF_TUNING")
        flags += os.environ.get("DNMF_EXTRA_FLAGS", "").split()      # experiments: extra -D switches for the tuning build only
    if report:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    srcs = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
    only = os.environ.get("DNMF_BUILD_ONLY")          # development: recompile just this unit (e.g. dnmf_split), reuse the other objects
    objs = [os.path.join(objdir, os.path.basename(src)[:-4] + ".o") for src in srcs]

    def compile_one(pair):
        src, obj = pair
        if only and only not in os.path.basename(src) and os.path.exists(obj):
            return ""
        res = subprocess.run([HIPCC] + flags + UNIT_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj],
                             capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stderr)
            raise RuntimeError("hipcc failed on %s (%d)" % (os.path.basename(src), res.returncode))
        return res.stderr

    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        logs = list(ex.map(compile_one, zip(srcs, objs)))
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)
        raise RuntimeError("hipcc link failed (%d)" % res.returncode)
    if report:
        print(resource_report("".join(logs)))
    return out


def resource_report(log):
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", log)[1:]:
        name = blk.split("\n")[0].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True,
                                  text=True).stdout.strip()
        except OSError:
            pass
        name = name.replace("(anonymous namespace)::", "")

        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return m.group(1) if m else "?"

        rows.append("%-78s vgpr=%4s agpr=%4s spill=%3s scratch=%4s occ=%2s lds=%6s" % (
            name[:78], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
            g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    return "\n".join(rows)


if __name__ == "__main__":
    print(build_lib(force=True, report="--report" in sys.argv, tuning="--tuning" in sys.argv))
