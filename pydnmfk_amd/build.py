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
SRC = os.path.join(HERE, "csrc", "dnmf.hip")
LIB = os.path.join(HERE, "libdnmf_hip.so")
TUNE_LIB = os.path.join(ROOT, "tools", "_build", "libdnmf_hip_tune.so")   # -DDNMF_TUNING: experiment switches (tools only)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale():
    if not os.path.exists(LIB):
        return True
    import glob
    deps = glob.glob(os.path.join(HERE, "csrc", "*")) + [os.path.join(ROOT, "include", "dnmf.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build_lib(force=False, report=False, tuning=False):
    """Compile csrc/dnmf.hip (one translation unit, kernels in csrc/dnmf_*.h) -> libdnmf_hip.so.  Returns the library path.
    `tuning=True` builds tools/_build/libdnmf_hip_tune.so instead: the same source with -DDNMF_TUNING, in which the
    DNMF_* environment switches and the extra kernel variants of the A/B runs exist (tools/README.md); the shipped
    library reads no environment."""
    out = TUNE_LIB if tuning else LIB
    if not tuning and not force and not report and not _stale():
        return LIB
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(HERE, "csrc"),
           "-shared", "-fPIC", "-o", out, SRC]
    if tuning:
        cmd.insert(1, "-DDNMF_TUNING")
    if report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)
        raise RuntimeError("hipcc failed (%d)" % res.returncode)
    if report:
        print(resource_report(res.stderr))
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
