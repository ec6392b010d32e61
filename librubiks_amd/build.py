"""
Builds librubiks_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m librubiks_amd.build [--force] [--tune]

--tune additionally builds benchmarks/librubiks_hip_tune.so: the same sources with -DRK_TUNING and csrc/rk_tuning.hip in the
place of rk_cube_kernels.hip, which adds the kernel-shape variants and geometry diagnostics used by benchmarks/tune_*.py.
The shipped library carries none of them (rk_tuning.hip is never one of its translation units).

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librubiks_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
	"--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
	"-ffp-contract=off",          # search statistics must round like NumPy (no fused multiply-add)
	"-Wall", "-Wno-unused-function",
	"-Wl,-rpath,/opt/rocm/lib",
]


TUNING_UNIT = "rk_tuning.hip"          # tuning build only: includes rk_cube_kernels.hip and adds the kernel variants / diagnostics


def sources():
	"""The translation units of the shipped library: every .hip file but the tuning unit."""
	return sorted(s for s in glob.glob(os.path.join(CSRC, "*.hip")) if os.path.basename(s) != TUNING_UNIT)


def tune_sources():
	"""The tuning build: rk_tuning.hip stands in for rk_cube_kernels.hip (it includes it)."""
	return sorted([s for s in sources() if os.path.basename(s) != "rk_cube_kernels.hip"] + [os.path.join(CSRC, TUNING_UNIT)])


def stale() -> bool:
	if not os.path.exists(OUT):
		return True
	t = os.path.getmtime(OUT)
	deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
	return any(os.path.getmtime(d) > t for d in deps)


TUNE_OUT = os.path.join(HERE, "..", "benchmarks", "librubiks_hip_tune.so")


def build_tune(verbose: bool = False) -> str:
	"""The tuning build (one compile of everything with -DRK_TUNING; not cached, not shipped)."""
	cmd = [HIPCC] + FLAGS + ["-DRK_TUNING", "-o", os.path.abspath(TUNE_OUT)] + tune_sources()
	if verbose:
		print(" ".join(cmd))
	subprocess.run(cmd, check=True)
	return os.path.abspath(TUNE_OUT)


#: what the last build() did: "compiled" = sources that went through hipcc, "reused" = objects that were newer than their sources
#: and every header, "linked" = whether the shared object was linked again (RK_BUILD_FORCE=1 or force=True compiles everything)
LAST = {"compiled": [], "reused": [], "linked": False}


def build(force: bool = False, verbose: bool = False) -> str:
	force = force or os.environ.get("RK_BUILD_FORCE", "") not in ("", "0")
	LAST.update(compiled=[], reused=[os.path.basename(s) for s in sources()], linked=False)
	if not force and not stale():
		return OUT
	LAST["reused"] = []
	objs = []
	for src in sources():
		obj = os.path.join(CSRC, os.path.basename(src)[:-4] + ".o")
		if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
				[os.path.getmtime(src)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))]
				+ [os.path.getmtime(h) for h in glob.glob(os.path.join(HERE, "..", "include", "*.h"))]):
			cmd = [HIPCC] + [f for f in FLAGS if f != "-shared" and not f.startswith("-Wl")] + ["-c", src, "-o", obj]
			if verbose:
				print(" ".join(cmd))
			subprocess.run(cmd, check=True)
			LAST["compiled"].append(os.path.basename(src))
		else:
			LAST["reused"].append(os.path.basename(src))
		objs.append(obj)
	cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", OUT] + objs
	if verbose:
		print(" ".join(cmd))
	subprocess.run(cmd, check=True)
	LAST["linked"] = True
	return OUT


if __name__ == "__main__":
	print(build(force="--force" in sys.argv, verbose=True))
	if "--tune" in sys.argv:
		print(build_tune(verbose=True))
